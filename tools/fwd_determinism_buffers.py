"""Which saved forward buffer is the first to differ between two identical steps?  (bf16 path; tools/fwd_determinism.py showed
the loss is not bit-reproducible.)  Hashes every workspace tensor after two forward_backward calls on the same model and batch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper
from oracle import whisper_oracle as O
dev = "cuda:0"
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "whisper_small_ref_b8_10steps.json")))
params = O.init_params(O.make_config("small"), seed=gold["seed"], dtype=torch.float32)
feats, labels = O.create_dummy_pool(seed=gold["seed"])
f, l = next(O.batches(feats, labels, gold["batch_size"]))
f, l = torch.from_numpy(np.ascontiguousarray(f)).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)
model = whisper.create_whisper_model("small", device=dev, precision="bf16")
model.arena.load_ref(params)
model.refresh_shadows()


def snap():
    out = {}
    for k, t in model.ws.items():
        if not torch.is_tensor(t) or t.numel() == 0:
            continue
        b = t.contiguous().view(torch.uint8).reshape(-1).to(torch.int64)
        w = torch.arange(1, 1025, device=b.device, dtype=torch.int64)
        n = b.numel() // 1024 * 1024
        out[k] = int((b[:n].reshape(-1, 1024) * w).sum().item()) + int(b[n:].sum().item())
    return out


model.forward_backward(f, l); torch.cuda.synchronize(); s0 = snap()
for rep in range(3):
    model.forward_backward(f, l); torch.cuda.synchronize(); s1 = snap()
    diff = [k for k in s0 if s0[k] != s1[k]]
    print(f"rep {rep}: {len(diff)} of {len(s0)} buffers differ; same: {[k for k in s0 if s0[k] == s1[k]][:60]}")
    print("   differ:", diff[:80])
