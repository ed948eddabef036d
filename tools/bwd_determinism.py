"""Which gradients differ bitwise between two identical bf16 steps (forward is bit-reproducible since the round-4 attention
fix): names the backward kernels whose reductions still depend on the order of fp32 atomics."""
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
gs = []
for rep in range(3):
    model.forward_backward(f, l)
    torch.cuda.synchronize()
    gs.append({k: v.clone() for k, v in model.arena.ref_views(model.arena.g).items()})
from collections import Counter
kinds = Counter()
for k in gs[0]:
    n = max(int((gs[0][k].view(torch.int32) != gs[r][k].view(torch.int32)).sum()) for r in (1, 2))
    if n:
        kind = ".".join(p for p in k.split(".") if not p.isdigit())
        kinds[kind] += 1
        if kinds[kind] == 1:
            rel = float((gs[0][k] - gs[1][k]).norm() / gs[0][k].norm().clamp_min(1e-30))
            print(f"{k}: {n} of {gs[0][k].numel()} elements differ (rel L2 between runs {rel:.1e})")
print("tensor kinds with run-to-run differences:", dict(kinds))
print("kinds without:", sorted(set(".".join(p for p in k.split(".") if not p.isdigit()) for k in gs[0]) - set(kinds)))
