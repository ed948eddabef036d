"""Per-tensor relative L2 error of the bf16 path's gradients against the fp32 parity path (which holds the fp64 oracle to
2e-5) on the headline model and batch (small-ref, B = 8, step 0 of the golden run, dropout 0)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper
from oracle import whisper_oracle as O
dev = "cuda:0"
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "whisper_small_ref_b8_10steps.json")))
ocfg = O.make_config("small")
params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
feats, labels = O.create_dummy_pool(seed=gold["seed"])
f, l = next(O.batches(feats, labels, gold["batch_size"]))
f, l = torch.from_numpy(np.ascontiguousarray(f)).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)
grads = {}
for prec in ("fp32", "bf16"):
    model = whisper.create_whisper_model("small", device=dev, precision=prec)
    model.arena.load_ref(params)
    model.refresh_shadows()
    loss = float(model.forward_backward(f, l).item())
    torch.cuda.synchronize()
    grads[prec] = {k: v.double().cpu() for k, v in model.arena.ref_views(model.arena.g).items()}
    print(prec, "loss", loss)
    del model
    torch.cuda.empty_cache()
rows = []
for k, g in grads["fp32"].items():
    e = float((grads["bf16"][k] - g).norm() / max(float(g.norm()), 1e-30))
    rows.append((e, k, float(g.norm())))
tot = sum(float((grads["bf16"][k] - g).norm() ** 2) for k, g in grads["fp32"].items()) ** 0.5 / sum(float(g.norm() ** 2) for g in grads["fp32"].values()) ** 0.5
print(f"whole gradient rel L2 {tot:.3e}")
for e, k, n in sorted(rows, reverse=True)[:25]:
    print(f"{e:.3e}  |g| {n:.3e}  {k}")
print("...")
for e, k, n in sorted(rows)[:8]:
    print(f"{e:.3e}  |g| {n:.3e}  {k}")
