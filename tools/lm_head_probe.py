"""The LM head's three GEMMs with the step's own calls (whisper.py: forward into the padded logits, weight gradient
[d, Vp] from K = B*S rows, dgrad over K = Vp = 51904 split-K into fp32) - time per launch and the max error against torch.
Library switches are read once per process: run once per setting (PROBE_TAG names the line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
tag = os.environ.get("PROBE_TAG", "default")
only = os.environ.get("PROBE_ONLY")


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


torch.manual_seed(0)
R, d, Vp = 800, 768, 51904
x = torch.randn(R, d, device=dev).to(bf)
W = (torch.randn(d, Vp, device=dev) * 0.03).to(bf)          # lm_head.kernel, stored [d, Vp]
logits = torch.empty(R, Vp, device=dev, dtype=bf)
dlog = (torch.randn(R, Vp, device=dev) * 0.01).to(bf)
dW = torch.empty(d, Vp, device=dev, dtype=torch.float32)
dx = torch.zeros(R, d, device=dev, dtype=torch.float32)
cases = {
    "fwd   [800,51904]  K 768": (lambda: ops.gemm(x, W, logits, R, Vp, d, d, 1, Vp, 1, Vp), lambda: (logits.float(), x.float() @ W.float())),
    "wgrad [768,51904]  K 800": (lambda: ops.gemm(x, dlog, dW, d, Vp, R, 1, d, Vp, 1, Vp, splitk=0), lambda: (dW, x.float().t() @ dlog.float())),
    "dgrad [800,768] K 51904": (lambda: ops.gemm(dlog, W, dx, R, d, Vp, Vp, 1, 1, Vp, d, splitk=0), lambda: (dx, dlog.float() @ W.float().t())),
}
for name, (f, ref) in cases.items():
    if only and only not in name:
        continue
    if "dgrad" in name:
        dx.zero_()
    f()
    torch.cuda.synchronize()
    got, want = ref()
    err, scale = float((got - want).abs().max()), float(want.abs().max())
    t = timed(f)
    print(f"{tag:44s} {name:26s} {t:8.1f} us {2.0 * R * d * Vp / t / 1e6:7.1f} TF/s  max|err| {err:.3e} (|ref| {scale:.2f})", flush=True)
