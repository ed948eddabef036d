"""Time tmi_layernorm_fwd / _bwd on the step's shapes (bf16 [12000, 768] and [800, 768])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for rows, C, dt in ((12000, 768, torch.bfloat16), (800, 768, torch.bfloat16), (12000, 768, torch.float32)):
    x = torch.randn(rows, C, device=dev).to(dt); dy = torch.randn(rows, C, device=dev).to(dt)
    y = torch.empty_like(x); dx = torch.zeros_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    es = x.element_size()
    f = t(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-5))
    bw = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate_dx=False))
    bwa = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate_dx=True))
    n = rows * C * es
    print(f"rows={rows} C={C} {str(dt)[6:]:9s} fwd {f:6.1f} us {2 * n / f * 1e-6:5.2f} TB/s | bwd {bw:6.1f} us {3 * n / bw * 1e-6:5.2f} TB/s"
          f" | bwd+acc {bwa:6.1f} us {4 * n / bwa * 1e-6:5.2f} TB/s")
