"""Time tmi_layernorm_fwd / _bwd / _bwd_emit alone on the step's shapes (bf16): us and GB/s of algorithmic bytes
(forward 2 passes, backward 3, emit + masked copy 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=50):
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


def run(rows, C, dtype=torch.bfloat16):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(rows, C, device=dev, generator=g).to(dtype)
    dy = torch.randn(rows, C, device=dev, generator=g).to(dtype)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x); masked = torch.empty_like(x)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); cs = torch.zeros(C, device=dev)
    es = x.element_size()
    n = rows * C * es
    t = timeit(lambda: ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, 1e-5))
    print(f"[{rows:6d},{C:5d}] ln_fwd            {t:7.1f} us  {2 * n / t * 1e-3:7.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db))
    print(f"[{rows:6d},{C:5d}] ln_bwd            {t:7.1f} us  {3 * n / t * 1e-3:7.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, accumulate_dx=True))
    print(f"[{rows:6d},{C:5d}] ln_bwd (+= dx)    {t:7.1f} us  {4 * n / t * 1e-3:7.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd_emit(dy, x, gamma, mean, rstd, dx, dg, db, cs, accumulate_dx=True))
    print(f"[{rows:6d},{C:5d}] ln_bwd_emit (+=)  {t:7.1f} us  {4 * n / t * 1e-3:7.0f} GB/s")
    t = timeit(lambda: ops.layernorm_bwd_emit(dy, x, gamma, mean, rstd, dx, dg, db, cs, masked=masked, dropout_p=0.1, dropout_seed=5,
                                              accumulate_dx=True))
    print(f"[{rows:6d},{C:5d}] ln_bwd_emit+mask  {t:7.1f} us  {5 * n / t * 1e-3:7.0f} GB/s")


for rows, C in ((12000, 768), (800, 768), (12000, 1280)):
    run(rows, C)
