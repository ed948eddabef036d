"""Time tmi_colsum (bias gradients) on the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")
for rows, N in ((12000, 768), (12000, 2304), (12000, 3072), (12000, 1536), (800, 768), (800, 3072)):
    dy = torch.randn(rows, N, device=dev).to(torch.bfloat16)
    out = torch.zeros(N, device=dev)
    fn = lambda: ops.bias_grad(dy, out)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    out.zero_(); fn(); torch.cuda.synchronize()
    err = (out - dy.float().sum(0)).abs().max().item()
    print(f"rows={rows:6d} N={N:5d} {us:6.1f} us {rows * N * 2 / us * 1e-6:5.2f} TB/s  maxerr {err:.3e}")
