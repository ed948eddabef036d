"""Fixed (K-independent) cost of a GEMM config: time vs K at the step's M, N.  TMI_GEMM_CFG selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops
dev = "cuda:0"; bf = torch.bfloat16
def t(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M = 12000
for N in (768, 2304, 3072):
    row = []
    for K in (128, 384, 768, 1536, 3072):
        A = torch.randn(M, K, device=dev).to(bf); Wt = torch.randn(N, K, device=dev).to(bf); C = torch.empty(M, N, device=dev, dtype=bf)
        row.append(f"K{K}:{t(lambda: ops.gemm(A, Wt, C, M, N, K, K, 1, 1, K, N)):6.1f}")
    print(f"KC,KC M{M} N{N}  " + "  ".join(row), flush=True)
