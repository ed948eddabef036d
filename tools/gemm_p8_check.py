"""Correctness + timing of the eight-phase 256x256 (KC, KC) kernel (TMI_GEMM_CFG=10) against torch fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = "cuda:0"
bf = torch.bfloat16
torch.manual_seed(0)


def run(M, N, K, iters=20, check=True):
    A = torch.randn(M, K, device=dev).to(bf)            # dY [M, K]   (k contiguous)
    Bt = (torch.randn(N, K, device=dev) * 0.1).to(bf)   # W  [N, K]   (k contiguous): C = A @ Bt^T
    C = torch.empty(M, N, device=dev, dtype=bf)
    fn = lambda: ops.gemm(A, Bt, C, M, N, K, K, 1, 1, K, N)
    fn(); torch.cuda.synchronize()
    err = 0.0
    if check:
        ref = A.float() @ Bt.float().t()
        err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us {2.0 * M * N * K / us * 1e-6:7.1f} TF/s  rel.err {err:.2e}", flush=True)
    return err


bad = 0
for (M, N, K) in ((256, 256, 128), (256, 256, 192), (300, 500, 256), (1000, 260, 832), (12000, 768, 3072), (12000, 3072, 768),
                  (12000, 768, 768), (12000, 768, 2304), (4096, 4096, 4096), (8192, 8192, 8192)):
    e = run(M, N, K, check=(M * N <= 12000 * 3072 + 1) or M == 4096)
    bad += e > 2e-2
print("FAIL" if bad else "OK")

print("---- forward layout: B [K, N] (k-strided), bias + GELU + aux")
bad = 0
for (M, N, K) in ((300, 500, 256), (1000, 264, 832), (12000, 3072, 768), (12000, 768, 3072), (12000, 768, 768), (12000, 2304, 768),
                  (800, 51904, 768), (4096, 4096, 4096)):
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(K, N, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev) * 0.1
    C = torch.empty(M, N, device=dev, dtype=bf); U = torch.empty(M, N, device=dev, dtype=bf)
    fn = lambda: ops.gemm(A, W, C, M, N, K, K, 1, N, 1, N, bias=bias, act=1, aux_out=U)
    fn(); torch.cuda.synchronize()
    err = eu = 0.0
    if M * N <= 12000 * 3072 + 1:
        pre = A.float() @ W.float() + bias
        ref = torch.nn.functional.gelu(pre)
        err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
        eu = ((U.float() - pre).abs().max() / pre.abs().max()).item()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us {2.0 * M * N * K / us * 1e-6:7.1f} TF/s  rel.err {err:.2e} aux {eu:.2e}", flush=True)
    bad += (err > 2e-2) + (eu > 2e-2)
print("FAIL" if bad else "OK")
