"""The fp32 (parity-mode) GEMM on the step's shapes: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), roof 157 TF/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev = "cuda:0"


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def row(name, us, fl):
    print(f"{name:44s} {us:9.1f} us {fl / us / 1e6:7.1f} TF/s", flush=True)


for name, M, N, K, layout in (("fc1 fwd x[M,K] W[K,N]", 12000, 3072, 768, "nn"), ("fc2 fwd", 12000, 768, 3072, "nn"),
                              ("fc2 dgrad dy[M,K] W[N,K]^T", 12000, 3072, 768, "nt"), ("fc wgrad x[K,M]^T dy[K,N]", 768, 3072, 12000, "tn"),
                              ("dec fc1 fwd", 800, 3072, 768, "nn"), ("square", 4096, 4096, 4096, "nt")):
    if layout == "nn":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(A, B, C, M, N, K, K, 1, N, 1, N)
    elif layout == "nt":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(A, B, C, M, N, K, K, 1, 1, K, N)
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
        f = lambda: ops.gemm(A, B, C, M, N, K, 1, M, N, 1, N, splitk=0)
    row(f"{name} ({M},{N},{K})", timed(f), 2.0 * M * N * K)
# attention as the fp32 path runs it: per batch row, heads as the GEMM batch
T, H, hd, d = 1500, 12, 64, 768
qkv = torch.randn(T, 3 * d, device=dev); P = torch.empty(H, T, T, device=dev); ctx = torch.empty(T, d, device=dev)
f = lambda: ops.gemm(qkv, qkv, P, T, T, hd, 3 * d, 1, 1, 3 * d, T, nbatch=H, a_sb=hd, b_sb=hd, c_sb=T * T, b_off=d)
row("scores q.k^T (1500,1500,64) x12 heads", timed(f), 2.0 * T * T * hd * H)
f = lambda: ops.gemm(P, qkv, ctx, T, hd, T, T, 1, 3 * d, 1, d, nbatch=H, a_sb=T * T, b_sb=hd, c_sb=hd, b_off=2 * d)
row("probs.v (1500,64,1500) x12 heads", timed(f), 2.0 * T * T * hd * H)
