"""The step's GEMM shapes (Whisper small-ref, B = 8): tmi_gemm (plain epilogue) beside the vendor library through torch.matmul
(hipBLASLt / rocBLAS), bf16 in and out, fp32 accumulate.  Diagnostic: how far are the hand-written kernels from what the
library reaches on the SAME shapes (not on 8192^3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16


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


print(f"{'shape (M, N, K) layout':44s} {'tmi us':>8s} {'TF/s':>7s} {'lib us':>8s} {'TF/s':>7s}")
for name, M, N, K, layout in [("enc fc1 fwd   x[M,K] W[K,N]", 12000, 3072, 768, "nn"), ("enc fc2 fwd   x[M,K] W[K,N]", 12000, 768, 3072, "nn"),
                              ("enc qkv fwd   x[M,K] W[K,N]", 12000, 2304, 768, "nn"), ("enc out fwd   x[M,K] W[K,N]", 12000, 768, 768, "nn"),
                              ("enc fc2 dgrad dy[M,K] W[N,K]^T", 12000, 3072, 768, "nt"), ("enc fc1 dgrad dy[M,K] W[N,K]^T", 12000, 768, 3072, "nt"),
                              ("enc qkv dgrad dy[M,K] W[N,K]^T", 12000, 768, 2304, "nt"), ("enc out dgrad dy[M,K] W[N,K]^T", 12000, 768, 768, "nt"),
                              ("enc fc wgrad  x[K,M]^T dy[K,N]", 768, 3072, 12000, "tn"), ("enc qkv wgrad x[K,M]^T dy[K,N]", 768, 2304, 12000, "tn"),
                              ("enc out wgrad x[K,M]^T dy[K,N]", 768, 768, 12000, "tn"),
                              ("dec fc1 fwd   x[M,K] W[K,N]", 800, 3072, 768, "nn"), ("dec fc2 fwd   x[M,K] W[K,N]", 800, 768, 3072, "nn"),
                              ("dec out fwd   x[M,K] W[K,N]", 800, 768, 768, "nn"), ("lm head fwd   x[M,K] W[K,N]", 800, 51904, 768, "nn")]:
    if layout == "nn":
        A = torch.randn(M, K, device=dev).to(bf); Bm = (torch.randn(K, N, device=dev) * 0.03).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        f_t = lambda: ops.gemm(A, Bm, C, M, N, K, K, 1, N, 1, N)
        f_l = lambda: torch.matmul(A, Bm, out=C)
    elif layout == "nt":
        A = torch.randn(M, K, device=dev).to(bf); Bt = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        f_t = lambda: ops.gemm(A, Bt, C, M, N, K, K, 1, 1, K, N)
        f_l = lambda: torch.matmul(A, Bt.t(), out=C)
    else:  # tn: C[M,N] = X^T dY with X [K,M], dY [K,N]; fp32 output as the weight gradients are
        X = torch.randn(K, M, device=dev).to(bf); dY = torch.randn(K, N, device=dev).to(bf)
        C = torch.zeros(M, N, device=dev, dtype=torch.float32)
        Cb = torch.empty(M, N, device=dev, dtype=bf)
        f_t = lambda: ops.gemm(X, dY, C, M, N, K, 1, M, N, 1, N, splitk=0)
        f_l = lambda: torch.matmul(X.t(), dY, out=Cb)   # (the library writes bf16 here: an easier output than fp32)
    t, l = timed(f_t), timed(f_l)
    fl = 2.0 * M * N * K
    print(f"{name:32s} {str((M, N, K)):>20s} {t:8.1f} {fl / t / 1e6:7.1f} {l:8.1f} {fl / l / 1e6:7.1f}", flush=True)
