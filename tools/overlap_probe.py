"""Do an HBM-bound kernel (Adam over the arena) and MFMA-bound GEMMs overlap on two streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16
n = 147_781_632
p = torch.zeros(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
mir = torch.zeros(n, device=dev, dtype=bf)
M, N, K = 12000, 3072, 768
A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(K, N, device=dev) * 0.05).to(bf); Cm = torch.empty(M, N, device=dev, dtype=bf)
A2 = torch.randn(M, N, device=dev).to(bf); W2 = (torch.randn(N, K, device=dev) * 0.05).to(bf); C2 = torch.empty(M, K, device=dev, dtype=bf)
side = torch.cuda.Stream(device=dev)


def gemms(reps):
    for _ in range(reps):
        ops.gemm(A, W, Cm, M, N, K, K, 1, N, 1, N)          # fwd fc1 shape
        ops.gemm(A2, W2, C2, M, K, N, N, 1, K, 1, K)        # fwd fc2 shape (p8)


def adam(reps, chunks=1):
    c = n // chunks // 8 * 8
    for _ in range(reps):
        for i in range(chunks):
            lo = i * c; hi = n if i == chunks - 1 else lo + c
            ops.adam_step(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], hi - lo, 1e-4, 0.9, 0.999, 1e-7, 5, mirror=mir[lo:hi])


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


gemms(2); adam(1); torch.cuda.synchronize()
G = 12
tg = timed(lambda: gemms(G))
ta = timed(lambda: adam(2))
def both(chunks):
    ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
    prev = ops.set_stream(side.cuda_stream)
    adam(2, chunks)
    ops.set_stream(prev)
    gemms(G)
tb = timed(lambda: both(1))
tb16 = timed(lambda: both(16))
print(f"GEMMs alone {tg:.2f} ms, 2x Adam alone {ta:.2f} ms, concurrent {tb:.2f} ms (sum {tg + ta:.2f}, max {max(tg, ta):.2f}); Adam in 16 chunks: {tb16:.2f} ms")
