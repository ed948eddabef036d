"""Latency of decoder-sized GEMMs vs K (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd
from tethys_speech_amd import ops
dev = "cuda:0"; bf = torch.bfloat16

def bench(name, fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) * 1e3 / iters:8.1f} us", flush=True)

for M in (800, 128):
    for N in (768, 3072):
        for K in (64, 768, 1536, 3072, 6144):
            X = torch.randn(M, K, device=dev).to(bf); Wt = torch.randn(N, K, device=dev).to(bf); W = torch.randn(K, N, device=dev).to(bf)
            Y = torch.empty(M, N, device=dev, dtype=bf)
            bench(f"KC,KC M{M} N{N} K{K}", lambda: ops.gemm(X, Wt, Y, M, N, K, K, 1, 1, K, N))
            bench(f"KC,KS M{M} N{N} K{K}", lambda: ops.gemm(X, W, Y, M, N, K, K, 1, N, 1, N))
