"""Whisper-large's decoder-sized GEMMs (M = 800, d_model 1280, d_ff 5120) per tile configuration (TMI_GEMM_CFG forces one;
unset: the library's rules).  usage: [TMI_GEMM_CFG=n] gemm_large_dec_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev = "cuda:0"; bf = torch.bfloat16


def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


M = int(os.environ.get("PROBE_M", "800"))
out = []
SHAPES = ((1280, 1280), (1280, 5120), (3840, 1280), (5120, 1280))
if os.environ.get("PROBE_SHAPES"):   # "N,K;N,K;..."
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in os.environ["PROBE_SHAPES"].split(";"))
for N, K in SHAPES:
    X = torch.randn(M, K, device=dev).to(bf); Wt = torch.randn(N, K, device=dev).to(bf); W = torch.randn(K, N, device=dev).to(bf)
    Y = torch.empty(M, N, device=dev, dtype=bf)
    a = bench(lambda: ops.gemm(X, W, Y, M, N, K, K, 1, N, 1, N))      # forward: weights k-strided
    b = bench(lambda: ops.gemm(X, Wt, Y, M, N, K, K, 1, 1, K, N))     # dgrad: weights k-contiguous
    out.append(f"N{N} K{K}: fwd {a:5.1f} dgrad {b:5.1f}")
print(f"cfg={os.environ.get('TMI_GEMM_CFG', 'rules'):>5s} M{M} | " + " | ".join(out), flush=True)
