"""How the un-overlapped epilogue of the eight-phase kernel scales with the number of busy CUs: one tile per workgroup, N = 3072,
K = 768, M chosen for 48 / 96 / 192 / 252 tiles (one round) and the step's 12000 (three rounds).  Run with and without
TMI_GEMM_DBG=1 (no epilogue); TMI_GEMM_CFG=14 forces the 192 x 256 tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
tag = os.environ.get("PROBE_TAG", "")


def timed(fn, iters=50):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


N, K = int(os.environ.get("PROBE_N", 3072)), int(os.environ.get("PROBE_K", 768))
for M in (768, 1536, 3072, 4032, 12000):
    A = torch.randn(M, K, device=dev).to(bf)
    Bt = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    Cm = torch.empty(M, N, device=dev, dtype=bf)
    t = timed(lambda: ops.gemm(A, Bt, Cm, M, N, K, K, 1, 1, K, N))
    tiles = -(-M // 192) * (N // 256)
    print(f"{tag:40s} M {M:6d} tiles(192x256) {tiles:4d}  {t:7.1f} us  C {M * N * 2 / 1e6:6.1f} MB", flush=True)
