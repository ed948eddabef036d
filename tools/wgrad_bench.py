"""Weight-gradient GEMMs dW[K_in, N] = X[T, K_in]^T dY[T, N] (fp32 out) against the split-K count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = "cuda:0"
bf = torch.bfloat16


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


splits = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6, 8, 12, 16]
for T in (12000, 800):
    for (Kin, N) in ((768, 3072), (3072, 768), (768, 2304), (768, 768), (768, 1536)):
        X = torch.randn(T, Kin, device=dev).to(bf); dY = torch.randn(T, N, device=dev).to(bf)
        dW = torch.zeros(Kin, N, device=dev, dtype=torch.float32)
        fl = 2.0 * T * Kin * N
        row = []
        ref = X.float().t() @ dY.float()
        for sk in splits:
            dW.zero_()
            ops.gemm(X, dY, dW, Kin, N, T, 1, Kin, N, 1, N, splitk=sk)
            err = ((dW - ref).abs().max() / ref.abs().max()).item()
            us = t(lambda: ops.gemm(X, dY, dW, Kin, N, T, 1, Kin, N, 1, N, splitk=sk))
            row.append(f"s{sk}:{us:6.1f}" + ("" if err < 1e-4 else f" ERR {err:.1e}"))
        print(f"T={T:5d} dW[{Kin:4d},{N:4d}] {fl * 1e-9:5.1f} GF  " + " ".join(row), flush=True)
