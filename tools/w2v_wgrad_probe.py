"""The Wav2Vec2 encoder's batched weight gradients as wav2vec2.py issues them (three layers per launch, K = B*T = 792 tokens,
fp32 output, library-chosen split): dW[l][Kin, N] = X[l]^T dY[l].  Time per launch and error; library switches are read once
per process (TMI_GEMM_CFG=12 / 13 / 15 force the small-tile configurations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
tag = os.environ.get("PROBE_TAG", "default")


def timed(fn, iters=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


R, L = 792, 3
for Kin, N in ((768, 768), (768, 3072), (3072, 768)):
    X = torch.randn(L, R, Kin, device=dev).to(bf); dY = torch.randn(L, R, N, device=dev).to(bf)
    dW = torch.zeros(L, Kin, N, device=dev, dtype=torch.float32)
    f = lambda: ops.gemm(X, dY, dW, Kin, N, R, 1, Kin, N, 1, N, nbatch=L, a_sb=R * Kin, b_sb=R * N, c_sb=Kin * N, splitk=0)
    dW.zero_(); f(); torch.cuda.synchronize()
    ref = torch.einsum("lrk,lrn->lkn", X.float(), dY.float())
    err = float((dW - ref).abs().max() / ref.abs().max())
    print(f"{tag:24s} dW[{L}][{Kin:4d},{N:4d}] K={R}: {timed(f):6.1f} us  err {err:.1e}", flush=True)
