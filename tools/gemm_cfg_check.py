"""Correctness of whatever tile configuration TMI_GEMM_CFG forces, against torch fp32, on the step's activation x weight shapes
(forward: k-strided B; dgrad: k-contiguous B), with bias + residual epilogue and ragged edges."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
torch.manual_seed(0)
worst = 0.0
for (M, N, K) in ((12000, 768, 768), (12000, 768, 2304), (12000, 3072, 768), (1000, 520, 192), (161, 300, 64), (95, 256, 128)):
    X = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(K, N, device=dev) * 0.05).to(bf); Wt = W.t().contiguous()
    bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).to(bf)
    ref = X.float() @ W.float() + bias + R.float()
    for name, Bm, bsk, bsn in (("fwd", W, N, 1), ("dgrad", Wt, 1, K)):
        Y = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        ops.gemm(X, Bm, Y, M, N, K, K, 1, bsk, bsn, N, bias=bias, resid=R, r_ld=N)
        err = ((Y.float() - ref).abs().max() / ref.abs().max()).item()
        worst = max(worst, err)
        print(f"cfg={os.environ.get('TMI_GEMM_CFG', 'auto')} {name} M{M} N{N} K{K}: rel err {err:.2e}")
assert worst < 1e-2, worst
print("ok")
