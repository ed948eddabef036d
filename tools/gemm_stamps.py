import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd
from tethys_speech_amd import ops, _lib
dev = "cuda:0"; bf = torch.bfloat16
h = _lib.lib()
fn = ctypes.CDLL(_lib.LIB_PATH).tmi_debug_gemm_stamps
for (M, N, K) in ((800, 768, 3072), (12000, 768, 3072), (12000, 3072, 768)):
    X = torch.randn(M, K, device=dev).to(bf); Wt = torch.randn(N, K, device=dev).to(bf); Y = torch.empty(M, N, device=dev, dtype=bf)
    for _ in range(3):
        ops.gemm(X, Wt, Y, M, N, K, K, 1, 1, K, N)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 5)()
    fn(out)
    n = max(1, out[4])
    print(f"M{M} N{N} K{K} cfg={os.environ.get('TMI_GEMM_CFG')}: per-iteration cycles: stage-issue {out[0]/n:.0f}  mma {out[1]/n:.0f}  vmcnt-wait {out[2]/n:.0f}  barrier {out[3]/n:.0f}  (iters {n})")
