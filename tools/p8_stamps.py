"""In-kernel stamps of the eight-phase kernel (TMI_GEMM_DBG=16; block 8, wave 0): cycles of prologue / K loop / rejoin + drain /
epilogue / store drain of one tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import _lib, ops
fn = ctypes.CDLL(_lib.LIB_PATH).tmi_debug_gemm_stamps
dev, bf = "cuda:0", torch.bfloat16
for M, N, K in ((768, 3072, 768), (12000, 3072, 768), (12000, 768, 3072)):
    A = torch.randn(M, K, device=dev).to(bf)
    Bt = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    Cm = torch.empty(M, N, device=dev, dtype=bf)
    for _ in range(5):
        ops.gemm(A, Bt, Cm, M, N, K, K, 1, 1, K, N)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 5)()
    fn(out)
    print(f"M {M} N {N} K {K}: prologue {out[0]} loop {out[1]} ({out[1] / (K // 64):.0f}/K-tile) rejoin+drain {out[2]} epilogue {out[3]} store drain {out[4]} cycles")
if os.environ.get("STAMP_WGRAD"):
    # weight-gradient shapes: C[M,N] fp32 = X^T dY, X [K,M], dY [K,N] (both k-strided), split-K through the workspace
    for M, N, K in ((768, 3072, 12000), (768, 2304, 12000), (768, 768, 12000)):
        X = torch.randn(K, M, device=dev).to(bf)
        dY = torch.randn(K, N, device=dev).to(bf)
        Cw = torch.zeros(M, N, device=dev, dtype=torch.float32)
        for _ in range(5):
            ops.gemm(X, dY, Cw, M, N, K, 1, M, N, 1, N, splitk=0)
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 5)()
        fn(out)
        print(f"wgrad M {M} N {N} K {K}: prologue {out[0]} loop {out[1]} rejoin+drain {out[2]} epilogue {out[3]} store drain {out[4]} cycles")
