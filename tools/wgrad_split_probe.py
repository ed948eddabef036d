"""Weight-gradient GEMM (x^T dy, fp32 out, both operands k-strided) of the encoder's fc layers, alone on the chip, by split-K cap
(TMI_GEMM_P8_MAXSPLIT is read once per process: run once per value).  Prints time, TF/s and CU-time (workgroups x time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
for M, N, K in ((768, 3072, 12000), (3072, 768, 12000), (768, 2304, 12000), (768, 768, 12000)):
    X = torch.randn(K, M, device=dev).to(bf); dY = torch.randn(K, N, device=dev).to(bf)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32)
    f = lambda: ops.gemm(X, dY, C, M, N, K, 1, M, N, 1, N, splitk=0)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"cap={os.environ.get('TMI_GEMM_P8_MAXSPLIT', 'default')} ({M},{N},{K}) {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
