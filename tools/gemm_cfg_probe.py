"""One GEMM shape under the library's choice and under forced configurations (TMI_GEMM_CFG is read once per process: run per value).
usage: gemm_cfg_probe.py M N K nn|nt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
M, N, K, lay = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev, bf = "cuda:0", torch.bfloat16
A = torch.randn(M, K, device=dev).to(bf)
C = torch.empty(M, N, device=dev, dtype=bf)
if lay == "nn":
    B = (torch.randn(K, N, device=dev) * 0.03).to(bf)
    fn = lambda: ops.gemm(A, B, C, M, N, K, K, 1, N, 1, N)
else:
    B = (torch.randn(N, K, device=dev) * 0.03).to(bf)
    fn = lambda: ops.gemm(A, B, C, M, N, K, K, 1, 1, K, N)
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print(f"cfg={os.environ.get('TMI_GEMM_CFG', 'auto'):>4s} ({M},{N},{K}) {lay}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
