"""Runs a few tmi_gemm shapes a fixed number of times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd
from tethys_speech_amd import ops
dev = "cuda:0"; bf = torch.bfloat16
M = 12000
for (N, K) in ((3072, 768), (768, 3072)):
    X = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(K, N, device=dev) * 0.05).to(bf)
    Y = torch.empty(M, N, device=dev, dtype=bf); dY = torch.randn(M, N, device=dev).to(bf)
    dX = torch.empty(M, K, device=dev, dtype=bf)
    for _ in range(5):
        ops.gemm(X, W, Y, M, N, K, K, 1, N, 1, N)
        ops.gemm(dY, W, dX, M, K, N, N, 1, 1, N, K)
torch.cuda.synchronize()
