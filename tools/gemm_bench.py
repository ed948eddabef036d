"""Micro-benchmark of tmi_gemm shapes from the Whisper small-ref step (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd
from tethys_speech_amd import ops

dev = "cuda:0"
bf = torch.bfloat16


def bench(name, fn, flops, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:46s} {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s", flush=True)


def main():
    M = 12000
    for (N, K) in ((3072, 768), (768, 3072), (768, 768), (2304, 768)):
        X = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(K, N, device=dev) * 0.05).to(bf)      # natural [in,out]
        Y = torch.empty(M, N, device=dev, dtype=bf)
        dY = torch.randn(M, N, device=dev).to(bf)
        dX = torch.empty(M, K, device=dev, dtype=bf)
        dW = torch.zeros(K, N, device=dev, dtype=torch.float32)
        bias = torch.zeros(N, device=dev)
        fl = 2.0 * M * N * K
        bench(f"fwd  (KC,KS) M{M} N{N} K{K}", lambda: ops.gemm(X, W, Y, M, N, K, K, 1, N, 1, N, bias=bias), fl)
        bench(f"fwd+gelu+aux   M{M} N{N} K{K}", lambda: ops.gemm(X, W, Y, M, N, K, K, 1, N, 1, N, bias=bias, act=1, aux_out=dY), fl)
        bench(f"dgrad(KC,KC) M{M} N{K} K{N}", lambda: ops.gemm(dY, W, dX, M, K, N, N, 1, 1, N, K), fl)
        bench(f"wgrad(KS,KS) M{K} N{N} K{M}", lambda: ops.gemm(X, dY, dW, K, N, M, 1, K, N, 1, N, splitk=0), fl)
    # decoder-sized
    M = 800
    X = torch.randn(M, 768, device=dev).to(bf); W = torch.randn(768, 768, device=dev).to(bf); Y = torch.empty(M, 768, device=dev, dtype=bf)
    bench("fwd small M800 N768 K768", lambda: ops.gemm(X, W, Y, M, 768, 768, 768, 1, 768, 1, 768), 2.0 * M * 768 * 768)


if __name__ == "__main__":
    main()
