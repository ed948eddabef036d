"""Short-K (768), k-strided-weight forward launches that stay on the two-stage 256x256 kernel (CFG 5): the fused qkv projection
[12000, 2304] and the decoder's cross k|v projection of all layers [12000, 6144], with their bias (+ q scale) epilogues.
Library switches are read once per process: run once per setting (TMI_GEMM_P8_ALL=1, TMI_GEMM_CFG=10 / 14)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
tag = os.environ.get("PROBE_TAG", "default")


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


M, K = 12000, 768
A = torch.randn(M, K, device=dev).to(bf)
for name, N, kw in (("qkv fwd bias+scale", 2304, dict(scale_cols=768, scale=0.125)), ("cross k|v fwd bias", 6144, {}), ("fc1 fwd bias+gelu+aux", 3072, dict(act=1))):
    W = (torch.randn(K, N, device=dev) * 0.03).to(bf)
    bias = torch.randn(N, device=dev) * 0.1
    C = torch.empty(M, N, device=dev, dtype=bf)
    if "aux" in name:
        kw = dict(kw, aux_out=torch.empty_like(C))
    f = lambda: ops.gemm(A, W, C, M, N, K, K, 1, N, 1, N, bias=bias, **kw)
    f(); torch.cuda.synchronize()
    ref = A[:512].float() @ W.float() + bias
    if "scale" in name:
        ref[:, :768] *= 0.125
    if "gelu" in name:
        ref = torch.nn.functional.gelu(ref.to(bf).float())
    err = float((C[:512].float() - ref).abs().max())
    t = timed(f)
    print(f"{tag:28s} {name:24s} ({M}, {N}, {K}) {t:7.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s  err {err:.2e}", flush=True)
