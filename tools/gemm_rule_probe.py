"""Times the layer GEMM forms (with their epilogues) of a given width, to tune the tile-selection rules:
run once per TMI_GEMM_CFG value (unset = library choice, 10 = eight-phase 256x256, 5 = 16-wave 256x256, 4 = 128x128).
usage: gemm_rule_probe.py d_model d_ff [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops

dev, bf = "cuda:0", torch.bfloat16
d, ff = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 12000


def timed(name, fn, flops, iters=20):
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
    print(f"cfg={os.environ.get('TMI_GEMM_CFG', '-'):>3s} d{d} {name:34s} {us:8.1f} us {flops / us / 1e6:7.1f} TF/s", flush=True)


x = torch.randn(M, d, device=dev).to(bf)
h = torch.randn(M, ff, device=dev).to(bf)
u = torch.randn(M, ff, device=dev).to(bf)
W1 = (torch.randn(d, ff, device=dev) * 0.03).to(bf)
W2 = (torch.randn(ff, d, device=dev) * 0.03).to(bf)
Wq = (torch.randn(d, 3 * d, device=dev) * 0.03).to(bf)
Wo = (torch.randn(d, d, device=dev) * 0.03).to(bf)
b1, b2, bq = torch.zeros(ff, device=dev), torch.zeros(d, device=dev), torch.zeros(3 * d, device=dev)
y = torch.empty(M, d, device=dev, dtype=bf)
qkv = torch.empty(M, 3 * d, device=dev, dtype=bf)
g = torch.empty(M, ff, device=dev, dtype=bf)
f1 = 2.0 * M * d * ff
timed("fc1 fwd  (KC,KS) +bias+gelu+aux", lambda: ops.gemm(x, W1, g, M, ff, d, d, 1, ff, 1, ff, bias=b1, act=1, aux_out=u), f1)
timed("fc2 fwd  (KC,KS) +bias+resid", lambda: ops.gemm(h, W2, y, M, d, ff, ff, 1, d, 1, d, bias=b2, resid=x, r_ld=d), f1)
timed("fc2 dgrad(KC,KC) *gelu'(aux)", lambda: ops.gemm(x, W2, g, M, ff, d, d, 1, 1, d, ff, aux_in=u), f1)
timed("fc1 dgrad(KC,KC)", lambda: ops.gemm(h, W1, y, M, d, ff, ff, 1, 1, ff, d), f1)
timed("qkv fwd  (KC,KS) +bias+scale", lambda: ops.gemm(x, Wq, qkv, M, 3 * d, d, d, 1, 3 * d, 1, 3 * d, bias=bq, scale_cols=d, scale=0.125), 2.0 * M * d * 3 * d)
timed("qkv dgrad(KC,KC)", lambda: ops.gemm(qkv, Wq, y, M, d, 3 * d, 3 * d, 1, 1, 3 * d, d), 2.0 * M * d * 3 * d)
timed("out fwd  (KC,KS) +bias+resid", lambda: ops.gemm(x, Wo, y, M, d, d, d, 1, d, 1, d, bias=b2, resid=x, r_ld=d), 2.0 * M * d * d)
timed("out dgrad(KC,KC)", lambda: ops.gemm(x, Wo, y, M, d, d, d, 1, 1, d, d), 2.0 * M * d * d)
