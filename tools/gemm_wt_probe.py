"""Forward Dense GEMMs with their real epilogues: weights in the natural Keras layout [K, N] (k-strided B operand) against a
transposed copy [N, K] (k-contiguous B).  Decides whether a transposed bf16 mirror of the weights pays for its upkeep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


d, ff = 768, 3072
tot_n = tot_t = 0.0
for M, mult in ((12000, 4), (800, 4)):
    x = torch.randn(M, d, device=dev).to(bf); h = torch.randn(M, ff, device=dev).to(bf)
    u = torch.empty(M, ff, device=dev, dtype=bf); g = torch.empty(M, ff, device=dev, dtype=bf)
    y = torch.empty(M, d, device=dev, dtype=bf); qkv = torch.empty(M, 3 * d, device=dev, dtype=bf)
    W1 = (torch.randn(d, ff, device=dev) * 0.03).to(bf); W2 = (torch.randn(ff, d, device=dev) * 0.03).to(bf)
    Wq = (torch.randn(d, 3 * d, device=dev) * 0.03).to(bf); Wo = (torch.randn(d, d, device=dev) * 0.03).to(bf)
    W1t, W2t, Wqt, Wot = (w.t().contiguous() for w in (W1, W2, Wq, Wo))
    b1, b2, bq = torch.zeros(ff, device=dev), torch.zeros(d, device=dev), torch.zeros(3 * d, device=dev)
    drop = dict(dropout_p=0.1, dropout_seed=1234)
    cases = [
        ("fc1 +bias+gelu+aux", lambda: ops.gemm(x, W1, g, M, ff, d, d, 1, ff, 1, ff, bias=b1, act=1, aux_out=u),
                               lambda: ops.gemm(x, W1t, g, M, ff, d, d, 1, 1, d, ff, bias=b1, act=1, aux_out=u)),
        ("fc2 +bias+drop+resid", lambda: ops.gemm(h, W2, y, M, d, ff, ff, 1, d, 1, d, bias=b2, resid=x, r_ld=d, **drop),
                                 lambda: ops.gemm(h, W2t, y, M, d, ff, ff, 1, 1, ff, d, bias=b2, resid=x, r_ld=d, **drop)),
        ("qkv +bias+scale", lambda: ops.gemm(x, Wq, qkv, M, 3 * d, d, d, 1, 3 * d, 1, 3 * d, bias=bq, scale_cols=d, scale=0.125),
                            lambda: ops.gemm(x, Wqt, qkv, M, 3 * d, d, d, 1, 1, d, 3 * d, bias=bq, scale_cols=d, scale=0.125)),
        ("out +bias+drop+resid", lambda: ops.gemm(x, Wo, y, M, d, d, d, 1, d, 1, d, bias=b2, resid=x, r_ld=d, **drop),
                                 lambda: ops.gemm(x, Wot, y, M, d, d, d, 1, 1, d, d, bias=b2, resid=x, r_ld=d, **drop)),
    ]
    for name, fn_n, fn_t in cases:
        a, b = timed(fn_n), timed(fn_t)
        tot_n += a * mult; tot_t += b * mult
        print(f"M={M:6d} {name:24s} natural {a:7.1f} us   transposed {b:7.1f} us", flush=True)
M, V = 800, 51904
xd = torch.randn(M, d, device=dev).to(bf)
Wl = (torch.randn(d, V, device=dev) * 0.03).to(bf); Wlt = Wl.t().contiguous()
lg = torch.empty(M, V, device=dev, dtype=bf)
a = timed(lambda: ops.gemm(xd, Wl, lg, M, V, d, d, 1, V, 1, V)); b = timed(lambda: ops.gemm(xd, Wlt, lg, M, V, d, d, 1, 1, d, V))
print(f"M={M:6d} {'lm head':24s} natural {a:7.1f} us   transposed {b:7.1f} us")
tot_n += a; tot_t += b
print(f"per step (4 encoder + 4 decoder layers of these + LM head): natural {tot_n / 1e3:.3f} ms, transposed {tot_t / 1e3:.3f} ms")
