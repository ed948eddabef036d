"""Time tmi_layernorm_fwd / _bwd on the step's shapes (bf16 [12000, 768] and [800, 768])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for rows, C, dt in ((12000, 768, torch.bfloat16), (800, 768, torch.bfloat16), (12000, 768, torch.float32)):
    x = torch.randn(rows, C, device=dev).to(dt); dy = torch.randn(rows, C, device=dev).to(dt)
    y = torch.empty_like(x); dx = torch.zeros_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    es = x.element_size()
    f = t(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-5))
    bw = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate_dx=False))
    bwa = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate_dx=True))
    n = rows * C * es
    print(f"rows={rows} C={C} {str(dt)[6:]:9s} fwd {f:6.1f} us {2 * n / f * 1e-6:5.2f} TB/s | bwd {bw:6.1f} us {3 * n / bw * 1e-6:5.2f} TB/s"
          f" | bwd+acc {bwa:6.1f} us {4 * n / bwa * 1e-6:5.2f} TB/s")


# Round 3: the partial-row + fold form against the atomic form, alone and beside a GEMM stream that saturates L2
# (the configuration the step runs in: weight gradients on the second stream)
print("\nLayerNorm backward (emit form, bf16 [12000, 768]): workspace partials + fold vs fp32 atomics")
rows, C = 12000, 768
x = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn(rows, C, device=dev).bfloat16()
y = torch.empty_like(x); dx = torch.zeros_like(x); masked = torch.empty_like(x)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
ops.layernorm_fwd(x, g, b, y, mean, rstd, 1e-5)
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); cs = torch.zeros(C, device=dev)
A = torch.randn(12000, 3072, device=dev).bfloat16(); Bm = torch.randn(12000, 768, device=dev).bfloat16()
Cw = torch.zeros(3072, 768, device=dev)
side = torch.cuda.Stream()


def bwd():
    ops.layernorm_bwd_emit(dy, x, g, mean, rstd, dx, dg, db, cs, masked=masked, dropout_p=0.1, dropout_seed=3)


def beside(fn, iters=30):
    """fn on the main stream while weight-gradient GEMMs run back to back on a second stream."""
    torch.cuda.synchronize()
    prev = ops.set_stream(side.cuda_stream)
    for _ in range(3 * iters):
        ops.gemm(A, Bm, Cw, 3072, 768, 12000, 1, 3072, 768, 1, 768, splitk=0)
    ops.set_stream(prev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for atomic in (False, True, False, True):
    ops.LN_ATOMIC = atomic
    print(f"atomic={int(atomic)}: alone {t(bwd):6.1f} us   beside wgrad GEMMs {beside(bwd):6.1f} us", flush=True)
ops.LN_ATOMIC = False
