"""Per-kernel parity: every C-ABI entry point against the oracle's op-level restatement
(oracle/whisper_oracle.py) or a closed-form fp64 expression, on seeded inputs.

Tolerances: fp32 kernels vs fp64 reference: max|err| <= 2e-5 * max|ref| (forward) and
1e-4 (gradients) — SURVEY.md 8(d).  bf16 kernels: inputs are rounded to bf16 first and the
reference is evaluated in fp64 on those rounded inputs, so the tolerance only has to cover
bf16 rounding of the OUTPUT plus fp32 accumulation: 1.5e-2 * max|ref|.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402  (checker only)


def _ops():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import ops
    return ops


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def tol(dtype, grad=False):
    if dtype == torch.bfloat16:
        return 1.5e-2
    return 1e-4 if grad else 2e-5


def rnd(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)
    return x.to(dev)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 136, 96), (37, 51, 29), (300, 768, 768)])
def test_gemm_layouts(dev, dtype, layout, shape):
    ops = _ops()
    M, N, K = shape
    if dtype == torch.bfloat16 and (M, N, K) == (37, 51, 29):
        pass  # scalar path (unaligned) is exercised for bf16 too
    A = rnd((M, K), dtype, dev, 1)
    B = rnd((K, N), dtype, dev, 2)
    ref = A.double() @ B.double()
    Cm = torch.empty((M, N), dtype=dtype, device=dev)
    if layout == "nn":
        ops.gemm(A, B, Cm, M, N, K, K, 1, N, 1, N)
    elif layout == "nt":  # B stored [N,K]
        Bt = B.t().contiguous()
        ops.gemm(A, Bt, Cm, M, N, K, K, 1, 1, K, N)
    else:  # A stored [K,M] (wgrad form), B natural
        At = A.t().contiguous()
        ops.gemm(At, B, Cm, M, N, K, 1, M, N, 1, N)
    torch.cuda.synchronize()
    assert rel_err(Cm, ref) <= (8e-3 if dtype == torch.bfloat16 else 2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_chain(dev, dtype):
    """bias, column scale, aux_out, gelu, residual — in the documented order."""
    ops = _ops()
    M, N, K = 260, 192, 128
    A, B = rnd((M, K), dtype, dev, 3, 0.5), rnd((K, N), dtype, dev, 4, 0.5)
    bias = rnd((N,), torch.float32, dev, 5)
    R = rnd((M, N), dtype, dev, 6)
    Cm = torch.empty((M, N), dtype=dtype, device=dev)
    U = torch.empty_like(Cm)
    ops.gemm(A, B, Cm, M, N, K, K, 1, N, 1, N, bias=bias, scale_cols=64, scale=0.125, act=1, aux_out=U,
             resid=R, r_ld=N)
    torch.cuda.synchronize()
    u = A.double() @ B.double() + bias.double()
    u[:, :64] *= 0.125
    ref = O.gelu_erf(u) + R.double()
    assert rel_err(U, u) <= tol(dtype) * 4
    assert rel_err(Cm, ref) <= tol(dtype) * 4
    # backward-through-GELU epilogue + accumulate
    G = rnd((M, K), dtype, dev, 7, 0.5)
    W = rnd((K, N), dtype, dev, 8, 0.5)
    D = rnd((M, N), dtype, dev, 9)
    D0 = D.clone()
    ops.gemm(G, W, D, M, N, K, K, 1, N, 1, N, accumulate=True, aux_in=U)
    torch.cuda.synchronize()
    uu = U.double()
    gp = 0.5 * (1 + torch.erf(uu / math.sqrt(2))) + uu * torch.exp(-0.5 * uu * uu) / math.sqrt(2 * math.pi)
    ref2 = (G.double() @ W.double() + D0.double()) * gp
    assert rel_err(D, ref2) <= tol(dtype) * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_conv_addressing(dev, dtype):
    """Conv1D "same" as a GEMM over overlapping rows of a channels-last padded buffer,
    checked against the oracle's conv1d_same (W:311-312 semantics), strides 1 and 2."""
    ops = _ops()
    Bn, T, Cin, Cout, k = 3, 50, 16, 40, 3
    x = rnd((Bn, T, Cin), dtype, dev, 10)
    w = rnd((k, Cin, Cout), dtype, dev, 11, 0.3)
    bias = rnd((Cout,), torch.float32, dev, 12)
    for stride in (1, 2):
        Tout, pl, pr = O.same_pad(T, k, stride)
        Tp = T + pl + pr
        xp = torch.zeros((Bn, Tp + 2, Cin), dtype=dtype, device=dev)  # +2 rows slack
        xp[:, pl:pl + T] = x
        y = torch.empty((Bn, Tout, Cout), dtype=dtype, device=dev)
        ops.gemm(xp, w, y, Tout, Cout, k * Cin, stride * Cin, 1, Cout, 1, Cout, nbatch=Bn,
                 a_sb=(Tp + 2) * Cin, c_sb=Tout * Cout, bias=bias)
        torch.cuda.synchronize()
        ref = O.conv1d_same(x.double().cpu(), w.double().cpu(), bias.double().cpu(), stride)
        assert rel_err(y, ref) <= tol(dtype) * 4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_two_batch_levels(dev, dtype):
    """tmi_gemm_desc.nbatch2: per-(sample, head) products on [B, T, H*hd] tensors in one launch (the fp32 parity mode's
    q.k^T of W:147-153): heads are the inner batch level (stride hd inside a token row), samples the outer one."""
    ops = _ops()
    B, H, Tq, Tk, hd = 3, 4, 70, 150, 64
    D = H * hd
    q = rnd((B, Tq, D), dtype, dev, 61)
    k = rnd((B, Tk, D), dtype, dev, 62)
    P = torch.full((B, H, Tq, Tk), float("nan"), dtype=dtype, device=dev)
    ops.gemm(q, k, P, Tq, Tk, hd, D, 1, 1, D, Tk, nbatch=H, a_sb=hd, b_sb=hd, c_sb=Tq * Tk, scale_cols=Tk, scale=0.125,
             nbatch2=B, a_sb2=Tq * D, b_sb2=Tk * D, c_sb2=H * Tq * Tk)
    qh = q.double().reshape(B, Tq, H, hd).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, Tk, H, hd).permute(0, 2, 1, 3)
    ref = 0.125 * (qh @ kh.transpose(-1, -2))
    assert rel_err(P, ref) <= tol(dtype)
    # and back: ctx[b, t, h*hd + j] = sum_k P[b, h, t, k] v[b, k, h*hd + j]
    v = rnd((B, Tk, D), dtype, dev, 63)
    ctx = torch.full((B, Tq, D), float("nan"), dtype=dtype, device=dev)
    ops.gemm(P, v, ctx, Tq, hd, Tk, Tk, 1, D, 1, D, nbatch=H, a_sb=Tq * Tk, b_sb=hd, c_sb=hd,
             nbatch2=B, a_sb2=H * Tq * Tk, b_sb2=Tk * D, c_sb2=Tq * D)
    vh = v.double().reshape(B, Tk, H, hd).permute(0, 2, 1, 3)
    refc = (P.double() @ vh).permute(0, 2, 1, 3).reshape(B, Tq, D)
    assert rel_err(ctx, refc) <= tol(dtype)


def test_gemm_kbatch_splitk(dev):
    """wgrad form with the reduction running over (batch, time) and split-K atomics."""
    ops = _ops()
    Bn, T, I, Nn = 4, 70, 96, 64
    X = rnd((Bn, T, I), torch.bfloat16, dev, 13)
    dY = rnd((Bn, T, Nn), torch.bfloat16, dev, 14)
    ref = torch.einsum("bti,btn->in", X.double(), dY.double())
    for splitk in (1, 3):
        dW = torch.zeros((I, Nn), dtype=torch.float32, device=dev)
        ops.gemm(X, dY, dW, I, Nn, T, 1, I, Nn, 1, Nn, kbatch=Bn, a_skb=T * I, b_skb=T * Nn, splitk=splitk)
        torch.cuda.synchronize()
        assert rel_err(dW, ref) <= 2e-3


@pytest.mark.parametrize("M,N,K", [(800, 768, 768), (333, 200, 448), (64, 64, 64), (130, 72, 1024), (800, 768, 576),
                                   # CFG 14 (257..512 tiles, K >= 1024: two workgroups per CU on 2-stage rings): Whisper-large's
                                   # decoder shapes, an odd trip count (K = 1088: 17 tiles) with ragged edges
                                   (800, 1280, 1280), (800, 1280, 5120), (790, 1300, 1088)])
def test_gemm_k_groups_inside_the_workgroup(dev, M, N, K):
    """Grids of at most one 64x64 tile per CU with K % 64 == 0 run two K-groups of four waves (gemm_fast.hip CFG 13; CFG 14 up
    to two tiles per CU when K >= 1024): group g
    multiplies K-tiles g, g + 2, ... and the partial accumulators meet in LDS.  Odd and even trip counts (K = 576: 9 tiles,
    K = 64: one tile, the second group idle), ragged edges, the whole epilogue (bias, saved pre-activation, GELU, residual),
    and the qkv-dgrad form whose reduction runs over three K batches."""
    ops = _ops()
    bf = torch.bfloat16
    A = rnd((M, K), bf, dev, 41)
    W = rnd((K, N), bf, dev, 42, 0.1)
    bias = rnd((N,), torch.float32, dev, 43)
    R = rnd((M, N), bf, dev, 44)
    pre = A.double() @ W.double() + bias.double()
    ref = torch.nn.functional.gelu(pre) + R.double()
    for Bm, bsk, bsn in ((W, N, 1), (W.t().contiguous(), 1, K)):
        Cm = torch.full((M, N), float("nan"), dtype=bf, device=dev)
        U = torch.full((M, N), float("nan"), dtype=bf, device=dev)
        ops.gemm(A, Bm, Cm, M, N, K, K, 1, bsk, bsn, N, bias=bias, act=1, aux_out=U, resid=R, r_ld=N)
        assert rel_err(U, pre) <= 1.5e-2 and rel_err(Cm, ref) <= 1.5e-2
    if K % 192 == 0:  # dX = sum_j dY[:, j-th block] @ W_j^T, W_j = three [H, H] kernels side by side (W:89-92's q / k / v)
        H = K // 3
        Wq = rnd((3, H, N), bf, dev, 45, 0.1)           # [j][k][n]
        Cm = torch.empty((M, N), dtype=bf, device=dev)
        ops.gemm(A, Wq, Cm, M, N, H, K, 1, N, 1, N, kbatch=3, a_skb=H, b_skb=H * N)
        ref3 = sum(A[:, j * H:(j + 1) * H].double() @ Wq[j].double() for j in range(3))
        assert rel_err(Cm, ref3) <= 1.5e-2


# ----------------------------------------------------------------------------- LayerNorm etc.
# Shapes below are chosen to land on the shape-selected fast paths of gemm_fast.hip: the eight-phase
# 256x256 kernel (k-contiguous A with either B layout; M >= 2048, long K or a light epilogue), its
# k-strided-A form with split-K through workspace slabs and a K tail (weight gradients, K >= 4096),
# slab split-K on the 128x128 kernel (>= 64 K-tiles), and the 4-stage 64x64 ring (<= 256 tiles, K >= 1536).
@pytest.mark.parametrize("M,N,K,b_kn", [(2304, 1544, 1536, False), (4352, 768, 704, False), (2304, 1544, 1600, True),
                                        (4100, 776, 3008, True),  # ^ 192-row tiles (fewer CU-rounds than 256-row ones)
                                        (12000, 768, 768, False), (12000, 768, 3072, True),  # the step's N = 768 shapes (192)
                                        (8192, 2048, 1024, False), (2048, 6144, 1536, True)])  # 256-row tiles
def test_gemm_eight_phase_paths(dev, M, N, K, b_kn):
    """(>= 200 128x128 tiles: not the small-problem path.)  The library picks 192- or 256-row tiles by CU-rounds of work
    (gemm_fast.hip launch_fast); both variants and a ragged last row tile (12000 = 62 * 192 + 96) are covered."""
    ops = _ops()
    bf = torch.bfloat16
    A = rnd((M, K), bf, dev, 11)
    Bm = rnd((K, N), bf, dev, 12, 0.1)
    bias = rnd((N,), torch.float32, dev, 13)
    resid = rnd((M, N), bf, dev, 14)
    pre = A.double() @ Bm.double() + bias.double()
    Cm = torch.empty((M, N), dtype=bf, device=dev)
    if b_kn:  # forward layout: B [K, N], k-strided
        U = torch.empty_like(Cm)
        ops.gemm(A, Bm, Cm, M, N, K, K, 1, N, 1, N, bias=bias, act=1, aux_out=U, resid=resid, r_ld=N)
        ref = torch.tensor(np.vectorize(lambda v: 0.5 * v * (1 + math.erf(v / math.sqrt(2))))(pre.cpu().numpy())).to(dev) \
            + resid.double()
        assert rel_err(U, pre) <= 1.5e-2
    else:     # dgrad layout: B stored [N, K], k-contiguous
        Bt = Bm.t().contiguous()
        ops.gemm(A, Bt, Cm, M, N, K, K, 1, 1, K, N, bias=bias, resid=resid, r_ld=N)
        ref = pre + resid.double()
    assert rel_err(Cm, ref) <= 1.5e-2


@pytest.mark.parametrize("cls", ["simple", "gelu_aux", "scale", "auxin", "resid_drop", "acc", "generic"])
@pytest.mark.parametrize("M,N,K,b_kn", [(12000, 3072, 768, True),     # 756 tiles of 192 x 256: the PERSISTENT kernel, three per CU
                                        (12000, 3072, 768, False),    # the same as a dgrad (k-contiguous B)
                                        (5000, 2240, 256, False)])    # ragged both ways: 5000 = 26 * 192 + 8, 2240 = 8 * 256 + 192
def test_gemm_lean_epilogue_classes_and_persistent_tiles(dev, cls, M, N, K, b_kn):
    """Round 4: the eight-phase kernel's persistent form (more tiles than CUs: the K-tile ring runs on across tiles) and
    the class-specialised interior epilogue (gemm_fast.hip epi_class / lean_rows: EPI_SIMPLE with bias, q scale, GELU +
    saved pre-activation; EPI_AUXIN = x GELU'(aux_in); EPI_RESID = bias, dropout, + residual; EPI_ACC = C +=; everything
    else and every edge piece through the generic code) against an fp64 reference of the same bf16 operands.  Every
    row and column is checked, so a tile the persistent walk skipped or wrote twice, or an edge piece taking the interior
    path, shows."""
    ops = _ops()
    bf = torch.bfloat16
    A = rnd((M, K), bf, dev, 21)
    Bm = rnd((K, N), bf, dev, 22, 0.1)
    bias = rnd((N,), torch.float32, dev, 23)
    X = rnd((M, N), bf, dev, 24)
    Cm = torch.empty((M, N), dtype=bf, device=dev)
    if b_kn:
        Bop, b_sk, b_sn = Bm, N, 1
    else:
        Bop, b_sk, b_sn = Bm.t().contiguous(), 1, K
    acc = A.double() @ Bm.double()
    gelu = lambda v: 0.5 * v * (1 + torch.erf(v / math.sqrt(2)))
    gelu_grad = lambda u: 0.5 * (1 + torch.erf(u / math.sqrt(2))) + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    kw, aux = {}, None
    if cls == "simple":
        kw, ref = dict(bias=bias), acc + bias.double()
    elif cls == "gelu_aux":
        aux = torch.empty_like(Cm)
        kw, ref = dict(bias=bias, act=1, aux_out=aux), gelu(acc + bias.double())
    elif cls == "scale":   # the q columns of a fused qkv projection: bias, then scale on the first 256 columns
        ref = acc + bias.double()
        ref[:, :256] *= 0.125
        kw = dict(bias=bias, scale_cols=256, scale=0.125)
    elif cls == "auxin":
        kw, ref = dict(aux_in=X), acc * gelu_grad(X.double())
    elif cls == "resid_drop":
        p, seed = 0.1, 99
        from oracle import dropout as DO
        keep = torch.from_numpy(DO.keep_flat(seed, M, N, p)).to(dev)
        ref = (acc + bias.double()) * keep.double() * DO.keep_scale(p) + X.double()
        kw = dict(bias=bias, resid=X, r_ld=N, dropout_p=p, dropout_seed=seed)
    elif cls == "acc":
        Cm.copy_(X)
        kw, ref = dict(accumulate=True), acc + X.double()
    else:                  # a combination outside the classes: GELU + residual (the conv stem's epilogue)
        aux = torch.empty_like(Cm)
        kw, ref = dict(bias=bias, act=1, aux_out=aux, resid=X, r_ld=N), gelu(acc + bias.double()) + X.double()
    ops.gemm(A, Bop, Cm, M, N, K, K, 1, b_sk, b_sn, N, **kw)
    assert rel_err(Cm, ref) <= 1.5e-2, cls
    # row- and column-wise: no tile may be off (a skipped tile is an O(1) relative error of ITS rows only)
    err_r = ((Cm.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-6)).max().item()
    err_c = ((Cm.double() - ref).norm(dim=0) / ref.norm(dim=0).clamp_min(1e-6)).max().item()
    assert err_r <= 3e-2 and err_c <= 3e-2, (cls, err_r, err_c)
    if aux is not None:
        assert rel_err(aux, acc + bias.double()) <= 1.5e-2


@pytest.mark.parametrize("R,d,Vp", [(800, 768, 51904), (520, 512, 16448), (520, 256, 66048)])
def test_gemm_lm_head_shapes(dev, R, d, Vp):
    """Round 4: the LM head's three launches as whisper.py issues them (W:545, W:579-600).  Forward [R, Vp] from K = d - the
    eight-phase kernel's persistent form on 256-row tiles although R < 2048 (more than 512 tiles); weight gradient
    [d, Vp] from K = R (short reduction, > 512 tiles: eight-phase, no split); dgrad [R, d] from K = Vp into fp32 - a
    handful of tiles under a very deep reduction: 16 / 32 slab splits dealt 2 / 4 per XCD (gemm_fast.hip launch_p8,
    xsplit >= 16).  The other two shapes reach the 8-split and the 32-split forms of the same dgrad rule.  Row- and
    column-wise errors: a skipped tile or a K range added twice / never shows in ITS rows."""
    ops = _ops()
    bf = torch.bfloat16
    x = rnd((R, d), bf, dev, 31)
    W = rnd((d, Vp), bf, dev, 32, 0.05)
    dlog = rnd((R, Vp), bf, dev, 33, 0.02)
    logits = torch.empty((R, Vp), dtype=bf, device=dev)
    dW = torch.empty((d, Vp), dtype=torch.float32, device=dev)
    dx = torch.zeros((R, d), dtype=torch.float32, device=dev)

    def check(got, ref, tol):
        err_r = ((got.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-9)).max().item()
        err_c = ((got.double() - ref).norm(dim=0) / ref.norm(dim=0).clamp_min(1e-9)).max().item()
        assert err_r <= tol and err_c <= tol, (err_r, err_c)

    ops.gemm(x, W, logits, R, Vp, d, d, 1, Vp, 1, Vp)
    check(logits, x.double() @ W.double(), 1e-2)               # bf16 output rounding
    ops.gemm(x, dlog, dW, d, Vp, R, 1, d, Vp, 1, Vp, splitk=0)
    check(dW, x.double().t() @ dlog.double(), 1e-5)            # fp32 accumulation of exact bf16 products
    ops.gemm(dlog, W, dx, R, d, Vp, Vp, 1, 1, Vp, d, splitk=0)
    check(dx, dlog.double() @ W.double().t(), 1e-5)
    again = torch.zeros_like(dx)
    ops.gemm(dlog, W, again, R, d, Vp, Vp, 1, 1, Vp, d, splitk=0)
    assert torch.equal(again, dx)  # (slab splits, reduced in a fixed order: bit-reproducible)


@pytest.mark.parametrize("T,Kin,N", [(4128, 768, 2304), (4640, 1544, 1160), (4128, 256, 768), (8200, 128, 384)])
def test_gemm_weight_gradient_paths(dev, T, Kin, N):
    """dW[Kin, N] = X^T dY over T tokens (T % 64 = 32 or 8: K tail), library-chosen split-K with the
    workspace (slabs) — the first two shapes take the eight-phase kernel, the others the 128x128 one."""
    ops = _ops()
    bf = torch.bfloat16
    X = rnd((T, Kin), bf, dev, 21)
    dY = rnd((T, N), bf, dev, 22)
    ref = X.double().t() @ dY.double()
    dW = torch.full((Kin, N), 7.0, dtype=torch.float32, device=dev)   # slabs overwrite: no zeroing needed
    ops.gemm(X, dY, dW, Kin, N, T, 1, Kin, N, 1, N, splitk=0)
    assert rel_err(dW, ref) <= 1e-4
    # accumulate into an existing gradient
    base = rnd((Kin, N), torch.float32, dev, 23)
    dW2 = base.clone()
    ops.gemm(X, dY, dW2, Kin, N, T, 1, Kin, N, 1, N, splitk=0, accumulate=True)
    assert rel_err(dW2, ref + base.double()) <= 1e-4


@pytest.mark.parametrize("M,N,K", [(800, 768, 3072), (792, 512, 1536), (100, 768, 2304)])
def test_gemm_deep_ring_small_tiles(dev, M, N, K):
    ops = _ops()
    bf = torch.bfloat16
    A = rnd((M, K), bf, dev, 31)
    W = rnd((K, N), bf, dev, 32, 0.1)
    Cm = torch.empty((M, N), dtype=bf, device=dev)
    ops.gemm(A, W, Cm, M, N, K, K, 1, N, 1, N)                       # k-strided B
    ref = A.double() @ W.double()
    assert rel_err(Cm, ref) <= 1.5e-2
    Wt = W.t().contiguous()
    ops.gemm(A, Wt, Cm, M, N, K, K, 1, 1, K, N)                      # k-contiguous B
    assert rel_err(Cm, ref) <= 1.5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,Cn", [(130, 768), (7, 384), (1000, 1280), (33, 32), (513, 512), (65, 1024), (20, 2048), (4100, 768)])
def test_layernorm(dev, dtype, rows, Cn):
    ops = _ops()
    x = rnd((rows, Cn), dtype, dev, 20, 2.0) + 0.5
    gamma = rnd((Cn,), torch.float32, dev, 21) + 1.0
    beta = rnd((Cn,), torch.float32, dev, 22)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, 1e-5)
    xr = x.double().cpu().requires_grad_(True)
    gr = gamma.double().cpu().requires_grad_(True)
    br = beta.double().cpu().requires_grad_(True)
    yr = O.layer_norm(xr, gr, br, 1e-5)
    assert rel_err(y, yr) <= tol(dtype)
    dy = rnd((rows, Cn), dtype, dev, 23)
    yr.backward(dy.double().cpu())
    dx = rnd((rows, Cn), dtype, dev, 24)
    dx0 = dx.clone()
    dg = torch.zeros(Cn, dtype=torch.float32, device=dev)
    db = torch.zeros_like(dg)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, accumulate_dx=True)
    torch.cuda.synchronize()
    assert rel_err(dx, xr.grad + dx0.double().cpu()) <= tol(dtype, True)
    assert rel_err(dg, gr.grad) <= (1e-4 if dtype == torch.float32 else 2e-3)
    assert rel_err(db, br.grad) <= (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_bwd_partials_are_reproducible_and_accumulate(dev, dtype):
    """Round 3, the deterministic form (TMI_LN_DETERMINISTIC=1; not the default, see ops.LN_ATOMIC): the per-column sums
    (dgamma, dbeta, the emitted bias gradient) leave the kernel as one partial row per
    workgroup in the caller's workspace and are folded in workgroup order by a second launch - no fp32 atomics.  Two
    launches on the same inputs must therefore agree BIT FOR BIT (the atomic form differed in the last bits from run to
    run), the sums are ADDED to what the destination holds, and both forms agree to rounding."""
    ops = _ops()
    rows, Cn = 12000, 768
    x = rnd((rows, Cn), dtype, dev, 20, 2.0) + 0.5
    gamma = rnd((Cn,), torch.float32, dev, 21) + 1.0
    beta = rnd((Cn,), torch.float32, dev, 22)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, 1e-5)
    dy = rnd((rows, Cn), dtype, dev, 23)
    keep0, ops.LN_ATOMIC = ops.LN_ATOMIC, False   # the deterministic form (TMI_LN_DETERMINISTIC=1)
    assert ops.ln_bwd_workspace(dev, rows, Cn, True) is not None
    outs = []
    for rep in range(3):
        dx = torch.empty_like(x)
        dg = torch.full((Cn,), 1.5, dtype=torch.float32, device=dev)   # accumulated into
        db = torch.full((Cn,), -2.0, dtype=torch.float32, device=dev)
        cs = torch.full((Cn,), 0.25, dtype=torch.float32, device=dev)
        masked = torch.empty_like(x)
        ops.layernorm_bwd_emit(dy, x, gamma, mean, rstd, dx, dg, db, cs, masked=masked, dropout_p=0.1, dropout_seed=99)
        torch.cuda.synchronize()
        outs.append((dg.clone(), db.clone(), cs.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    # the plain entry point gives the same dgamma / dbeta (its partial rows have two sums instead of three)
    dx = torch.empty_like(x)
    dg = torch.full((Cn,), 1.5, dtype=torch.float32, device=dev)
    db = torch.full((Cn,), -2.0, dtype=torch.float32, device=dev)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db)
    torch.cuda.synchronize()
    assert torch.equal(dg, outs[0][0]) and torch.equal(db, outs[0][1])
    # against the atomic form (workspace == NULL): same sums up to the order of the additions
    ops.LN_ATOMIC = True
    try:
        dg2 = torch.full((Cn,), 1.5, dtype=torch.float32, device=dev)
        db2 = torch.full((Cn,), -2.0, dtype=torch.float32, device=dev)
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg2, db2)
        torch.cuda.synchronize()
    finally:
        ops.LN_ATOMIC = keep0
    assert rel_err(dg2, dg.double().cpu()) <= 1e-5 and rel_err(db2, db.double().cpu()) <= 1e-5
    # and the reference value: column sums of dy (dbeta) on top of the initial -2.0
    ref_db = dy.double().cpu().sum(0) - 2.0
    assert rel_err(db, ref_db) <= (1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,Cn,accumulate,drop", [(4100, 768, True, 0.1), (130, 768, False, 0.0), (1000, 1280, True, 0.25),
                                                     (65, 1024, True, 0.0), (37, 384, False, 0.1)])
def test_layernorm_bwd_emit(dev, dtype, rows, Cn, accumulate, drop):
    """tmi_layernorm_bwd_emit: the LayerNorm backward that also emits what the Dense layer below needs of dx - its bias gradient
    (column sums, accumulated) and, with dropout, the masked copy (W:205 / V:396 / V:431 in backward).  Against the oracle's
    LayerNorm (autograd) and the host generator's mask; dx / dgamma / dbeta must be those of the plain kernel."""
    ops = _ops()
    from oracle import dropout as DO
    x = rnd((rows, Cn), dtype, dev, 20, 2.0) + 0.5
    gamma = rnd((Cn,), torch.float32, dev, 21) + 1.0
    beta = rnd((Cn,), torch.float32, dev, 22)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, 1e-5)
    xr = x.double().cpu().requires_grad_(True)
    gr, br = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    dy = rnd((rows, Cn), dtype, dev, 23)
    O.layer_norm(xr, gr, br, 1e-5).backward(dy.double().cpu())
    dx0 = rnd((rows, Cn), dtype, dev, 24)
    dx = dx0.clone()
    dg = torch.zeros(Cn, dtype=torch.float32, device=dev)
    db = torch.zeros_like(dg)
    colsum = rnd((Cn,), torch.float32, dev, 25)  # accumulated into
    colsum0 = colsum.clone()
    masked = torch.full_like(x, 7.0) if drop > 0 else None
    seed = 0xABCDEF12345
    ops.layernorm_bwd_emit(dy, x, gamma, mean, rstd, dx, dg, db, colsum, masked=masked, dropout_p=drop, dropout_seed=seed,
                           accumulate_dx=accumulate)
    torch.cuda.synchronize()
    ref_dx = xr.grad + (dx0.double().cpu() if accumulate else 0.0)
    assert rel_err(dx, ref_dx) <= tol(dtype, True)
    assert rel_err(dg, gr.grad) <= (1e-4 if dtype == torch.float32 else 2e-3)
    assert rel_err(db, br.grad) <= (1e-4 if dtype == torch.float32 else 2e-3)
    emitted = dx.double().cpu()  # what the kernel stored (rounded to dtype) is what it masks / sums in fp32 before rounding
    if drop > 0:
        keep = torch.from_numpy(DO.keep_flat(seed, rows, Cn, drop))
        ref_m = ref_dx * keep * DO.keep_scale(drop)
        assert rel_err(masked, ref_m) <= tol(dtype, True)
        assert bool(((masked.double().cpu() == 0) | keep).all()) and bool((masked.double().cpu()[~keep] == 0).all())
        emitted = ref_m
    else:
        emitted = ref_dx
    got = (colsum - colsum0).double().cpu()
    ref_c = emitted.sum(0)
    scale = float(emitted.abs().sum(0).max())
    assert float((got - ref_c).abs().max()) <= (1e-5 if dtype == torch.float32 else 6e-3) * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_grad_and_gelu_bwd(dev, dtype):
    ops = _ops()
    rows, N = 1000, 768
    dy = rnd((rows, N), dtype, dev, 30)
    db = torch.zeros(N, dtype=torch.float32, device=dev)
    ops.bias_grad(dy, db)
    assert rel_err(db, dy.double().sum(0)) <= 1e-5
    u = rnd((rows, N), dtype, dev, 31, 1.5)
    dx = torch.empty_like(dy)
    ops.gelu_bwd(dy, u, dx)
    ur = u.double().cpu().requires_grad_(True)
    O.gelu_erf(ur).backward(dy.double().cpu())
    assert rel_err(dx, ur.grad) <= tol(dtype)


@pytest.mark.parametrize("Tq,Tk,mask", [(100, 100, 1), (100, 1500, 0), (64, 257, 0), (5, 5, 1)])
def test_softmax(dev, Tq, Tk, mask):
    ops = _ops()
    BH = 6
    s = rnd((BH * Tq, Tk), torch.float32, dev, 40, 3.0)
    ref_in = s.double().cpu().reshape(BH, Tq, Tk)
    if mask:
        m = torch.from_numpy(O.decoder_mask(Tq))
        add = (1.0 - m) * -1e9
        ref_in = torch.where(add != 0, (ref_in.float() + add).double(), ref_in)
    pr = torch.softmax(ref_in, -1)
    p = s.clone()
    ops.softmax_fwd(p, BH * Tq, Tq, Tk, mask)
    assert rel_err(p.reshape(BH, Tq, Tk), pr) <= 2e-6
    if mask:
        last = p.reshape(BH, Tq, Tk)[:, -1]
        assert torch.all(last == last[:, :1]), "fully masked row must be exactly uniform"
    dp = rnd((BH * Tq, Tk), torch.float32, dev, 41)
    ref = pr * (dp.double().cpu().reshape(BH, Tq, Tk) - (pr * dp.double().cpu().reshape(BH, Tq, Tk)).sum(-1, keepdim=True))
    ops.softmax_bwd(p, dp, BH * Tq, Tk)
    assert rel_err(dp.reshape(BH, Tq, Tk), ref) <= 1e-5


# ----------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, mask_mode, keep=None, keep_scale=1.0):
    """fp64 reference of W:147-167 on [B,H,T,hd] inputs (q pre-scaled)."""
    s = q @ k.transpose(-1, -2)
    if mask_mode:
        Tq = q.shape[2]
        add = (1.0 - torch.from_numpy(O.decoder_mask(Tq))) * -1e9
        s32 = s.float() + add
        s = torch.where(add != 0, s + (s32.double() - s).detach(), s)
    p = torch.softmax(s, -1)
    if keep is not None:  # tf.keras.layers.Dropout on the probabilities (W:160) with an explicit mask
        p = p * keep * keep_scale
    return p @ v


def _decode_dropmask(dm, B, H, Tq, Tk):
    """The stored keep bits of tmi_attn_fwd (include/tethys_mi.h: drop_mask; layout in csrc/attention.hip at mask_pos) as a
    [B, H, Tq, Tk] bool array."""
    import numpy as np
    NT, TQP = (Tk + 63) // 64, (Tq + 127) // 128 * 128
    w = dm.cpu().numpy().view(np.uint32).reshape(B * H, NT, 2, TQP)
    q = np.arange(Tq)
    rr = q & 31
    qpos = (q & ~63) + (q & 32) + 2 * ((rr & 3) + 4 * (rr >> 3)) + ((rr >> 2) & 1)
    k = np.arange(Tk)
    t, kk = k // 64, k % 64
    hh = (kk >> 2) & 1
    j = 8 * (kk >> 5) + 2 * ((kk >> 3) & 3) + ((kk >> 1) & 1)
    bit = j + 16 * (kk & 1)
    words = w[:, t[None, :], hh[None, :], qpos[:, None]]          # [BH, Tq, Tk]
    return ((words >> bit[None, None, :].astype(np.uint32)) & 1).astype(bool).reshape(B, H, Tq, Tk)


@pytest.mark.parametrize("B,H,Tq,Tk,mask,drop", [(2, 3, 100, 100, 1, 0.0), (1, 2, 200, 333, 0, 0.0), (2, 2, 100, 1500, 0, 0.0),
                                                 (1, 1, 31, 31, 1, 0.0), (1, 12, 1500, 1500, 0, 0.0),
                                                 (2, 3, 100, 100, 1, 0.1), (1, 2, 200, 333, 0, 0.1), (2, 2, 100, 1500, 0, 0.25),
                                                 (1, 4, 1500, 1500, 0, 0.1),
                                                 # key-split path (one query tile, >= 8 key tiles, workspace given): 3 ranges
                                                 # of 3 tiles / 4 ranges with a ragged last tile / the cross-attention shape
                                                 (2, 2, 128, 520, 0, 0.0), (1, 3, 37, 1000, 0, 0.1), (8, 12, 100, 1500, 0, 0.1),
                                                 # one query tile, <= 2 key blocks: both backward passes in one launch
                                                 (2, 3, 99, 99, 0, 0.1), (2, 2, 128, 256, 0, 0.1), (1, 2, 128, 128, 1, 0.1),
                                                 (2, 2, 249, 249, 0, 0.1), (1, 2, 256, 256, 1, 0.0)])
def test_flash_attention(dev, B, H, Tq, Tk, mask, drop):
    ops = _ops()
    D = H * 64
    bf = torch.bfloat16
    # fused QKV buffer [B,T,3D] for self-attention shapes, separate buffers otherwise
    q = rnd((B, Tq, D), bf, dev, 50, 0.35)
    k = rnd((B, Tk, D), bf, dev, 51)
    v = rnd((B, Tk, D), bf, dev, 52)
    o = torch.empty((B, Tq, D), dtype=bf, device=dev)
    stats = torch.empty((B, H, Tq, 2), dtype=torch.float32, device=dev)
    seed = 0x1234ABCD5678 + Tq
    # W:160: the mask is drawn once, in the forward, and kept for the gradient (1 bit per score in a caller-owned buffer)
    dmask = ops.attn_dropmask(dev, B, H, Tq, Tk) if drop > 0 else None
    if dmask is not None:
        dmask.fill_(0xA5)  # every word the backward reads must have been written by the forward
    ops.attn_fwd((q, 0, Tq * D, D), (k, 0, Tk * D, D), (v, 0, Tk * D, D), (o, 0, Tq * D, D), stats, B, H, Tq, Tk, mask,
                 dropout_p=drop, dropout_seed=seed, drop_mask=dmask)
    torch.cuda.synchronize()
    keep, ks = None, 1.0
    if drop > 0:  # the generator restated on the host: the kernels must drop exactly these probabilities
        from oracle import dropout as DO
        keep_np = DO.keep_attention(seed, B, H, Tq, Tk, drop)
        keep = torch.from_numpy(keep_np).double()
        ks = DO.keep_scale(drop)
        assert abs(float(keep.mean()) - (1.0 - drop)) < 0.01
        # the bits the forward stored for the backward ARE the generator's output, bit for bit
        assert (_decode_dropmask(dmask, B, H, Tq, Tk) == keep_np).all()

    def heads(t, T):
        return t.double().cpu().reshape(B, T, H, 64).permute(0, 2, 1, 3)

    qr, kr, vr = heads(q, Tq).requires_grad_(True), heads(k, Tk).requires_grad_(True), heads(v, Tk).requires_grad_(True)
    outr = _attn_ref(qr, kr, vr, mask, keep, ks)
    ref_o = outr.permute(0, 2, 1, 3).reshape(B, Tq, D)
    from _margins import within
    tag = f"attn[{'drop' if drop > 0 else 'nodrop'},mask{mask}]"
    within(tag + " fwd o", rel_err(o, ref_o), 1e-2)        # measured <= 4.4e-3 (profiles/r02_test_margins.json)
    do = rnd((B, Tq, D), bf, dev, 53)
    outr.backward(heads(do, Tq))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty((B, H, Tq), dtype=torch.float32, device=dev)
    ops.attn_bwd((q, 0, Tq * D, D), (k, 0, Tk * D, D), (v, 0, Tk * D, D), (o, 0, Tq * D, D), stats,
                 (do, 0, Tq * D, D), (dq, 0, Tq * D, D), (dk, 0, Tk * D, D), (dv, 0, Tk * D, D), delta,
                 B, H, Tq, Tk, mask, dq_scale=0.5, dropout_p=drop, dropout_seed=seed + 1, drop_mask=dmask)  # (the backward never hashes: its seed is unused)
    torch.cuda.synchronize()

    def merge(t, T):
        return t.permute(0, 2, 1, 3).reshape(B, T, D)

    within(tag + " bwd dv", rel_err(dv, merge(vr.grad, Tk)), 1.5e-2)   # measured <= 4.7e-3
    within(tag + " bwd dk", rel_err(dk, merge(kr.grad, Tk)), 1.5e-2)   # measured <= 6.2e-3
    within(tag + " bwd dq", rel_err(dq, 0.5 * merge(qr.grad, Tq)), 1.5e-2)   # measured <= 5.9e-3


@pytest.mark.parametrize("B,H,Tq,Tk,mask,drop", [(2, 3, 100, 100, 1, 0.1), (8, 12, 99, 99, 0, 0.1), (1, 1, 31, 31, 1, 0.0),
                                                 (2, 2, 128, 256, 0, 0.1), (2, 2, 100, 200, 0, 0.0), (1, 2, 128, 128, 1, 0.1),
                                                 (4, 12, 249, 249, 0, 0.1), (1, 2, 256, 256, 1, 0.0), (1, 2, 200, 130, 0, 0.1)])
def test_attn_bwd_one_launch_equals_the_two_passes(dev, B, H, Tq, Tk, mask, drop):
    """Small problems (Tq <= 256, Tk <= 256: the decoder's self-attention, Wav2Vec2's) run both backward passes as ONE launch
    (attn_bwd_small_kernel: the dK/dV blocks compute the row sums delta themselves, in the dQ pass's summation order).  The
    result is bit for bit what the two separate launches (bwd_passes 1, then 2) write."""
    ops = _ops()
    D = H * 64
    bf = torch.bfloat16
    q, k, v = rnd((B, Tq, D), bf, dev, 60, 0.35), rnd((B, Tk, D), bf, dev, 61), rnd((B, Tk, D), bf, dev, 62)
    do = rnd((B, Tq, D), bf, dev, 63)
    o = torch.empty((B, Tq, D), dtype=bf, device=dev)
    stats = torch.empty((B, H, Tq, 2), dtype=torch.float32, device=dev)
    dmask = ops.attn_dropmask(dev, B, H, Tq, Tk) if drop > 0 else None
    Q, K, V, O = (q, 0, Tq * D, D), (k, 0, Tk * D, D), (v, 0, Tk * D, D), (o, 0, Tq * D, D)
    ops.attn_fwd(Q, K, V, O, stats, B, H, Tq, Tk, mask, dropout_p=drop, dropout_seed=77, drop_mask=dmask)
    res = []
    for passes in ((0,), (1, 2)):
        dq, dk, dv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
        delta = torch.zeros((B, H, Tq), dtype=torch.float32, device=dev)
        for ps in passes:
            ops.attn_bwd(Q, K, V, O, stats, (do, 0, Tq * D, D), (dq, 0, Tq * D, D), (dk, 0, Tk * D, D), (dv, 0, Tk * D, D), delta,
                         B, H, Tq, Tk, mask, dropout_p=drop, dropout_seed=77, drop_mask=dmask, passes=ps)
        torch.cuda.synchronize()
        res.append((dq, dk, dv, delta))
    for name, a, b in zip(("dq", "dk", "dv", "delta"), res[0], res[1]):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_resid", [False, True])
@pytest.mark.parametrize("cols", [768, 770])  # 16-byte path / pair-at-a-time path
def test_dropout_kernel_matches_host_generator(dev, dtype, with_resid, cols):
    """tmi_dropout against the generator restated in oracle/dropout.py: same mask bit for bit, kept values
    scaled by 65536/(65536-thr), residual added in fp32 (W:205 / W:342 / W:411 sites and their backward)."""
    from oracle import dropout as DO
    ops = _ops()
    rows, p, seed = 301, 0.1, 0xFEEDFACE12345
    x = rnd((rows, cols + 8), dtype, dev, 90)[:, :cols]      # row stride != cols
    resid = rnd((rows, cols), dtype, dev, 91) if with_resid else None
    out = torch.empty((rows, cols), dtype=dtype, device=dev)
    ops.dropout(x, out, rows, cols, p, seed, resid=resid)
    keep = torch.from_numpy(DO.keep_flat(seed, rows, cols, p)).to(dev)
    ref = torch.where(keep, x.float() * DO.keep_scale(p), torch.zeros((), device=dev))
    if with_resid:
        ref = ref + resid.float()
    assert torch.equal(out, ref.to(dtype))
    assert abs(float(keep.float().mean()) - 0.9) < 5e-3
    # in place, and the backward use: the same call on a gradient reproduces the mask
    g = rnd((rows, cols), dtype, dev, 92)
    g2 = g.clone()
    ops.dropout(g2, g2, rows, cols, p, seed)
    assert torch.equal(g2, torch.where(keep, g.float() * DO.keep_scale(p), torch.zeros((), device=dev)).to(dtype))


@pytest.mark.parametrize("M,N,K,variant", [(600, 768, 3072, "kc_ks"), (3000, 768, 768, "kc_ks"), (800, 3072, 768, "gelu"),
                                             (300, 96, 64, "small"), (2100, 768, 1024, "kc_kc")])
def test_gemm_epilogue_dropout_matches_host_generator(dev, M, N, K, variant):
    """Dropout as a GEMM-epilogue term (tmi_gemm_desc.dropout_p; W:205 x + Dropout(fc2(g)), V:393 / V:396 / V:431): the
    mask of tmi_dropout over the [M, N] output (oracle/dropout.py keep_flat), applied to the fp32 epilogue value after
    bias / GELU / gelu' and before the residual add.  The shapes walk the kernels the step uses (eight-phase 256x256,
    128x128, 64x64 tiles; k-strided and k-contiguous B)."""
    from oracle import dropout as DO
    ops = _ops()
    bf = torch.bfloat16
    p, seed = 0.1, 0x1234ABCD5678
    A = rnd((M, K), bf, dev, 1, 0.5)
    W = rnd((K, N), bf, dev, 2, 0.05)
    bias = rnd((N,), torch.float32, dev, 3, 0.1)
    resid = rnd((M, N), bf, dev, 4)
    keep = torch.from_numpy(DO.keep_flat(seed, M, N, p)).to(dev)
    base = A.float() @ W.float() + bias
    out = torch.empty((M, N), dtype=bf, device=dev)
    if variant == "gelu":      # act before the mask, saved pre-activation untouched
        u = torch.empty_like(out)
        ops.gemm(A, W, out, M, N, K, K, 1, N, 1, N, bias=bias, act=1, aux_out=u, dropout_p=p, dropout_seed=seed)
        ref = torch.where(keep, torch.nn.functional.gelu(base) * DO.keep_scale(p), torch.zeros((), device=dev))
        assert rel_err(u, base) <= 1e-2
    elif variant == "kc_kc":   # dgrad form: B given k-contiguous ([N, K] rows), gelu' factor, then the mask
        Wt = W.t().contiguous()
        uu = rnd((M, N), bf, dev, 5)
        ops.gemm(A, Wt, out, M, N, K, K, 1, 1, K, N, aux_in=uu, dropout_p=p, dropout_seed=seed)
        x = uu.float()
        gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * 3.141592653589793) ** 0.5
        ref = torch.where(keep, (A.float() @ W.float()) * gp * DO.keep_scale(p), torch.zeros((), device=dev))
        resid = None
    else:
        ops.gemm(A, W, out, M, N, K, K, 1, N, 1, N, bias=bias, resid=resid, r_ld=N, dropout_p=p, dropout_seed=seed)
        ref = torch.where(keep, base * DO.keep_scale(p), torch.zeros((), device=dev)) + resid.float()
    torch.cuda.synchronize()
    if variant == "gelu":
        resid = None
    # dropped positions hold exactly the residual (or zero); kept ones the scaled value to bf16 accuracy
    dropped = ~keep
    want0 = resid.float() if resid is not None else torch.zeros((M, N), device=dev)
    assert torch.equal(out.float()[dropped], want0.to(bf).float()[dropped])
    assert rel_err(out, ref) <= 1.5e-2
    assert abs(float(keep.float().mean()) - 0.9) < 5e-3


def test_adam_rows_skips_idle_rows_and_matches_dense_bit_for_bit(dev):
    """tmi_adam_step_rows over an embedding table (W:382): rows that never see a gradient are not touched, every other
    row gets exactly the dense kernel's update - including rows whose gradient is zero in a later step (m, v decay)."""
    ops = _ops()
    rows, d = 300, 768
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(rows, d, generator=g).to(dev)
    pa, pb = p0.clone(), p0.clone()
    ma, va = torch.zeros_like(pa), torch.zeros_like(pa)
    mb, vb = torch.zeros_like(pa), torch.zeros_like(pa)
    mira, mirb = torch.zeros(rows, d, dtype=torch.bfloat16, device=dev), torch.zeros(rows, d, dtype=torch.bfloat16, device=dev)
    mirb.copy_(p0)
    mira.copy_(p0)
    active = torch.zeros(rows, dtype=torch.uint8, device=dev)
    touched = set()
    for step in range(1, 5):
        idx = torch.randint(0, rows // 3, (40,), generator=g)  # only the first third of the table ever sees gradients
        if step == 3:
            idx = idx[:5]                                        # most earlier rows now have g = 0 but m, v != 0
        touched |= set(idx.tolist())
        gr = torch.zeros(rows, d)
        gr[idx] = torch.randn(len(idx), d, generator=g)
        ga, gb = gr.to(dev), gr.to(dev)
        ops.adam_step(pa.view(-1), ga.view(-1), ma.view(-1), va.view(-1), rows * d, 1e-2, 0.9, 0.999, 1e-7, step, mirror=mira.view(-1),
                      zero_grad=True)
        ops.adam_step_rows(pb.view(-1), gb.view(-1), mb.view(-1), vb.view(-1), rows, d, active, 1e-2, 0.9, 0.999, 1e-7, step,
                           mirror=mirb.view(-1), zero_grad=True)
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(mira, mirb)
        assert float(gb.abs().max()) == 0.0 and float(ga.abs().max()) == 0.0
    assert set(torch.nonzero(active).flatten().tolist()) == touched
    assert torch.equal(pb[rows // 3:], p0[rows // 3:])


# ----------------------------------------------------------------------------- embed / xent / adam / misc
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding(dev, dtype):
    ops = _ops()
    B, S, D, V, start = 4, 12, 64, 128, 127
    _, labels = O.create_dummy_pool(seed=3, n_mels=4, seq_len=8, max_target_length=S, num_samples=B)
    lab = torch.from_numpy(labels).to(dev)
    table = rnd((V, D), torch.float32, dev, 60)
    pe = torch.from_numpy(O.positional_encoding(S, D)).to(dev)
    out = torch.empty((B, S, D), dtype=dtype, device=dev)
    ops.embed_fwd(lab, table, pe, out, B, S, D, start)
    ids = O.decoder_input_ids(torch.from_numpy(labels), start).long()
    ref = table.double().cpu()[ids] + pe.double().cpu()[:S]
    assert rel_err(out, ref) <= (1e-7 if dtype == torch.float32 else 8e-3)
    dy = rnd((B, S, D), dtype, dev, 61)
    dt = torch.zeros((V, D), dtype=torch.float32, device=dev)
    ops.embed_bwd(lab, dy, dt, B, S, D, start)
    refg = torch.zeros((V, D), dtype=torch.float64)
    refg.index_add_(0, ids.reshape(-1), dy.double().cpu().reshape(-1, D))
    assert rel_err(dt, refg) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
# (bf16: rows held on chip - the step's own (V, ld), whole padding chunks behind V, the longest row that kernel takes, a ragged
# chunk in the middle of a chunk slot - / too long for it)
@pytest.mark.parametrize("V,ld", [(51865, 51872), (51865, 51904), (53241, 53248), (1003, 1024), (128, 128), (60001, 60008)])
def test_xent(dev, dtype, V, ld):
    ops = _ops()
    B, S = 3, 10
    _, labels = O.create_dummy_pool(seed=4, n_mels=4, seq_len=8, max_target_length=S, num_samples=B)
    lab = torch.from_numpy(labels).to(dev)
    logits = torch.zeros((B * S, ld), dtype=dtype, device=dev)
    logits[:, :V] = rnd((B * S, V), dtype, dev, 70, 2.0)
    lr = logits[:, :V].double().cpu().reshape(B, S, V).requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lr[:, :-1].reshape(-1, V), torch.from_numpy(labels)[:, 1:].long().reshape(-1))
    loss.backward()
    row_loss = torch.empty(B * S, dtype=torch.float32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    gs = 1.0 / (B * (S - 1))
    ops.xent_fwd_bwd(logits, ld, lab, row_loss, B, S, V, gs)
    ops.sum_scale(row_loss, out, B * S, gs)
    torch.cuda.synchronize()
    assert abs(float(out) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach())) + 1e-6
    assert rel_err(logits[:, :V].reshape(B, S, V), lr.grad) <= (1e-5 if dtype == torch.float32 else 8e-3)
    assert float(logits[:, V:].abs().max()) == 0.0 if ld > V else True


@pytest.mark.parametrize("R_S,d,V,ld", [((3, 10), 96, 1003, 1024), ((2, 7), 768, 51865, 51904), ((2, 5), 64, 60001, 60008)])
def test_linear_xent_takes_the_target_logit_from_the_operands(dev, R_S, d, V, ld):
    """tmi_linear_xent (round 4): cross-entropy of bf16 logits = x . w whose LOSS uses z_target recomputed in fp32 from x and w
    (the bf16 logits have lost its low bits); lse and the gradient come from the stored logits exactly as in tmi_xent_fwd_bwd.
    Checked against that definition in fp64, for the on-chip-row kernel and the generic one (60001 columns), with logits large
    enough (|z| up to ~8) for the rounding to matter; the gradient must be bit-identical to the plain entry's; and the loss
    must be closer to the cross-entropy of the unrounded logits than the plain one is."""
    ops = _ops()
    B, S = R_S
    bf = torch.bfloat16
    x = rnd((B * S, d), bf, dev, 91, 1.0)
    w = torch.zeros((d, ld), dtype=bf, device=dev)
    w[:, :V] = rnd((d, V), bf, dev, 92, 2.0 / math.sqrt(d))
    _, labels = O.create_dummy_pool(seed=4, n_mels=4, seq_len=8, max_target_length=S, num_samples=B)
    lab = torch.from_numpy(labels).to(dev)
    logits = torch.empty((B * S, ld), dtype=bf, device=dev)
    ops.gemm(x, w, logits, B * S, ld, d, d, 1, ld, 1, ld)
    stored = logits.clone()
    plain = stored.clone()
    gs = 1.0 / (B * (S - 1))
    rl_plain = torch.empty(B * S, dtype=torch.float32, device=dev)
    rl = torch.empty_like(rl_plain)
    ops.xent_fwd_bwd(plain, ld, lab, rl_plain, B, S, V, gs)
    ops.xent_fwd_bwd(logits, ld, lab, rl, B, S, V, gs, lm=(x, d, w, ld, 1, d))
    torch.cuda.synchronize()
    assert torch.equal(logits, plain)                      # same gradient, bit for bit
    z = stored.double()[:, :V].view(B, S, V)               # what the kernel sees
    z_exact = (x.double() @ w.double())[:, :V].view(B, S, V)
    tgt = lab[:, 1:].long()
    # the log-sum-exp over the stored logits with the TARGET's term taken at the recomputed logit too (both terms
    # of the loss see the same target logit, so its rounding cancels where the target dominates)
    zt_exact = z_exact[:, :-1].gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
    zmix = z[:, :-1].clone()
    zmix.scatter_(-1, tgt.unsqueeze(-1), zt_exact.unsqueeze(-1))
    want = torch.logsumexp(zmix, dim=-1) - zt_exact
    got = rl.view(B, S)
    assert float(got[:, -1].abs().max()) == 0.0
    assert float((got[:, :-1].double() - want).abs().max()) <= 5e-5
    full = torch.logsumexp(z_exact[:, :-1], dim=-1) - z_exact[:, :-1].gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
    err_new = float((got[:, :-1].double() - full).abs().mean())
    err_old = float((rl_plain.view(B, S)[:, :-1].double() - full).abs().mean())
    assert err_new < 0.5 * err_old, (err_new, err_old)


@pytest.mark.parametrize("V,ld", [(1000, 1024), (60001, 60032)])  # the on-chip-row kernel / the generic one
def test_linear_xent_with_a_dominant_target_logit(dev, V, ld):
    """a confident model - target logit ~ 20, every other ~ 0 - is where lse(rounded logits) - z_target(fp32)
    keeps the rounding error of the stored target logit (half a bf16 ulp: 0.06 at |z| in [16, 32)) and can go negative.
    With the target's term of the log-sum-exp at the recomputed logit as well, every row loss is >= 0 and at least as
    close to the fp64 loss of the unrounded logits as the plain entry's."""
    ops = _ops()
    B, S, d = 2, 9, 64
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    lab = torch.randint(3, V, (B, S), generator=g, dtype=torch.int32).to(dev)
    # row r reads feature r only: logits[r, :] = 3 * w[r, :]; w[r, target_r] ~ 6.3 .. 7.0 puts the row's target logit at
    # 18.9 .. 21.1 (mostly off the bf16 grid, ulp 0.125 there), every other logit of the row is ~ 0.03
    x = torch.zeros((B * S, d), dtype=bf, device=dev)
    x[torch.arange(B * S), torch.arange(B * S)] = 3.0
    w = (torch.randn((d, ld), generator=g) * 0.01).to(bf).to(dev)
    w[:, V:] = 0
    tg = torch.cat([lab[:, 1:], lab[:, :1]], dim=1).reshape(-1).long()   # target of row (b, t) = labels[b, t + 1]
    w[torch.arange(B * S), tg] = torch.linspace(6.31, 7.03, B * S).to(bf).to(dev)
    logits = torch.empty((B * S, ld), dtype=bf, device=dev)
    ops.gemm(x, w, logits, B * S, ld, d, d, 1, ld, 1, ld)
    plain = logits.clone()
    rl_plain = torch.empty(B * S, dtype=torch.float32, device=dev)
    rl = torch.empty_like(rl_plain)
    gs = 1.0 / (B * (S - 1))
    ops.xent_fwd_bwd(plain, ld, lab, rl_plain, B, S, V, gs)
    ops.xent_fwd_bwd(logits, ld, lab, rl, B, S, V, gs, lm=(x, d, w, ld, 1, d))
    torch.cuda.synchronize()
    assert torch.equal(logits, plain)
    ze = (x.double() @ w.double())[:, :V].view(B, S, V)[:, :-1]
    tgt = lab[:, 1:].long()
    full = torch.logsumexp(ze, dim=-1) - ze.gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
    got = rl.view(B, S)[:, :-1].double()
    old = rl_plain.view(B, S)[:, :-1].double()
    assert float(got.min()) >= 0.0, float(got.min())
    assert float((got - full).abs().max()) <= float((old - full).abs().max()) + 1e-6, ((got - full).abs().max(), (old - full).abs().max())


@pytest.mark.parametrize("eps_mode", [0, 1])
def test_adam(dev, eps_mode):
    ops = _ops()
    n = 10007
    p = rnd((n,), torch.float32, dev, 80)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    params = {"w": p.double().cpu().clone()}
    st = O.AdamState()
    for step in range(1, 4):
        g = rnd((n,), torch.float32, dev, 80 + step, 0.01)
        ops.adam_step(p, g, m, v, n, 1e-4, 0.9, 0.999, 1e-7, step, eps_mode=eps_mode)
        O.adam_step(params, {"w": g.double().cpu()}, st, lr=1e-4, eps=1e-7, eps_mode="tf" if eps_mode == 0 else "torch")
    torch.cuda.synchronize()
    # 3 steps on |p| up to ~4: a couple of fp32 ulps (2.4e-7 at |p| in [2,4))
    assert float((p.double().cpu() - params["w"]).abs().max()) <= 1e-6


def test_casts_and_feats(dev):
    ops = _ops()
    R, Cc = 300, 205
    src = rnd((R, Cc), torch.float32, dev, 90)
    ldd = 208
    d1 = torch.full((R, ldd), 7.0, dtype=torch.bfloat16, device=dev)
    ops.cast_bf16(src, Cc, d1, ldd, R, Cc)
    assert torch.equal(d1[:, :Cc], src.to(torch.bfloat16)) and float(d1[:, Cc:].abs().max()) == 0.0
    d2 = torch.empty((Cc, 304), dtype=torch.bfloat16, device=dev)
    ops.transpose_cast_bf16(src, Cc, d2, 304, R, Cc)
    assert torch.equal(d2[:, :R], src.t().to(torch.bfloat16))
    Bn, Cn, T = 2, 80, 300
    f = rnd((Bn, Cn, T), torch.float32, dev, 91)
    for dtype in (torch.float32, torch.bfloat16):
        out = torch.full((Bn, T + 2, Cn), 9.0, dtype=dtype, device=dev)
        ops.feat_to_channels_last(f, out, Bn, Cn, T, 1, 1)
        assert torch.equal(out[:, 1:T + 1], f.transpose(1, 2).to(dtype))
        assert float(out[:, 0].abs().max()) == 0.0 and float(out[:, T + 1].abs().max()) == 0.0
    x = rnd((100003,), torch.float32, dev, 92)
    o = torch.empty(1, dtype=torch.float32, device=dev)
    ops.sumsq(x, o, x.numel())
    assert abs(float(o) - float((x.double() ** 2).sum())) <= 1e-4 * float((x.double() ** 2).sum())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_dropout_fused_equals_layernorm_then_dropout(dev, dtype):
    """tmi_layernorm_dropout_fwd / _bwd (V:296, V:560, V:779: LayerNorm then Dropout) against the two-launch form they
    replace: the forward's output is tmi_dropout(tmi_layernorm_fwd(x)) (same generator, same counters), the backward
    equals tmi_layernorm_bwd fed the tmi_dropout-masked dy - bit for bit in fp32 (dgamma / dbeta up to the order of their
    atomics), to a rounding in bf16 (the fused passes round once where the two launches round twice) with the same zero
    pattern - and the mask is the host generator's."""
    from oracle import dropout as DO
    ops = _ops()
    rows, C, p, seed = 803, 768, 0.1, 0xABCDEF12345
    x = rnd((rows, C), dtype, dev, 11, 1.3)
    gamma = rnd((C,), torch.float32, dev, 12, 0.2) + 1.0
    beta = rnd((C,), torch.float32, dev, 13, 0.2)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    y_ref = torch.empty_like(x)
    ops.layernorm_fwd(x, gamma, beta, y_ref, mean, rstd, 1e-5)
    ln_only = y_ref.clone()
    ops.dropout(y_ref, y_ref, rows, C, p, seed)
    y = torch.empty_like(x)
    ops.layernorm_dropout_fwd(x, gamma, beta, y, mean, rstd, 1e-5, p, seed)
    exact = dtype == torch.float32  # (bf16: the two-launch form rounds LayerNorm's output, scales, and rounds again; the fused one rounds once)
    assert torch.equal(y, y_ref) if exact else rel_err(y, y_ref) <= 1e-2
    assert torch.equal(y != 0, y_ref != 0)
    keep = torch.from_numpy(DO.keep_flat(seed, rows, C, p)).to(dev)
    assert torch.equal(y != 0, keep & (ln_only != 0))
    dy = rnd((rows, C), dtype, dev, 14)
    dym = dy.clone()
    ops.dropout(dym, dym, rows, C, p, seed)
    dx_ref, dg_ref, db_ref = torch.empty_like(x), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_bwd(dym, x, gamma, mean, rstd, dx_ref, dg_ref, db_ref)
    dx, dg, db = torch.empty_like(x), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_dropout_bwd(dy, x, gamma, mean, rstd, dx, dg, db, p, seed)
    assert torch.equal(dx, dx_ref) if exact else rel_err(dx, dx_ref) <= 1.5e-2
    tol_ = 1e-5 if exact else 1e-2
    assert rel_err(dg, dg_ref) <= tol_ and rel_err(db, db_ref) <= tol_


# ----------------------------------------------------------------------------- kernel-level reproducibility
def _busy(dev, stream, seconds_of_kernels=0.004):
    """Co-running filler on a second stream: uneven load is what exposed the round-4 attention hazard (stale MFMA
    accumulators were read on SOME waves of SOME launches, more of them the busier the CU)."""
    a = torch.randn(2048, 2048, device=dev)
    with torch.cuda.stream(stream):
        for _ in range(12):
            a = torch.tanh(a @ a * 1e-3)
    return a


def test_kernels_are_bit_reproducible_under_load(dev):
    """every hand-scheduled kernel launched TWICE on the same inputs, with other work running beside
    it on a second stream, must give the same bits - tmi_attn_fwd and both passes of tmi_attn_bwd (with and without
    dropout: the stored mask too), the eight-phase bf16 GEMM (persistent walk and plain, split-K dealt over XCDs through
    workspace slabs) and the fp32-MFMA GEMM.  A VALU read of an MFMA result inside its hazard window (the round-4
    attention bug) shows up here as a handful of differing elements; tools/check_mfma_hazards.py is the static half."""
    ops = _ops()
    bf = torch.bfloat16
    side = torch.cuda.Stream(device=dev)

    def twice(fn, outs):
        got = []
        for _ in range(2):
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            keep = _busy(dev, side)
            fn()
            torch.cuda.synchronize()
            got.append([o.clone() for o in outs])
            del keep
        return all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(got[0], got[1]))

    # attention: the encoder shape cut to 2 x 12 x 1500 (every tile kind: full tiles, the ragged last key tile) and the
    # cross-attention shape with its key split + combine
    for (B, H, Tq, Tk) in ((2, 12, 1500, 1500), (8, 12, 100, 1500)):
        D = H * 64
        q, k, v, do = (rnd((B, T, D), bf, dev, 60 + i, 0.5) for i, T in enumerate((Tq, Tk, Tk, Tq)))
        o = torch.empty((B, Tq, D), dtype=bf, device=dev)
        stats = torch.empty((B, H, Tq, 2), dtype=torch.float32, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty((B, H, Tq), dtype=torch.float32, device=dev)
        Q, K, V, Om = (q, 0, Tq * D, D), (k, 0, Tk * D, D), (v, 0, Tk * D, D), (o, 0, Tq * D, D)
        for p in (0.0, 0.1):
            dm = ops.attn_dropmask(dev, B, H, Tq, Tk) if p > 0 else None
            outs = [o, stats] + ([dm] if dm is not None else [])
            assert twice(lambda: ops.attn_fwd(Q, K, V, Om, stats, B, H, Tq, Tk, 0, dropout_p=p, dropout_seed=5, drop_mask=dm), outs), \
                f"tmi_attn_fwd not reproducible (B {B} Tq {Tq} Tk {Tk} p {p})"
            ops.attn_fwd(Q, K, V, Om, stats, B, H, Tq, Tk, 0, dropout_p=p, dropout_seed=5, drop_mask=dm)
            assert twice(lambda: ops.attn_bwd(Q, K, V, Om, stats, (do, 0, Tq * D, D), (dq, 0, Tq * D, D), (dk, 0, Tk * D, D),
                                              (dv, 0, Tk * D, D), delta, B, H, Tq, Tk, 0, dropout_p=p, dropout_seed=5, drop_mask=dm),
                         [dq, dk, dv, delta]), f"tmi_attn_bwd not reproducible (B {B} Tq {Tq} Tk {Tk} p {p})"
    # GEMMs: forward FFN shape (eight-phase, persistent walk: 12000 x 3072 x 768), its N = 768 sibling (189 tiles, no
    # persistence), the weight gradient (k-strided operands, fp32 out, split-K through slabs dealt over XCDs) and an fp32 one
    M, d_, ff = 12000, 768, 3072
    x = rnd((M, d_), bf, dev, 70, 0.5)
    w1 = rnd((d_, ff), bf, dev, 71, 0.05)
    w2 = rnd((ff, d_), bf, dev, 72, 0.05)
    h = torch.empty((M, ff), dtype=bf, device=dev)
    y = torch.empty((M, d_), dtype=bf, device=dev)
    assert twice(lambda: ops.gemm(x, w1, h, M, ff, d_, d_, 1, ff, 1, ff), [h]), "bf16 GEMM 12000x3072x768 not reproducible"
    ops.gemm(x, w1, h, M, ff, d_, d_, 1, ff, 1, ff)
    assert twice(lambda: ops.gemm(h, w2, y, M, d_, ff, ff, 1, d_, 1, d_), [y]), "bf16 GEMM 12000x768x3072 not reproducible"
    gw = torch.empty((d_, ff), dtype=torch.float32, device=dev)
    assert twice(lambda: ops.gemm(x, h, gw, d_, ff, M, 1, d_, ff, 1, ff, splitk=0), [gw]), "weight-gradient GEMM (slab split-K) not reproducible"
    a32 = rnd((1500, 768), torch.float32, dev, 73, 0.5)
    b32 = rnd((768, 2304), torch.float32, dev, 74, 0.05)
    c32 = torch.empty((1500, 2304), dtype=torch.float32, device=dev)
    assert twice(lambda: ops.gemm(a32, b32, c32, 1500, 2304, 768, 768, 1, 2304, 1, 2304), [c32]), "fp32-MFMA GEMM not reproducible"
