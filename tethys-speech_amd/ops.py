"""Host-side operator wrappers: torch tensors in, C-ABI calls out.

Each function resolves pointers / element strides from torch tensors (device memory is
PyTorch's caching allocator: plumbing) and launches the HIP kernel on torch's current
stream.  Nothing here computes on the host or falls back to torch ops.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import TMI_BF16, TMI_F32, AttnDesc, GemmDesc, check, lib


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TMI_F32
    if t.dtype == torch.bfloat16:
        return TMI_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


_STREAM_OVERRIDE: Optional[int] = None


def stream() -> int:
    """hipStream_t every wrapper launches on: torch's current stream, unless a caller has pinned
    another one with ``set_stream`` (cheaper than switching torch's current stream per launch)."""
    return _STREAM_OVERRIDE if _STREAM_OVERRIDE is not None else torch.cuda.current_stream().cuda_stream


def set_stream(handle: Optional[int]) -> Optional[int]:
    """Pin (or, with None, unpin) the launch stream; returns the previous pin so callers can nest."""
    global _STREAM_OVERRIDE
    prev, _STREAM_OVERRIDE = _STREAM_OVERRIDE, handle
    return prev


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class OpProfile:
    """bench.py's roofline probe: HIP events (recorded on the launch stream) around every wrapped launch,
    folded per kernel class together with the launch's ALGORITHMIC work (flop for the MFMA classes, bytes
    for the HBM classes).  Events are read out every ``flush_every`` records: with ~500 timing events
    outstanding the runtime stalls the stream for tens of milliseconds on a record."""

    def __init__(self, flush_every=160):
        self.records = []
        self.flush_every = flush_every
        self.acc = {}  # class -> [ms, work, launches]

    def add(self, cls, e0, e1, work):
        self.records.append((cls, e0, e1, work))
        if len(self.records) >= self.flush_every:
            self.flush()

    def flush(self):
        if not self.records:
            return
        torch.cuda.synchronize()
        for cls, a, b, w in self.records:
            t = self.acc.setdefault(cls, [0.0, 0.0, 0])
            t[0] += a.elapsed_time(b)
            t[1] += w
            t[2] += 1
        self.records = []

    def totals(self, cls="gemm"):
        self.flush()
        ms, work, n = self.acc.get(cls, [0.0, 0.0, 0])
        return ms, work, n


GemmProfile = OpProfile  # (earlier name)
PROFILE = None  # set to an OpProfile() to instrument


class _probe:
    """``with _probe(cls, work):`` times the launches inside it when a profile is installed."""
    __slots__ = ("cls", "work", "e0")

    def __init__(self, cls, work):
        self.cls, self.work, self.e0 = cls, work, None

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.add(self.cls, self.e0, e1, self.work)
        return False


_WORKSPACE = {}
GEMM_WORKSPACE_BYTES = 96 << 20  # TMI_GEMM_WORKSPACE_MIN


def gemm_workspace(device):
    """Split-K scratch of tmi_gemm (one fp32 slab per split): one per (device, stream), since calls
    that share it must be ordered on one stream."""
    key = (device.type, device.index, stream())  # one per stream in use
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = _WORKSPACE[key] = torch.empty(GEMM_WORKSPACE_BYTES, dtype=torch.uint8, device=device)  # contents irrelevant
    return ws


_GEMM_DESCS = {}   # call-site key -> (descriptor, byref): a step repeats the same ~150 launches with the same pointers and shapes


def gemm(A, B, Cm, M, N, K, a_sm, a_sk, b_sk, b_sn, ldc, *, nbatch=1, a_sb=0, b_sb=0, c_sb=0,
         kbatch=1, a_skb=0, b_skb=0, bias=None, bias_sb=0, scale_cols=0, scale=1.0, accumulate=False,
         act=0, aux_out=None, aux_in=None, resid=None, r_ld=0, r_sb=0, splitk=1,
         a_off=0, b_off=0, c_off=0, dropout_p=0.0, dropout_seed=0, nbatch2=1, a_sb2=0, b_sb2=0, c_sb2=0):
    """C = epi(A·B); see tmi_gemm in include/tethys_mi.h.  ``*_off`` are element offsets
    added to the tensors' base pointers (aux_* share c_off).  ``nbatch2`` / ``*_sb2``: an outer batch level.
    The descriptor is a pure function of the arguments (only the dropout seed changes from step to step), so it is built
    once per distinct call and reused: filling 45 ctypes fields cost more host time than the launch itself (round 4)."""
    ws = gemm_workspace(Cm.device) if (splitk == 0 and Cm.is_cuda) else None
    key = (A.data_ptr(), B.data_ptr(), Cm.data_ptr(), M, N, K, a_sm, a_sk, b_sk, b_sn, ldc, nbatch, a_sb, b_sb, c_sb, kbatch, a_skb,
           b_skb, None if bias is None else bias.data_ptr(), bias_sb, scale_cols, scale, accumulate, act,
           None if aux_out is None else aux_out.data_ptr(), None if aux_in is None else aux_in.data_ptr(),
           None if resid is None else resid.data_ptr(), r_ld, r_sb, splitk, a_off, b_off, c_off, dropout_p, nbatch2, a_sb2, b_sb2,
           c_sb2, A.dtype, B.dtype, Cm.dtype, None if ws is None else ws.data_ptr())
    hit = _GEMM_DESCS.get(key)
    if hit is None:
        d = GemmDesc()
        esA, esC = A.element_size(), Cm.element_size()
        d.A = A.data_ptr() + a_off * esA
        d.B = B.data_ptr() + b_off * B.element_size()
        d.C = Cm.data_ptr() + c_off * esC
        d.M, d.N, d.K = M, N, K
        d.a_sm, d.a_sk, d.b_sk, d.b_sn, d.ldc = a_sm, a_sk, b_sk, b_sn, ldc
        d.nbatch, d.a_sb, d.b_sb, d.c_sb = nbatch, a_sb, b_sb, c_sb
        d.kbatch, d.a_skb, d.b_skb = kbatch, a_skb, b_skb
        d.bias = ptr(bias)
        d.bias_sb = bias_sb
        d.scale_cols, d.scale = scale_cols, scale
        d.accumulate = 1 if accumulate else 0
        d.act = act
        d.aux_out = None if aux_out is None else aux_out.data_ptr() + c_off * esC
        d.aux_in = None if aux_in is None else aux_in.data_ptr() + c_off * esC
        d.resid = ptr(resid)
        d.r_ld, d.r_sb = r_ld, r_sb
        d.splitk = splitk
        d.dropout_p = dropout_p
        d.nbatch2, d.a_sb2, d.b_sb2, d.c_sb2 = nbatch2, a_sb2, b_sb2, c_sb2
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        assert A.dtype == B.dtype
        d.in_dtype, d.out_dtype = dt(A), dt(Cm)
        if len(_GEMM_DESCS) > 8192:   # (ragged batches / many models in one process: start over rather than grow without bound)
            _GEMM_DESCS.clear()
        hit = _GEMM_DESCS[key] = (d, C.byref(d))
    d, ref = hit
    if dropout_p > 0.0:
        d.dropout_seed = dropout_seed
    if PROFILE is None:
        check(lib().tmi_gemm(ref, stream()), "tmi_gemm")
        return
    with _probe("gemm", 2.0 * M * N * K * max(1, nbatch) * max(1, kbatch) * max(1, nbatch2)):
        check(lib().tmi_gemm(ref, stream()), "tmi_gemm")


def linear(x2d, w, out, *, w_is_kn=True, **kw):
    """out[M,N] = x2d[M,K] @ W with W given as [K,N] row-major (Keras Dense kernel layout)
    when ``w_is_kn`` else as [N,K] row-major."""
    M, K = x2d.shape
    if w_is_kn:
        N = w.shape[1]
        b_sk, b_sn = w.stride(0), w.stride(1)
    else:
        N = w.shape[0]
        b_sk, b_sn = w.stride(1), w.stride(0)
    gemm(x2d, w, out, M, N, K, x2d.stride(0), x2d.stride(1), b_sk, b_sn, out.stride(0), **kw)


def layernorm_fwd(x2d, gamma, beta, y2d, mean, rstd, eps):
    with _probe("layernorm", 2.0 * x2d.numel() * x2d.element_size()):
        rows, Cn = x2d.shape
        check(lib().tmi_layernorm_fwd(x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y2d.data_ptr(),
                                      mean.data_ptr(), rstd.data_ptr(), rows, Cn, eps, dt(x2d), stream()),
              "tmi_layernorm_fwd")


_LN_WS = {}
# tmi_layernorm_bwd's column sums: fp32 atomics (default) or per-workgroup partial rows in a workspace + a fixed-order
# fold launch (TMI_LN_DETERMINISTIC=1: bit-reproducible dgamma / dbeta / bias sums).  Measured on MI355X, round 3
# (profiles/r03_ln_bwd_forms.txt): the two forms take the same time alone (23-24 us at [12000, 768] bf16) AND beside the
# weight-gradient stream (88-108 us either way: the kernel is starved of CUs by the one-workgroup-per-CU GEMMs, it is
# not serialising on its atomics), and the extra launch per call costs the step 0.07-0.1 ms - so atomics stay the default.
LN_ATOMIC = os.environ.get("TMI_LN_DETERMINISTIC", "0") == "0"


def set_deterministic(on: bool) -> bool:
    """Reproducible reductions for the whole step (returns the previous setting): LayerNorm backward through its workspace +
    fixed-order fold, column sums with one workgroup per column group (tmi_set_deterministic).  Every other kernel of the
    Whisper step is order-independent already, so two runs of the same steps then agree bit for bit; costs ~0.1 ms/step."""
    global LN_ATOMIC
    was = not LN_ATOMIC
    LN_ATOMIC = not on
    lib().tmi_set_deterministic(1 if on else 0)
    return was


def ln_bwd_workspace(device, rows, C, emit):
    """Partial-sum table of tmi_layernorm_bwd (one row of 2-3 * C floats per workgroup): one per (device, stream) like the
    GEMM workspace - the fold launch that reads it follows on the same stream.  An outgrown buffer is parked, never
    returned to the allocator while queued kernels may read it (see attn_workspace)."""
    if LN_ATOMIC:
        return None
    need = int(lib().tmi_layernorm_bwd_workspace_bytes(rows, C, 1 if emit else 0))
    key = (device.type, device.index, stream())
    ws = _LN_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            _ATTN_WS_RETIRED.append(ws)
        ws = _LN_WS[key] = torch.empty(max(need // 4, 1 << 20), dtype=torch.float32, device=device)
    return ws


def layernorm_dropout_fwd(x2d, gamma, beta, y2d, mean, rstd, eps, dropout_p, dropout_seed):
    """y = Dropout_p(LayerNorm(x)) in one pass (tmi_layernorm_dropout_fwd): the mask of ``dropout`` over the [rows, C] output."""
    with _probe("layernorm", 2.0 * x2d.numel() * x2d.element_size()):
        rows, Cn = x2d.shape
        check(lib().tmi_layernorm_dropout_fwd(x2d.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y2d.data_ptr(), mean.data_ptr(),
                                              rstd.data_ptr(), rows, Cn, eps, dropout_p, dropout_seed, dt(x2d), stream()),
              "tmi_layernorm_dropout_fwd")


def layernorm_dropout_bwd(dy2d, x2d, gamma, mean, rstd, dx2d, dgamma, dbeta, dropout_p, dropout_seed, accumulate_dx=False):
    """layernorm_bwd of a LayerNorm whose output went through Dropout: the same mask (same seed) applied to dy on load."""
    with _probe("layernorm", (4.0 if accumulate_dx else 3.0) * x2d.numel() * x2d.element_size()):
        rows, Cn = x2d.shape
        ws = ln_bwd_workspace(x2d.device, rows, Cn, False)
        check(lib().tmi_layernorm_dropout_bwd(dy2d.data_ptr(), x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                              dx2d.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, Cn,
                                              1 if accumulate_dx else 0, dropout_p, dropout_seed, ptr(ws),
                                              0 if ws is None else ws.numel() * 4, dt(x2d), stream()), "tmi_layernorm_dropout_bwd")


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, dx2d, dgamma, dbeta, accumulate_dx=False):
    """dgamma / dbeta are accumulated into: zero them first (the grad arena is)."""
    with _probe("layernorm", (4.0 if accumulate_dx else 3.0) * x2d.numel() * x2d.element_size()):
        rows, Cn = x2d.shape
        ws = ln_bwd_workspace(x2d.device, rows, Cn, False)
        check(lib().tmi_layernorm_bwd(dy2d.data_ptr(), x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), dx2d.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, Cn,
                                      1 if accumulate_dx else 0, ptr(ws), 0 if ws is None else ws.numel() * 4, dt(x2d), stream()),
              "tmi_layernorm_bwd")


def layernorm_bwd_emit(dy2d, x2d, gamma, mean, rstd, dx2d, dgamma, dbeta, colsum, masked=None, dropout_p=0.0, dropout_seed=0,
                       accumulate_dx=False):
    """layernorm_bwd that also accumulates colsum[c] += sum_rows dy' and (with ``masked``) writes dy' = Dropout-mask(dx):
    what the Dense layer below needs of dx (tmi_layernorm_bwd_emit)."""
    es = x2d.element_size()
    work = ((4.0 if accumulate_dx else 3.0) + (1.0 if masked is not None and dropout_p > 0 else 0.0)) * x2d.numel() * es
    with _probe("layernorm", work):
        rows, Cn = x2d.shape
        ws = ln_bwd_workspace(x2d.device, rows, Cn, True)
        check(lib().tmi_layernorm_bwd_emit(dy2d.data_ptr(), x2d.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           dx2d.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, Cn,
                                           1 if accumulate_dx else 0, colsum.data_ptr(), ptr(masked), dropout_p, dropout_seed,
                                           ptr(ws), 0 if ws is None else ws.numel() * 4, dt(x2d), stream()), "tmi_layernorm_bwd_emit")


def bias_grad(dy2d, dbias):
    """dbias[N] += sum over rows of dy2d[rows, N] (atomics: zero dbias first)."""
    with _probe("colsum", 1.0 * dy2d.numel() * dy2d.element_size()):
        rows, N = dy2d.shape
        check(lib().tmi_colsum(dy2d.data_ptr(), dy2d.stride(0), dbias.data_ptr(), rows, N, dt(dy2d), stream()),
              "tmi_colsum")


def bias_grad_batched(dy3d, dbias, out_sb):
    """dy3d [L, rows, N] (constant layer stride) -> dbias + l * out_sb (elements) += column sums of dy3d[l], one launch."""
    with _probe("colsum", 1.0 * dy3d.numel() * dy3d.element_size()):
        L, rows, N = dy3d.shape
        check(lib().tmi_colsum_batched(dy3d.data_ptr(), dy3d.stride(1), dy3d.stride(0), dbias.data_ptr(), out_sb, rows, N, L,
                                       dt(dy3d), stream()), "tmi_colsum_batched")


def gelu_bwd(dy, u, dx):
    check(lib().tmi_gelu_bwd(dy.data_ptr(), u.data_ptr(), dx.data_ptr(), dy.numel(), dt(dy), stream()),
          "tmi_gelu_bwd")


def gelu_bwd_batched(dy, u, dx, n, nbatch, dy_sb, u_sb, dx_sb):
    """nbatch spans of n elements, batch strides in elements (tensors give the base pointers)."""
    check(lib().tmi_gelu_bwd_batched(dy.data_ptr(), u.data_ptr(), dx.data_ptr(), n, nbatch, dy_sb, u_sb, dx_sb, dt(dy),
                                     stream()), "tmi_gelu_bwd")


def softmax_fwd(s, rows, Tq, Tk, mask_mode):
    check(lib().tmi_softmax_fwd(s.data_ptr(), rows, Tq, Tk, mask_mode, stream()), "tmi_softmax_fwd")


def softmax_bwd(p, dp, rows, Tk):
    check(lib().tmi_softmax_bwd(p.data_ptr(), dp.data_ptr(), rows, Tk, stream()), "tmi_softmax_bwd")


_ATTN_WS = {}
_ATTN_WS_RETIRED = []


def attn_workspace(device, B, H, Tq):
    """Key-split scratch of tmi_attn_fwd / tmi_attn_bwd (cross-attention: one query tile against a long key side): one per
    (device, stream) like the GEMM workspace, grown to the largest (B, H, Tq) seen.  None when the shape never splits."""
    need = int(lib().tmi_attn_workspace_bytes(B, H, Tq))
    if need == 0:
        return None
    key = (device.type, device.index, stream())
    ws = _ATTN_WS.get(key)
    if ws is None or ws.numel() < need:
        # Kernels are launched on the ops-pinned stream, which is not torch's current stream, so the caching allocator
        # cannot know that a dropped buffer may still be read by queued key-split / combine kernels: a buffer that is
        # outgrown is parked (never handed back while the process lives; growth happens a handful of times per model)
        if ws is not None:
            _ATTN_WS_RETIRED.append(ws)
        ws = _ATTN_WS[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return ws


def _attn_desc(q, k, v, o, stats, B, H, Tq, Tk, mask_mode, score_scale=1.0):
    """q,k,v,o: (tensor, element_offset, batch_stride, token_stride)."""
    d = AttnDesc()
    if mask_mode == 0 and Tq <= 128 and Tk >= 512:
        ws = attn_workspace(stats.device, B, H, Tq)
        if ws is not None:
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    for name, (t, off, sb, st) in (("q", q), ("k", k), ("v", v), ("o", o)):
        setattr(d, name, t.data_ptr() + off * t.element_size())
        setattr(d, f"{name}_sb", sb)
        setattr(d, f"{name}_st", st)
    d.stats = stats.data_ptr()
    d.B, d.H, d.Tq, d.Tk = B, H, Tq, Tk
    d.mask_mode = mask_mode
    d.score_scale = score_scale
    return d


def attn_dropmask(device, B, H, Tq, Tk):
    """Buffer for the keep bits of one attention call with dropout (tmi_attn_desc.drop_mask): written by the forward, read by
    the backward of the same call, so the caller keeps one per dropout site from forward to backward."""
    return torch.empty(int(lib().tmi_attn_dropmask_bytes(B, H, Tq, Tk)), dtype=torch.uint8, device=device)


def _set_dropout(d, dropout_p, dropout_seed, drop_mask):
    d.dropout_p, d.dropout_seed = dropout_p, dropout_seed
    if dropout_p > 0.0:
        if drop_mask is None:
            raise ValueError("attention dropout needs a drop_mask buffer (ops.attn_dropmask): the forward stores the mask, "
                             "the backward reads it")
        d.drop_mask, d.drop_mask_bytes = drop_mask.data_ptr(), drop_mask.numel() * drop_mask.element_size()


def attn_fwd(q, k, v, o, stats, B, H, Tq, Tk, mask_mode=0, score_scale=1.0, dropout_p=0.0, dropout_seed=0, drop_mask=None):
    with _probe("attention", 4.0 * B * H * Tq * Tk * 64):
        d = _attn_desc(q, k, v, o, stats, B, H, Tq, Tk, mask_mode, score_scale)
        _set_dropout(d, dropout_p, dropout_seed, drop_mask)
        check(lib().tmi_attn_fwd(C.byref(d), stream()), "tmi_attn_fwd")


def attn_bwd(q, k, v, o, stats, do, dq, dk, dv, delta, B, H, Tq, Tk, mask_mode=0, dq_scale=1.0, score_scale=1.0,
             dropout_p=0.0, dropout_seed=0, passes=0, drop_mask=None):
    """``passes``: 0 both, 1 the dQ pass (fills ``delta``), 2 the dK/dV pass (needs ``delta`` from pass 1).
    ``drop_mask``: the buffer the forward of this call filled (dropout_p > 0)."""
    d = _attn_desc(q, k, v, o, stats, B, H, Tq, Tk, mask_mode, score_scale)
    _set_dropout(d, dropout_p, dropout_seed, drop_mask)
    d.bwd_passes = passes
    for name, field, (t, off, sb, st) in (("d_o", "do", do), ("dq", "dq", dq), ("dk", "dk", dk), ("dv", "dv", dv)):
        setattr(d, name, t.data_ptr() + off * t.element_size())
        setattr(d, f"{field}_sb", sb)
        setattr(d, f"{field}_st", st)
    d.delta = delta.data_ptr()
    d.dq_scale = dq_scale
    # algorithmic work as SURVEY 8(d) defines it: fwd + bwd = 3 x forward, i.e. backward = 2 x 4.B.H.Tq.Tk.64
    # (the two passes execute seven products: both recompute S and dP)
    # (3 of the 7 products are the dQ pass, 4 the dK/dV pass)
    with _probe("attention", 8.0 * B * H * Tq * Tk * 64 * (1.0 if passes in (0, 3) else (3.0 / 7.0 if passes == 1 else 4.0 / 7.0))):
        check(lib().tmi_attn_bwd(C.byref(d), stream()), "tmi_attn_bwd")


def fill_zero(t):
    """t[...] = 0 on the launch stream through the library (tmi_memset_async / tmi_memset2d_async), so that the fill is part
    of a recorded launch plan: a torch ``zero_()`` between two recorded launches would be missing from every replay.
    Contiguous tensors, or views whose slices along dim 0 are contiguous (the pad rows of a [B, T, C] buffer)."""
    es = t.element_size()
    if t.is_contiguous():
        check(lib().tmi_memset_async(t.data_ptr(), 0, t.numel() * es, stream()), "tmi_memset_async")
        return
    if t.dim() < 2 or not t[0].is_contiguous():
        raise ValueError("fill_zero: contiguous tensor or a view with contiguous slices along dim 0")
    if t.numel() == 0:
        return
    check(lib().tmi_memset2d_async(t.data_ptr(), t.stride(0) * es, 0, t[0].numel() * es, t.shape[0], stream()),
          "tmi_memset2d_async")


def copy(dst, src):
    """dst[...] = src[...] (same dtype and element count, both contiguous) on the launch stream, through the library."""
    if dst.dtype != src.dtype or dst.numel() != src.numel() or not dst.is_contiguous() or not src.is_contiguous():
        raise ValueError("copy: contiguous tensors of one dtype and size")
    check(lib().tmi_memcpy_async(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), stream()), "tmi_memcpy_async")


def dropout(x, out, rows, cols, p, seed, resid=None):
    """out = (resid or 0) + Dropout_p(x) over [rows, cols] (row strides from the tensors); the same call on the
    incoming gradient, with the same seed, is the backward.  See tmi_dropout in include/tethys_mi.h."""
    if PROFILE is not None:
        with _probe("dropout", (3.0 if resid is not None else 2.0) * rows * cols * x.element_size()):
            check(lib().tmi_dropout(x.data_ptr(), x.stride(0), ptr(resid), resid.stride(0) if resid is not None else 0,
                                    out.data_ptr(), out.stride(0), rows, cols, p, seed, dt(x), stream()), "tmi_dropout")
        return
    check(lib().tmi_dropout(x.data_ptr(), x.stride(0), ptr(resid), resid.stride(0) if resid is not None else 0,
                            out.data_ptr(), out.stride(0), rows, cols, p, seed, dt(x), stream()), "tmi_dropout")


def embed_fwd(labels, table, pe, out, B, S, D, start_id):
    check(lib().tmi_embed_fwd(labels.data_ptr(), table.data_ptr(), pe.data_ptr(), out.data_ptr(), B, S, D,
                              start_id, dt(out), stream()), "tmi_embed_fwd")


def embed_bwd(labels, dy, dtable, B, S, D, start_id):
    check(lib().tmi_embed_bwd(labels.data_ptr(), dy.data_ptr(), dtable.data_ptr(), B, S, D, start_id,
                              dt(dy), stream()), "tmi_embed_bwd")


def xent_fwd_bwd(logits, ld, labels, row_loss, B, S, V, grad_scale, lm=None):
    """``lm = (x, x_ld, w, w_sk, w_sn, d)``: the logits are the LM head's output x . w - tmi_linear_xent takes the target logit of
    the loss from those operands in fp32 instead of from the bf16-rounded logits."""
    # (the kernel holds a row on chip: ONE read and one write of the logits, csrc/softmax_xent.hip - not the three passes of an
    # unfused softmax + loss + gradient)
    with _probe("xent", 2.0 * B * S * V * logits.element_size()):
        if lm is None:
            check(lib().tmi_xent_fwd_bwd(logits.data_ptr(), ld, labels.data_ptr(), row_loss.data_ptr(), B, S, V,
                                         grad_scale, dt(logits), stream()), "tmi_xent_fwd_bwd")
        else:
            x, x_ld, w, w_sk, w_sn, d = lm
            check(lib().tmi_linear_xent(x.data_ptr(), x_ld, w.data_ptr(), w_sk, w_sn, d, logits.data_ptr(), ld, labels.data_ptr(),
                                        row_loss.data_ptr(), B, S, V, grad_scale, dt(logits), stream()), "tmi_linear_xent")


def sum_scale(x, out, n, scale):
    check(lib().tmi_sum_scale(x.data_ptr(), out.data_ptr(), n, scale, stream()), "tmi_sum_scale")


def adam_step(p, g, m, v, n, lr, beta1, beta2, eps, step, eps_mode=0, weight_decay=0.0, gscale=1.0, mirror=None,
              zero_grad=False, max_blocks=0):
    with _probe("adam", (28.0 + (2.0 if mirror is not None else 0.0)) * n):
        check(lib().tmi_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2,
                                  eps, step, eps_mode, weight_decay, gscale, ptr(mirror), 1 if zero_grad else 0,
                                  max_blocks, stream()), "tmi_adam_step")


def segment_chunks(seg_off, chunk=8192, device=None):
    """(lo, hi, variable) pieces of at most ``chunk`` elements that tile the variables of ``seg_off`` (a host list or
    tensor of nseg + 1 element offsets): the table tmi_adam_step_segments walks."""
    offs = [int(x) for x in (seg_off.tolist() if hasattr(seg_off, "tolist") else seg_off)]
    rows = []
    for s_, (lo, hi) in enumerate(zip(offs[:-1], offs[1:])):
        a = lo
        while a < hi:
            b = min(hi, a + chunk)
            rows.append((a, b, s_))
            a = b
    return torch.tensor(rows, dtype=torch.int64, device=device).reshape(-1, 3)


def adam_step_segments(p, g, m, v, n, chunks, sumsq, nseg, clip_global, clip_each, lr, beta1, beta2, eps, step, eps_mode=0,
                       weight_decay=0.0, gscale=1.0, mirror=None, zero_grad=False, max_blocks=0):
    """``chunks`` may be a row range of the model's table (absolute arena offsets): the same update, restricted to those rows;
    ``n`` = the elements they cover (probe accounting only)."""
    with _probe("adam", (28.0 + (2.0 if mirror is not None else 0.0)) * n):
        check(lib().tmi_adam_step_segments(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), chunks.data_ptr(),
                                           chunks.shape[0], ptr(sumsq), nseg, clip_global, clip_each, lr, beta1, beta2, eps, step, eps_mode,
                                           weight_decay, gscale, ptr(mirror), 1 if zero_grad else 0, max_blocks, stream()),
              "tmi_adam_step_segments")


def adam_step_rows(p, g, m, v, nrows, row_len, active, lr, beta1, beta2, eps, step, eps_mode=0, gscale=1.0, mirror=None,
                   zero_grad=False):
    """Adam over an embedding table, idle rows skipped (tmi_adam_step_rows).  Its own probe class: the ALGORITHMIC bytes are the
    dense kernel's (28 B/param whatever the kernel skips), but it moves only the rows with a gradient or a
    live moment (known on the device, not here) - bench.py reports the "adam" class from the dense launches (bytes really
    moved) and the algorithmic figure, rows included, beside it."""
    with _probe("adam_rows", (28.0 + (2.0 if mirror is not None else 0.0)) * nrows * row_len):
        check(lib().tmi_adam_step_rows(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), nrows, row_len,
                                       active.data_ptr(), lr, beta1, beta2, eps, step, eps_mode, 0.0, gscale, ptr(mirror),
                                       1 if zero_grad else 0, stream()), "tmi_adam_step_rows")


def adam_scalars(lr, beta1, beta2, step, eps_mode=0, weight_decay=0.0):
    """[step_size, vcorr_inv_sqrt, decay] of one Adam step (host floats, as tmi_adam_step derives them)."""
    out = (C.c_float * 3)()
    check(lib().tmi_adam_scalars(lr, beta1, beta2, step, eps_mode, weight_decay, C.cast(out, C.c_void_p)), "tmi_adam_scalars")
    return [out[0], out[1], out[2]]


def adam_step_dev(p, g, m, v, n, beta1, beta2, eps, dev_scalars, eps_mode=0, gscale=1.0, mirror=None):
    check(lib().tmi_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, beta1, beta2, eps,
                                  dev_scalars.data_ptr(), eps_mode, gscale, ptr(mirror), stream()), "tmi_adam_step_dev")


def grad_pack(src, dst, n, scale=1.0):
    check(lib().tmi_grad_pack(src.data_ptr(), dst.data_ptr(), n, scale, stream()), "tmi_grad_pack")


def grad_unpack(src, dst, n, nparts=1, part_stride=0, scale=1.0):
    check(lib().tmi_grad_unpack(src.data_ptr(), dt(src), nparts, part_stride, dst.data_ptr(), n, scale, stream()),
          "tmi_grad_unpack")


def cast_bf16(src, lds, dst, ldd, rows, cols, src_off=0, dst_off=0):
    check(lib().tmi_cast_bf16(src.data_ptr() + 4 * src_off, lds, dst.data_ptr() + 2 * dst_off, ldd, rows, cols,
                              stream()), "tmi_cast_bf16")


def transpose_cast_bf16(src, lds, dst, ldd, rows, cols, src_off=0, dst_off=0):
    check(lib().tmi_transpose_cast_bf16(src.data_ptr() + 4 * src_off, lds, dst.data_ptr() + 2 * dst_off, ldd,
                                        rows, cols, stream()), "tmi_transpose_cast_bf16")


def feat_to_channels_last(feats, out, B, Cn, T, pad_left, pad_right):
    check(lib().tmi_feat_to_channels_last(feats.data_ptr(), out.data_ptr(), B, Cn, T, pad_left, pad_right,
                                          dt(out), stream()), "tmi_feat_to_channels_last")


def sumsq(x, out, n, accumulate=False):
    check(lib().tmi_sumsq(x.data_ptr(), out.data_ptr(), n, 1 if accumulate else 0, stream()), "tmi_sumsq")


# ---------------------------------------------------------------- Wav2Vec2 operators
def groupnorm_chunks(T: int) -> int:
    return int(lib().tmi_groupnorm_chunks(T))


def groupnorm_gelu_fwd(x, x_sb, gamma, beta, y, y_sb, stats, part, B, T, Cn, G, eps=1e-5, x_off=0, y_off=0):
    # algorithmic bytes: statistics need the whole (time, C/G) slab before any output, so 2 reads + 1 write
    with _probe("groupnorm", 3.0 * B * T * Cn * x.element_size()):
        check(lib().tmi_groupnorm_gelu_fwd(x.data_ptr() + x_off * x.element_size(), x_sb, gamma.data_ptr(), beta.data_ptr(),
                                           y.data_ptr() + y_off * y.element_size(), y_sb, stats.data_ptr(), part.data_ptr(),
                                           B, T, Cn, G, eps, dt(x), stream()), "tmi_groupnorm_gelu_fwd")
    return
    check(lib().tmi_groupnorm_gelu_fwd(x.data_ptr() + x_off * x.element_size(), x_sb, gamma.data_ptr(), beta.data_ptr(),
                                       y.data_ptr() + y_off * y.element_size(), y_sb, stats.data_ptr(), part.data_ptr(),
                                       B, T, Cn, G, eps, dt(x), stream()), "tmi_groupnorm_gelu_fwd")


def groupnorm_gelu_bwd(x, x_sb, dy, dy_sb, gamma, beta, stats, dx, dx_sb, dgamma, dbeta, part, sums, B, T, Cn, G,
                       x_off=0, dy_off=0, dx_off=0):
    es = x.element_size()
    # algorithmic bytes: x and dy are each read twice (sums, then dx), dx written once
    with _probe("groupnorm", 5.0 * B * T * Cn * es):
        check(lib().tmi_groupnorm_gelu_bwd(x.data_ptr() + x_off * es, x_sb, dy.data_ptr() + dy_off * es, dy_sb,
                                           gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                           dx.data_ptr() + dx_off * es, dx_sb, dgamma.data_ptr(), dbeta.data_ptr(),
                                           part.data_ptr(), sums.data_ptr(), B, T, Cn, G, dt(x), stream()),
              "tmi_groupnorm_gelu_bwd")
    return
    check(lib().tmi_groupnorm_gelu_bwd(x.data_ptr() + x_off * es, x_sb, dy.data_ptr() + dy_off * es, dy_sb,
                                       gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                       dx.data_ptr() + dx_off * es, dx_sb, dgamma.data_ptr(), dbeta.data_ptr(),
                                       part.data_ptr(), sums.data_ptr(), B, T, Cn, G, dt(x), stream()),
          "tmi_groupnorm_gelu_bwd")


def fir_chunks(T: int) -> int:
    return int(lib().tmi_fir_chunks(T))


def fir_gn_workspace_floats(B, T, Cn):
    return int(lib().tmi_fir_gn_workspace_floats(B, T, Cn))


def fir_groupnorm_gelu_fwd(audio, pad_left, w, k, stride, gamma, beta, y, y_sb, stats, part, B, T, Cn, G, eps=1e-5, y_off=0):
    """Conv layer 0 as a filter bank + GroupNorm + GELU (tmi_fir_groupnorm_gelu_fwd): audio fp32 [B, Tin]."""
    # algorithmic bytes: the output written once (the audio is ~1/250 of it)
    with _probe("groupnorm", 1.0 * B * T * Cn * y.element_size()):
        check(lib().tmi_fir_groupnorm_gelu_fwd(audio.data_ptr(), audio.stride(0), audio.shape[1], pad_left, w.data_ptr(), k, stride,
                                               gamma.data_ptr(), beta.data_ptr(), y.data_ptr() + y_off * y.element_size(), y_sb,
                                               stats.data_ptr(), part.data_ptr(), B, T, Cn, G, eps, dt(y), stream()),
              "tmi_fir_groupnorm_gelu_fwd")


def fir_groupnorm_gelu_bwd(audio, pad_left, w, k, stride, dy, dy_sb, gamma, beta, stats, dW, dgamma, dbeta, part, sums, wpart,
                           B, T, Cn, G, dy_off=0):
    with _probe("groupnorm", 2.0 * B * T * Cn * dy.element_size()):  # dy read by both passes
        check(lib().tmi_fir_groupnorm_gelu_bwd(audio.data_ptr(), audio.stride(0), audio.shape[1], pad_left, w.data_ptr(), k, stride,
                                               dy.data_ptr() + dy_off * dy.element_size(), dy_sb, gamma.data_ptr(), beta.data_ptr(),
                                               stats.data_ptr(), dW.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                               part.data_ptr(), sums.data_ptr(), wpart.data_ptr(), B, T, Cn, G, dt(dy), stream()),
              "tmi_fir_groupnorm_gelu_bwd")


def group_pack(x, xg, B, T, Cn, G, Tp, pad_left):
    check(lib().tmi_group_pack(x.data_ptr(), xg.data_ptr(), B, T, Cn, G, Tp, pad_left, dt(x), stream()), "tmi_group_pack")


def group_unpack(yg, bias, resid, out, B, T, Cn, G, Tp, row_off):
    check(lib().tmi_group_unpack(yg.data_ptr(), ptr(bias), ptr(resid), out.data_ptr(), B, T, Cn, G, Tp, row_off,
                                 dt(out), stream()), "tmi_group_unpack")


def posconv_pack_weights(w, wf, wb, k, Cg, G, w_off=0):
    check(lib().tmi_posconv_pack_weights(w.data_ptr() + 4 * w_off, wf.data_ptr(), wb.data_ptr(), k, Cg, G, dt(wf),
                                         stream()), "tmi_posconv_pack_weights")


def vq_assign(codebook, idx, q, perplexity, rows, G, Nc, gd):
    check(lib().tmi_vq_assign(codebook.data_ptr(), idx.data_ptr(), q.data_ptr(), perplexity.data_ptr(), rows, G, Nc, gd,
                              dt(q), stream()), "tmi_vq_assign")


def vq_nearest(h, codebook, idx, q, perplexity, rows, G, Nc, gd):
    check(lib().tmi_vq_nearest(h.data_ptr(), codebook.data_ptr(), idx.data_ptr(), q.data_ptr(), perplexity.data_ptr(),
                               rows, G, Nc, gd, dt(h), stream()), "tmi_vq_nearest")


def vq_bwd(idx, dq, dcodebook, rows, G, Nc, gd):
    check(lib().tmi_vq_bwd(idx.data_ptr(), dq.data_ptr(), dcodebook.data_ptr(), rows, G, Nc, gd, dt(dq), stream()),
          "tmi_vq_bwd")


def contrastive_fwd_bwd(S, neg, row_loss, B, T, Nn, temperature, grad_scale, per_time=False):
    """``neg`` [B, Nn] (one row of indices per batch row, V:908-937) or, with ``per_time``, [T, Nn] (one row per
    time step shared by the batch, whisper_single.py:789-839)."""
    sb, st = (0, Nn) if per_time else (Nn, 0)
    check(lib().tmi_contrastive_fwd_bwd(S.data_ptr(), neg.data_ptr(), sb, st, row_loss.data_ptr(), B, T, Nn, temperature,
                                        grad_scale, stream()), "tmi_contrastive_fwd_bwd")


def segment_sumsq_chunks(g, chunks, out, nseg):
    """Per-variable sums of squares over the chunk table of ``segment_chunks`` (tmi_segment_sumsq_chunks)."""
    check(lib().tmi_segment_sumsq_chunks(g.data_ptr(), chunks.data_ptr(), chunks.shape[0], out.data_ptr(), nseg, stream()),
          "tmi_segment_sumsq_chunks")


def segment_sumsq(g, seg_off, out, nseg):
    check(lib().tmi_segment_sumsq(g.data_ptr(), seg_off.data_ptr(), out.data_ptr(), nseg, stream()), "tmi_segment_sumsq")


def segment_clip(g, seg_off, sumsq, nseg, clip):
    check(lib().tmi_segment_clip(g.data_ptr(), seg_off.data_ptr(), sumsq.data_ptr(), nseg, clip, stream()),
          "tmi_segment_clip")


def loss_combine(a, b, w, scale, out):
    check(lib().tmi_loss_combine(a.data_ptr(), b.data_ptr(), w, scale, out.data_ptr(), stream()), "tmi_loss_combine")
