"""Building blocks shared by the Whisper and Wav2Vec2 step orchestration: weight access
(fp32 master / bf16 mirror), workspace cache, Dense / LayerNorm / attention forward and
backward expressed as C-ABI calls.  A model class supplies ``arena``, ``precision``, ``dtype``,
``device``, ``mirror``, ``ws`` and ``hidden`` (the attention model width)."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Arena:
    """All trainable parameters in ONE flat fp32 buffer (plus same-shaped grad / Adam m / v
    buffers).  ``spec`` lists (name, storage shape) in forward order; ``logical`` gives the
    reference shape of tensors stored padded; fused ``*.qkv.*`` / ``*.kv.*`` tensors (column
    blocks of width ``fuse_width``) are exposed under the reference's q_proj / k_proj / v_proj
    names by ``ref_views``."""

    def __init__(self, spec, device, logical=None, fuse_width=None):
        self.logical = dict(logical or {})
        self.fuse_width = fuse_width
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        off = 0
        for name, shape in spec:
            self.offsets[name] = off
            self.shapes[name] = shape
            off += _round_up(int(np.prod(shape)), 8)  # 16-byte alignment in the bf16 mirror too
        self.numel = off
        self.n_params = sum(int(np.prod(self.logical.get(n, s))) for n, s in self.shapes.items())
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.names = [n for n, _ in spec]

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        # views are cached per (buffer, name): the step asks for ~400 of them and building one costs
        # more host time than launching a decoder-sized kernel
        cache = self.__dict__.setdefault("_views", {})
        key = (buf.data_ptr(), buf.dtype, name)
        t = cache.get(key)
        if t is None:
            o, s = self.offsets[name], self.shapes[name]
            t = cache[key] = buf[o:o + int(np.prod(s))].view(*s)
        return t

    def param(self, name):
        return self.view(self.p, name)

    def grad(self, name):
        return self.view(self.g, name)

    def ref_views(self, buf: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Views of ``buf`` keyed by the reference's variable paths (oracle.param_shapes)."""
        d = self.fuse_width
        out: Dict[str, torch.Tensor] = {}
        for name in self.names:
            t = self.view(buf, name)
            if name.endswith(".qkv.kernel"):
                base = name[:-len("qkv.kernel")]
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[f"{base}{n}.kernel"] = t[:, j * d:(j + 1) * d]
            elif name.endswith(".qkv.bias"):
                base = name[:-len("qkv.bias")]
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[f"{base}{n}.bias"] = t[j * d:(j + 1) * d]
            elif name.endswith(".cross_kv.kernel") or name.endswith(".cross_kv.bias"):
                # the k/v projections of EVERY decoder layer side by side: columns [(2 i + j) d, +d) belong
                # to layer i's k_proj (j = 0) / v_proj (j = 1)
                base, leaf = name.rsplit("cross_kv.", 1)
                for i in range(t.shape[-1] // (2 * d)):
                    for j, n in enumerate(("k_proj", "v_proj")):
                        out[f"{base}layers.{i}.encoder_attn.{n}.{leaf}"] = t[..., (2 * i + j) * d:(2 * i + j + 1) * d]
            elif name.endswith(".kv.kernel"):
                base = name[:-len("kv.kernel")]
                for j, n in enumerate(("k_proj", "v_proj")):
                    out[f"{base}{n}.kernel"] = t[:, j * d:(j + 1) * d]
            elif name.endswith(".kv.bias"):
                base = name[:-len("kv.bias")]
                for j, n in enumerate(("k_proj", "v_proj")):
                    out[f"{base}{n}.bias"] = t[j * d:(j + 1) * d]
            elif name.endswith(".qkv3.kernel") or name.endswith(".qkv3.bias"):
                # three separate, contiguous blocks [3][...]: one reference variable each
                base, leaf = name.rsplit("qkv3.", 1)
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    out[f"{base}{n}.{leaf}"] = t[j]
            elif name in self.logical:
                out[name] = t[tuple(slice(0, n) for n in self.logical[name])]
            else:
                out[name] = t
        return out

    def load_ref(self, params: Dict[str, torch.Tensor]):
        """Copy a reference-keyed dict (e.g. the oracle's init_params) into the arena."""
        views = self.ref_views(self.p)
        missing = set(views) ^ set(params)
        if missing:
            raise KeyError(f"parameter name mismatch: {sorted(missing)[:4]} ...")
        for k, v in views.items():
            v.copy_(params[k].to(torch.float32))

    def variable_segments(self) -> List[Tuple[int, int]]:
        """[start, end) arena ranges of the reference's variables (contiguous ones only): what
        Keras ``clipnorm`` clips one by one."""
        segs = []
        base_ptr = self.p.data_ptr()
        for k, v in self.ref_views(self.p).items():
            if not v.is_contiguous():
                raise ValueError(f"{k} is not contiguous in the arena")
            lo = (v.data_ptr() - base_ptr) // 4
            segs.append((lo, lo + v.numel()))
        return sorted(segs)

    def init_keras_defaults(self, seed: int = 1234):
        """Keras default initialisers (SURVEY a-14): glorot-uniform kernels (per reference
        tensor, so each fused q/k/v slice uses fan_in+fan_out of its own [d,d] kernel), zero
        biases, Embedding U(-0.05, 0.05), LayerNorm gamma=1 / beta=0.  Drawn on the arena's own device (the reference is
        unseeded; a short job's JCT is mostly model construction: 148 M host-side draws + copies took 2.4 of its 3.5 s)."""
        gen = torch.Generator(device=self.p.device).manual_seed(seed)
        dev = self.p.device
        for name, v in self.ref_views(self.p).items():
            shape = tuple(v.shape)
            if name.endswith(".kernel"):
                if len(shape) == 3:
                    fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
                else:
                    fan_in, fan_out = shape
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                v.copy_((torch.rand(shape, generator=gen, device=dev) * 2 - 1) * lim)
            elif name.endswith(".embeddings"):
                v.copy_((torch.rand(shape, generator=gen, device=dev) * 2 - 1) * 0.05)
            elif name.endswith("codevectors"):
                v.copy_(torch.randn(shape, generator=gen, device=dev))  # tf.random.normal (V:571)
            elif name.endswith(".gamma"):
                v.fill_(1.0)
            else:
                v.zero_()


_DEFER_WGRAD = os.environ.get("TMI_DEFER_WGRAD", "0") != "0"
_DKV_ON_SIDE = os.environ.get("TMI_CROSS_DKV_SIDE", "1") != "0"


class KernelBlocks:
    # ---- weight-gradient stream ------------------------------------------------------------
    # dW = xᵀ·dy and db = colsum(dy) feed nothing until the optimizer (or the all-reduce), so they run
    # on a second HIP stream beside the dgrad chain: many kernels of the chain (decoder-sized GEMMs,
    # the N = 768 dgrads on 144 of 256 CUs, LayerNorm) leave most of the chip idle.  Hazards are
    # tracked per dy buffer: the side stream starts after an event recorded when dy is complete, and
    # any main-stream kernel that overwrites a buffer a queued weight gradient still reads waits for
    # that reader first (_guard_write).  x operands are saved forward activations, never rewritten
    # during backward.
    _side = None
    _main = None  # the stream a step runs on, pinned for its duration (begin_step / end_step)

    def begin_step(self):
        """Pin the launch stream for one forward+backward: looking torch's current stream up costs
        ~1 us per launch (2800 lookups per step)."""
        if self.device.type == "cuda":
            self._main = torch.cuda.current_stream(self.device)
            self._prev_override = ops.set_stream(self._main.cuda_stream)

    def end_step(self):
        if self._main is not None:
            ops.set_stream(self._prev_override)
            self._main = None

    # Number of weight-gradient streams ("lanes"; TMI_WGRAD_LANES).  Round 3 tried two - a layer's weight gradients are
    # independent, so two side by side could each run with a smaller split-K (fewer workgroups, longer K loops, less slab
    # traffic) - and measured it level with one (8.58 = 8.58 ms/step at a split cap of 4; smaller caps lose with either:
    # the no-split kernel needs 1.36 us per K-tile, the per-CU staging rate, not the MFMA's 0.86).  One stays the default.
    N_LANES = max(1, int(os.environ.get("TMI_WGRAD_LANES", "1")))

    def enable_wgrad_stream(self, on=True):
        on = on and self.device.type == "cuda"
        self._sides = [torch.cuda.Stream(device=self.device) for _ in range(self.N_LANES)] if on else []
        self._side = self._sides[0] if on else None           # lane 0: also the stream of the non-GEMM side work
        self._side_handles = [st.cuda_stream for st in self._sides]
        self._side_handle = self._side_handles[0] if on else None
        self._side_reads = {}     # buffer address -> {lane: event recorded after the lane's last reader of it}
        self._done_events = {}
        # events are reused round-robin: creating two per launch costs more host time than the
        # decoder-sized kernels take (re-recording an event other work already waited on is legal)
        self._ev_ring = [torch.cuda.Event() for _ in range(128)] if on else []
        self._ev_i = 0
        return self

    def _event(self):
        ev = self._ev_ring[self._ev_i]
        self._ev_i = (self._ev_i + 1) % len(self._ev_ring)
        return ev

    def _run_on_side(self, fn, dy, defer=False, ready=None, lane=0):
        """Launch ``fn``'s kernels (readers of the finished buffer ``dy``, writers of gradients only)
        on a weight-gradient stream (``lane``), or inline if there is none.  ``defer`` (TMI_DEFER_WGRAD=1): do not enqueue
        yet - ``_flush_deferred`` does, at the point the caller picks (before a kernel the work should run beside)."""
        if self._side is None:
            fn()
            return
        lane = lane % len(self._sides)
        if ready is None:
            # a ring slot is only good for an event that is waited for at once; one that is parked (TMI_DEFER_WGRAD)
            # could be re-recorded by a later launch before its waiter is enqueued, so it gets an event of its own
            ready = torch.cuda.Event() if (defer and _DEFER_WGRAD) else self._event()
            ready.record(self._main or torch.cuda.current_stream())  # dy is complete on the main stream here
        if defer and _DEFER_WGRAD:
            self.__dict__.setdefault("_deferred", []).append((fn, dy, ready, lane))
            return
        side = self._sides[lane]
        side.wait_event(ready)
        prev = ops.set_stream(self._side_handles[lane])
        try:
            fn()
        finally:
            ops.set_stream(prev)
        # "done" events are STORED (``_side_reads``) and waited for arbitrarily later - decoder buffers two layers on,
        # the early-decoder / kv_rest / embedding hand-offs - so they never come from the ring: one event per (buffer
        # address, lane), re-recorded only by a later reader of the same buffer on the same stream, which supersedes it
        key = dy.data_ptr()
        done = self._done_events.get((key, lane))
        if done is None:
            done = self._done_events[(key, lane)] = torch.cuda.Event()
        done.record(side)
        self._side_reads.setdefault(key, {})[lane] = done

    def _flush_deferred(self):
        pend = self.__dict__.get("_deferred")
        if pend:
            self._deferred = []
            for fn, dy, ready, lane in pend:
                self._run_on_side(fn, dy, ready=ready, lane=lane)

    def _pop_side_reads(self, tensor):
        """Events after which every queued side-stream reader of ``tensor`` has finished (and forget them)."""
        return list(self._side_reads.pop(tensor.data_ptr(), {}).values())

    def _wait_events(self, events):
        main = self._main or torch.cuda.current_stream()
        for ev in events:
            main.wait_event(ev)

    def _guard_write(self, *tensors):
        if self._side is None:
            return
        pend = self.__dict__.get("_deferred")
        if pend and any(t.data_ptr() == dy.data_ptr() for t in tensors for _, dy, _, _ in pend):
            self._flush_deferred()  # a queued-but-not-enqueued reader of this buffer: enqueue it, then wait for it below
        if not self._side_reads:
            return
        for t in tensors:
            self._wait_events(self._pop_side_reads(t))

    # -- the optimizer's LATE slice (optim.Adam.apply_gradients(late=...)): the part of the update whose parameters the next
    # step reads late in its forward runs on the second stream while that step has already started; the step's first
    # reader on the main stream waits for it (``_wait_late``), everything the step puts on the second stream is ordered
    # behind it by the stream itself.
    _late_ev = None      # truthy while some late slice may still be running (tests look at it)

    def run_late(self, fn, key="late"):
        if self._side is None:
            fn()
            return
        main = self._main or torch.cuda.current_stream(self.device)
        ev = self._event()
        ev.record(main)                      # every gradient is final on the main stream here (backward joined the streams)
        self._side.wait_event(ev)
        prev = ops.set_stream(self._side_handle)
        try:
            fn()
        finally:
            ops.set_stream(prev)
        evs = self.__dict__.setdefault("_late_done", {})
        if key not in evs:
            evs[key] = torch.cuda.Event()    # its own event, never a ring slot: it is waited for a whole encoder later
        evs[key].record(self._side)
        pend = self.__dict__.setdefault("_late_pending", {})
        pend[key] = evs[key]
        self._late_ev = pend

    def _wait_late(self, key=None):
        """Order the main stream behind the pending late slice ``key`` (behind all of them without a key); no-op when
        nothing is pending.  Called before the first main-stream access to what a slice updates; ``finish_late`` is the
        same thing for callers outside a step."""
        pend = self.__dict__.get("_late_pending")
        if not pend:
            return
        main = self._main or torch.cuda.current_stream(self.device)
        for k in ([key] if key is not None else list(pend)):
            ev = pend.pop(k, None)
            if ev is not None:
                main.wait_event(ev)
        if not pend:
            self._late_ev = None

    def finish_late(self):
        self._wait_late()

    def gradient_streams(self):
        """Streams other than the compute stream on which gradient-producing kernels are queued."""
        self._flush_deferred()
        return list(self._sides) if self._side is not None else []

    def _join_side(self):
        """Main stream waits for everything queued on the weight-gradient streams."""
        self._flush_deferred()
        if self._side is not None:
            cur = torch.cuda.current_stream()
            for st in self._sides:
                cur.wait_stream(st)
            self._side_reads.clear()

    def refresh_shadows(self):
        """Re-derive the bf16 mirror from the fp32 master (after loading weights; the optimizer
        step keeps it current by itself)."""
        if self.precision != "bf16":
            return
        n = self.arena.numel
        ops.cast_bf16(self.arena.p, n, self.mirror, n, 1, n)

    def W(self, name) -> Tuple[torch.Tensor, int]:
        """(2-D weight tensor [in, out], leading dimension) in the compute dtype."""
        buf = self.mirror if self.precision == "bf16" else self.arena.p
        cache = self.__dict__.setdefault("_w2d", {})
        key = (buf.data_ptr(), name)
        hit = cache.get(key)
        if hit is None:
            shape = self.arena.shapes[name]
            rows = int(np.prod(shape[:-1]))
            hit = cache[key] = (self.arena.view(buf, name).view(rows, shape[-1]), shape[-1])
        return hit

    def _gemm_xw(self, A, wname, Cm, M, N, K, a_sm, *, n_off=0, **kw):
        """Cm = A · W[:, n_off:n_off+N] with W the natural [K, N_total] kernel (forward)."""
        w, ldw = self.W(wname)
        ops.gemm(A, w, Cm, M, N, K, a_sm, 1, ldw, 1, b_off=n_off, **kw)

    def _buf(self, name, shape, dtype=None, zero=False):
        t = self.ws.get(name)
        dtype = dtype or self.dtype
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            # diagnostics, read when a buffer is created: TMI_WS_GUARD=<elements> puts a guard zone behind every workspace
            # buffer (check_workspace_guards), TMI_WS_POISON=1|<names> fills the torch.empty ones with NaN
            guard, poison = int(os.environ.get("TMI_WS_GUARD", "0")), os.environ.get("TMI_WS_POISON", "")
            if guard:
                n = int(np.prod(shape))
                flat = (torch.zeros if zero else torch.empty)(n + guard, dtype=dtype, device=self.device)
                flat[n:] = 77
                self.__dict__.setdefault("_guards", {})[(id(self.ws), name)] = (name, flat[n:])
                t = flat[:n].view(shape)
            else:
                t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            if poison and not zero and t.is_floating_point() and (poison == "1" or name in poison.split(",")):
                t.fill_(float("nan"))  # a workspace buffer read before it is written shows up as NaN
            self.ws[name] = t
        return t

    def _buf_layers(self, prefix, L, name, shape, dtype=None):
        """Per-layer buffers ``{prefix}{i}.{name}`` as slices of ONE [L, *shape] allocation: a constant layer stride, so a
        batched GEMM (tmi_gemm's nbatch) can walk the same operand of every layer in one launch (_wgrad_batched)."""
        st = self._buf(f"{prefix}*.{name}", (max(L, 1),) + tuple(shape), dtype)
        for i in range(L):
            self.ws[f"{prefix}{i}.{name}"] = st[i]
        return st

    def _layer_stride(self, name_fmt, L):
        """Arena distance between the same variable of consecutive layers (the arena lays every layer out alike)."""
        offs = [self.arena.offsets[name_fmt.format(i)] for i in range(L)]
        stride = offs[1] - offs[0] if L > 1 else 0
        if any(offs[i + 1] - offs[i] != stride for i in range(L - 1)):
            raise RuntimeError(f"{name_fmt}: layers are not at a constant arena stride")
        return stride

    def _wgrad_batched(self, x_stack, dy_stack, wname_fmt, L, bias=True, lo=0, hi=None):
        """dW_l = x_lᵀ · dy_l (and db_l = colsum(dy_l)) for every layer l in ONE launch each: the deferred form of
        ``_dense_bwd``'s weight gradient.  At M = B * 100 rows a layer's weight gradient is a 13-26 us launch that cannot
        fill the chip (Wav2Vec2: 58 of them per step, the Whisper decoder 24); L of them side by side are one GEMM with L
        times the tiles.  x_stack [L, M, K_in], dy_stack [L, M, N] (``_buf_layers``); the gradient arena supplies the
        constant stride on the output side.  ``lo`` / ``hi``: only layers [lo, hi) - a chunk whose dy are already final can
        start while the backward chain is still working on the layers below it."""
        a = self.arena
        hi = L if hi is None else hi
        if hi <= lo:
            return
        w0 = wname_fmt.format(lo)
        K_in, N = a.shapes[w0][-2], a.shapes[w0][-1]
        M = x_stack.shape[1]
        stride = self._layer_stride(wname_fmt, L)
        xs, dys = x_stack[lo:hi], dy_stack[lo:hi]
        ops.gemm(xs, dys, a.grad(w0).view(K_in, N), K_in, N, M, 1, x_stack.stride(1), dy_stack.stride(1), 1, N,
                 nbatch=hi - lo, a_sb=x_stack.stride(0), b_sb=dy_stack.stride(0), c_sb=stride, splitk=0)
        b0 = w0.replace(".kernel", ".bias")
        if bias and b0 in a.offsets:
            ops.bias_grad_batched(dys, a.grad(b0), self._layer_stride(wname_fmt.replace(".kernel", ".bias"), L))

    def check_workspace_guards(self):
        """Names of workspace buffers whose guard zone (TMI_WS_GUARD=<elements>) was written: a kernel ran past their end."""
        torch.cuda.synchronize()
        return [name for name, tail in self.__dict__.get("_guards", {}).values() if not bool((tail == 77).all())]

    def _dense_fwd(self, x2d, wname, out2d, n_off=0, n_cols=None, **epi):
        """out = x @ W[:, n_off:n_off+n_cols] (+bias ...).  W is the Keras [in, out] kernel."""
        w, ldw = self.W(wname)
        K = w.shape[0]
        N = n_cols if n_cols is not None else self.arena.shapes[wname][-1]
        bias = None
        bname = wname.replace(".kernel", ".bias")
        if bname in self.arena.offsets:
            bias = self.arena.param(bname)[n_off:n_off + N]
        self._gemm_xw(x2d, wname, out2d, x2d.shape[0], N, K, x2d.stride(0), ldc=out2d.stride(0), n_off=n_off,
                      bias=bias, **epi)

    def _dense_bwd(self, x2d, dy2d, wname, dx2d=None, accumulate_dx=False, aux_in=None, dgrad_on_side=False,
                   dgrad_epi=None, bias_done=False, defer=False, wgrad=True, lane=0):
        """dW = xᵀ·dy, db = colsum(dy), optionally dx (=|+=) dy·Wᵀ (* gelu'(aux_in)).
        ``dgrad_on_side``: dx is not needed by the chain that follows (the caller joins the side stream
        before its consumer), so the dgrad goes to the weight-gradient stream too."""
        w, ldw = self.W(wname)
        K_in = w.shape[0]
        N = self.arena.shapes[wname][-1]
        M = x2d.shape[0]
        dW = self.arena.grad(wname).view(K_in, N)
        bname = wname.replace(".kernel", ".bias")

        def weight_grads():
            ops.gemm(x2d, dy2d, dW, K_in, N, M, 1, x2d.stride(0), dy2d.stride(0), 1, N,
                     splitk=0)
            if bname in self.arena.offsets and not bias_done:  # (bias_done: the kernel that produced dy emitted its column sums)
                ops.bias_grad(dy2d, self.arena.grad(bname))

        def dgrad():
            ops.gemm(dy2d, w, dx2d, M, K_in, N, dy2d.stride(0), 1, 1, ldw, dx2d.stride(0),
                     accumulate=accumulate_dx, aux_in=aux_in, **(dgrad_epi or {}))

        if dgrad_on_side and dx2d is not None and self._side is not None:
            self._guard_write(dx2d)
            self._run_on_side(lambda: (weight_grads(), dgrad()), dy2d)
            return
        if wgrad:  # (False: the caller batches this layer's weight gradient with the other layers', _wgrad_batched)
            self._run_on_side(weight_grads, dy2d, defer=defer, lane=lane)
        if dx2d is not None:
            self._guard_write(dx2d)
            dgrad()

    def _ln_fwd(self, x2d, pname, y2d, stat, drop_site=None):
        """``drop_site``: the LayerNorm is followed by Dropout (V:296, V:560, V:779): one pass writes Dropout(LayerNorm(x))."""
        a = self.arena
        if drop_site is not None and self._drop_p > 0.0:
            ops.layernorm_dropout_fwd(x2d, a.param(pname + ".gamma"), a.param(pname + ".beta"), y2d, self.ws[stat + ".mean"],
                                      self.ws[stat + ".rstd"], self.layer_norm_eps, self._drop_p, self._site_seed(drop_site))
            return
        ops.layernorm_fwd(x2d, a.param(pname + ".gamma"), a.param(pname + ".beta"), y2d,
                          self.ws[stat + ".mean"], self.ws[stat + ".rstd"], self.layer_norm_eps)

    def _ln_bwd(self, dy2d, x2d, pname, dx2d, stat, accumulate, emit=None, drop_site=None):
        """``emit`` = (bias gradient tensor, masked-copy buffer or None, dropout site or None): the Dense layer below this
        LayerNorm takes dx (or its Dropout-masked copy) as dy; its bias gradient and the masked copy come out of this
        kernel (tmi_layernorm_bwd_emit) instead of a dropout pass and a column-sum pass over dx.  A buffer without
        dropout (rate 0 or no site) receives a plain copy of dx: the snapshot a deferred weight gradient reads."""
        a = self.arena
        self._guard_write(dx2d)
        if emit is None:
            if drop_site is not None and self._drop_p > 0.0:  # (the forward was _ln_fwd(..., drop_site): dy is masked on load)
                ops.layernorm_dropout_bwd(dy2d, x2d, a.param(pname + ".gamma"), self.ws[stat + ".mean"], self.ws[stat + ".rstd"],
                                          dx2d, a.grad(pname + ".gamma"), a.grad(pname + ".beta"), self._drop_p,
                                          self._site_seed(drop_site), accumulate_dx=accumulate)
                return
            ops.layernorm_bwd(dy2d, x2d, a.param(pname + ".gamma"), self.ws[stat + ".mean"], self.ws[stat + ".rstd"],
                              dx2d, a.grad(pname + ".gamma"), a.grad(pname + ".beta"), accumulate_dx=accumulate)
            return
        if drop_site is not None:
            raise ValueError("_ln_bwd: the emitting form has no mask on dy")
        colsum, masked, site = emit
        p = self._drop_p if (masked is not None and site is not None) else 0.0
        if masked is not None:
            self._guard_write(masked)
        ops.layernorm_bwd_emit(dy2d, x2d, a.param(pname + ".gamma"), self.ws[stat + ".mean"], self.ws[stat + ".rstd"], dx2d,
                               a.grad(pname + ".gamma"), a.grad(pname + ".beta"), colsum, masked=masked,
                               dropout_p=p, dropout_seed=self._site_seed(site) if p > 0 else 0, accumulate_dx=accumulate)

    # ---- dropout (training mode of the reference, W:160 / W:205 / W:342 / W:411): counter-based masks
    # regenerated in backward (tmi_dropout, tmi_attn_*).  Off unless ``enable_dropout`` was called: with
    # dropout on a step has no parity definition against the reference (TF's RNG stream), only against
    # the oracle fed the same masks.
    _drop_p = 0.0          # hidden-state dropout rate (config.dropout)
    _drop_attn_p = 0.0     # attention-probability dropout rate (config.attention_dropout)
    _drop_act_p = 0.0      # FFN-intermediate rate (Wav2Vec2 activation_dropout, V:393; Whisper's is 0.0, W:31)
    _drop_base = 0
    _drop_step = 0

    def enable_dropout(self, p, attn_p, seed=0x5EED, act_p=0.0):
        if (p > 0 or attn_p > 0 or act_p > 0) and self.precision != "bf16":
            raise ValueError("dropout is implemented for the bf16 (fused attention) path; fp32 is the parity mode (rates 0)")
        self._drop_p, self._drop_attn_p, self._drop_act_p = float(p), float(attn_p), float(act_p)
        self._drop_base, self._drop_step = int(seed), 0
        self._ws_key = None  # the workspace layout depends on the mode: lay it out again on the next step

    def _site_seed(self, site: int) -> int:
        """Seed of dropout site ``site`` in the current step (restated in oracle/dropout.py)."""
        return (self._drop_base + self._drop_step * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _dropout(self, x2d, out2d, site, resid=None, p=None):
        """out = (resid or 0) + Dropout_p(x) at dropout site ``site`` (p defaults to the hidden-state rate)."""
        self._guard_write(out2d)
        ops.dropout(x2d, out2d, x2d.shape[0], x2d.shape[1], self._drop_p if p is None else p, self._site_seed(site),
                    resid=resid)

    def _drop_epi(self, site, p=None):
        """GEMM-epilogue form of ``_dropout`` (same generator, same counters): keyword arguments for ``ops.gemm``."""
        p = self._drop_p if p is None else p
        return {"dropout_p": p, "dropout_seed": self._site_seed(site)} if p > 0.0 else {}

    def _attn_fwd(self, key, q, k, v, ctx2d, B, H, Tq, Tk, mask, score_scale=1.0, site=None):
        """score_scale multiplies q·kᵀ (V:349); Whisper pre-scales q instead (W:141) and passes 1."""
        d = self.hidden
        (qt, qo), (kt, ko), (vt, vo) = q, k, v
        if self.precision == "bf16":
            dp = self._drop_attn_p if site is not None else 0.0
            ops.attn_fwd((qt, qo, Tq * qt.stride(0), qt.stride(0)), (kt, ko, Tk * kt.stride(0), kt.stride(0)),
                         (vt, vo, Tk * vt.stride(0), vt.stride(0)), (ctx2d, 0, Tq * d, d),
                         self.ws[key], B, H, Tq, Tk, mask, score_scale=score_scale,
                         dropout_p=dp, dropout_seed=self._site_seed(site) if dp > 0 else 0,
                         drop_mask=self._attn_dropmask(key, B, H, Tq, Tk) if dp > 0 else None)
            return
        # fp32 parity mode (W:147-167 as written: scores materialised).  One launch per product over all (sample, head)
        # pairs: heads are the inner batch level (stride hd inside a token row), samples the outer one (tmi_gemm nbatch2)
        P = self.ws[key]
        hd = d // H
        ops.gemm(qt, kt, P, Tq, Tk, hd, qt.stride(0), 1, 1, kt.stride(0), Tk, nbatch=H, a_sb=hd, b_sb=hd,
                 c_sb=Tq * Tk, a_off=qo, b_off=ko, scale_cols=Tk if score_scale != 1.0 else 0, scale=score_scale,
                 nbatch2=B, a_sb2=Tq * qt.stride(0), b_sb2=Tk * kt.stride(0), c_sb2=H * Tq * Tk)
        ops.softmax_fwd(P, B * H * Tq, Tq, Tk, mask)
        ops.gemm(P, vt, ctx2d, Tq, hd, Tk, Tk, 1, vt.stride(0), 1, d, nbatch=H, a_sb=Tq * Tk, b_sb=hd, c_sb=hd,
                 b_off=vo, nbatch2=B, a_sb2=H * Tq * Tk, b_sb2=Tk * vt.stride(0), c_sb2=Tq * d)

    def _attn_dropmask(self, key, B, H, Tq, Tk):
        """The stored keep bits of attention call ``key`` (W:160: TF keeps the mask it drew for the gradient): one buffer per
        call site, alive from its forward to its backward like the softmax statistics next to it."""
        return self._buf(key + ".dropmask", (int(ops.lib().tmi_attn_dropmask_bytes(B, H, Tq, Tk)),), torch.uint8)

    def _attn_bwd(self, key, q, k, v, ctx2d, dctx2d, dq, dk, dv, B, H, Tq, Tk, mask, score_scale=1.0,
                  q_prescaled=True, site=None, dkv_on_side=False):
        """``dkv_on_side``: the dK/dV pass goes to the weight-gradient stream (cross-attention: nothing on the decoder's
        backward chain reads dK / dV; the caller joins that stream before it does).  ``dctx2d`` must then be a buffer nothing
        on the main stream rewrites soon (a ``_guard_write`` on it waits for the pass)."""
        d = self.hidden
        hd = d // H
        scaling = hd ** -0.5
        (qt, qo), (kt, ko), (vt, vo) = q, k, v
        (dqt, dqo), (dkt, dko), (dvt, dvo) = dq, dk, dv
        self._guard_write(dqt, dkt, dvt)
        if self.precision == "bf16":
            def m(t, off, T):
                return (t, off, T * t.stride(0), t.stride(0))
            split = dkv_on_side and self._side is not None and _DKV_ON_SIDE
            # (a pass on another stream reads delta later: its own buffer, not the one every attention backward shares)
            delta = self._buf(key + ".delta", (B, H, Tq), torch.float32) if split else self.ws["delta"]

            def run(passes):
                ops.attn_bwd(m(qt, qo, Tq), m(kt, ko, Tk), m(vt, vo, Tk), m(ctx2d, 0, Tq), self.ws[key],
                             m(dctx2d, 0, Tq), m(dqt, dqo, Tq), m(dkt, dko, Tk), m(dvt, dvo, Tk), delta,
                             B, H, Tq, Tk, mask, dq_scale=scaling if q_prescaled else 1.0, score_scale=score_scale,
                             dropout_p=self._drop_attn_p if site is not None else 0.0,
                             dropout_seed=self._site_seed(site) if (site is not None and self._drop_attn_p > 0) else 0,
                             passes=passes,
                             drop_mask=self._attn_dropmask(key, B, H, Tq, Tk) if (site is not None and self._drop_attn_p > 0) else None)
            if split:
                run(1)
                self._run_on_side(lambda: run(2), dctx2d)
            else:
                run(0)
            return
        P = self.ws[key]
        dP = self.ws["dP"]
        PP = H * Tq * Tk  # one sample's scores
        # dP = dctx · Vᵀ
        ops.gemm(dctx2d, vt, dP, Tq, Tk, hd, d, 1, 1, vt.stride(0), Tk, nbatch=H, a_sb=hd, b_sb=hd, c_sb=Tq * Tk,
                 b_off=vo, nbatch2=B, a_sb2=Tq * d, b_sb2=Tk * vt.stride(0), c_sb2=PP)
        # dV = Pᵀ · dctx
        ops.gemm(P, dctx2d, dvt, Tk, hd, Tq, 1, Tk, d, 1, dvt.stride(0), nbatch=H, a_sb=Tq * Tk, b_sb=hd, c_sb=hd,
                 c_off=dvo, nbatch2=B, a_sb2=PP, b_sb2=Tq * d, c_sb2=Tk * dvt.stride(0))
        ops.softmax_bwd(P, dP, B * H * Tq, Tk)
        # dQ = dS · K  (then * scaling: chain rule of W:141, folded into this GEMM's column scale)
        ops.gemm(dP, kt, dqt, Tq, hd, Tk, Tk, 1, kt.stride(0), 1, dqt.stride(0), nbatch=H, a_sb=Tq * Tk, b_sb=hd,
                 c_sb=hd, b_off=ko, c_off=dqo, scale_cols=hd, scale=scaling,
                 nbatch2=B, a_sb2=PP, b_sb2=Tk * kt.stride(0), c_sb2=Tq * dqt.stride(0))
        # dK = dSᵀ · Q
        # (a pre-scaled q already carries the factor; otherwise the score scale applies here too)
        ops.gemm(dP, qt, dkt, Tk, hd, Tq, 1, Tk, qt.stride(0), 1, dkt.stride(0), nbatch=H, a_sb=Tq * Tk, b_sb=hd,
                 c_sb=hd, b_off=qo, c_off=dko, scale_cols=0 if q_prescaled else hd, scale=score_scale,
                 nbatch2=B, a_sb2=PP, b_sb2=Tq * qt.stride(0), c_sb2=Tk * dkt.stride(0))
