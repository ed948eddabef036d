"""tf.keras.optimizers.Adam over the flat parameter arena (W:901; V:1271-1275).

One kernel launch for the whole model (``tmi_adam_step``): Keras-V2 epsilon placement by
default (``eps_mode="tf"``), AdamW-style decoupled weight decay available but 0 as in the
reference.  ``apply_gradients`` is the analogue of ``optimizer.apply_gradients`` (W:834):
it first lets the strategy all-reduce the gradient arena (the implicit cross-replica SUM
of Keras OptimizerV2), then updates; the same kernel writes the model's bf16 weight mirror.
"""
from __future__ import annotations

import os

import torch

from . import ops


class Adam:
    def __init__(self, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, eps_mode="tf",
                 weight_decay=0.0):
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        if eps_mode not in ("tf", "torch"):
            raise ValueError("eps_mode must be 'tf' or 'torch'")
        self.eps_mode = 0 if eps_mode == "tf" else 1
        self.weight_decay = float(weight_decay)
        self.iterations = 0
        self.row_sparse = True  # embedding tables through tmi_adam_step_rows (idle rows skipped; same result bit for bit)

    def apply_gradients(self, model, strategy=None, grad_scale=1.0, zero_grad=False, late=None):
        """``zero_grad``: the kernel leaves the gradient arena zeroed (``arena.g_clean``), so the next
        ``forward_backward`` skips its fill pass; callers that still want to look at the gradients pass False.
        After ``begin_early`` only the ranges no early slice has updated are left to do (same step count).
        ``late`` = (lo, hi) or [(lo, hi, key), ...]: those arena ranges are updated on the model's second stream
        (``model.run_late``) and this call returns without waiting for them - the caller promises that the next thing to touch those parameters is the next
        step's forward (which waits, ``KernelBlocks._wait_late``) or comes after ``model.finish_late()``.  Whisper: the
        decoder layers, 26 % of the small-ref parameters, whose first reader is a whole encoder forward away."""
        a = model.arena
        if strategy is not None:
            strategy.all_reduce_gradients(a.g)
        early, self._early = self.__dict__.get("_early"), None
        if early is None:
            self.iterations += 1
            todo = [(0, a.numel)]
        else:
            if not zero_grad:
                raise ValueError("early Adam slices zero their gradients: finish the step with zero_grad=True")
            todo, pos = [], 0
            for lo, hi in sorted(early):
                if lo > pos:
                    todo.append((pos, lo))
                pos = max(pos, hi)
            if pos < a.numel:
                todo.append((pos, a.numel))
        if late and not zero_grad:
            raise ValueError("a late Adam slice zeroes its gradients behind the caller's back: zero_grad=True")
        # ``late``: (lo, hi) or a list of (lo, hi, key) in the order the next step will need them
        if late and not isinstance(late[0], (tuple, list)):
            late = [(late[0], late[1], "late")]
        pieces = [(lo, hi, None) for lo, hi in todo]
        for l_lo, l_hi, key in (late or []):
            cut = []
            for lo, hi, k in pieces:
                if k is not None or hi <= l_lo or lo >= l_hi:
                    cut.append((lo, hi, k))
                    continue
                if lo < l_lo:
                    cut.append((lo, l_lo, None))
                cut.append((max(lo, l_lo), min(hi, l_hi), key))
                if l_hi < hi:
                    cut.append((l_hi, hi, None))
            pieces = cut
        for lo, hi, k in pieces:          # what the next step reads first: now, on this stream
            if k is None:
                self._update(model, lo, hi, grad_scale, zero_grad, 0)
        for _, _, key in (late or []):    # the rest on the model's second stream, in the order given
            mine = [(lo, hi) for lo, hi, k in pieces if k == key]
            if mine:
                model.run_late(lambda mine=mine: [self._update(model, lo, hi, grad_scale, zero_grad, self.LATE_BLOCKS)
                                                  for lo, hi in mine], key)
        a.g_clean = bool(zero_grad)

    # grid of the late slice: it runs beside the next step's first kernels (0 = full width)
    LATE_BLOCKS = int(os.environ.get("TMI_ADAM_LATE_BLOCKS", "128"))

    def begin_early(self, model, grad_scale=1.0):
        """Start this step's update before backward has finished: returns ``update(lo, hi)``, which runs Adam on the arena
        range [lo, hi) NOW, on whatever stream ``ops`` is pinned to (the caller orders it after the last writer of those
        gradients and the last reader of those weights); ``apply_gradients`` then does the rest.  Whisper uses it for the
        LM head and the embedding table (54 % of the small-ref parameters): their gradients are final at the very start /
        end of the decoder's backward - a chain of decoder-sized kernels that leaves the chip idle - so their share of the
        4.4 GB Adam stream runs on the second stream under it instead of alone at the end of the step.  Per-parameter
        arithmetic is unchanged (Adam is elementwise): the result is bit for bit the single-launch update."""
        self.iterations += 1
        self._early = []
        if self.row_sparse and self.weight_decay == 0.0 and hasattr(model, "embedding_tables"):
            for off, rows, _ in model.embedding_tables():   # created (and filled) HERE, on the main stream, before any
                self._row_flags(model, off, rows)            # side-stream work of this step is ordered behind it
        blocks = self.EARLY_BLOCKS  # a throttled grid: the full-width kernel saturates HBM and starves the chain it runs under

        def update(lo, hi):
            self._update(model, lo, hi, grad_scale, True, blocks)
            self._early.append((lo, hi))
        return update

    EARLY_BLOCKS = int(os.environ.get("TMI_ADAM_EARLY_BLOCKS", "256"))

    # -- early slices with replicas: the same two slices, released by the bucket that carries them ---------------
    def begin_early_buckets(self, model, strategy, ranges, grad_scale=1.0):
        """``begin_early`` when the gradients are exchanged first (replicas, W:834): the strategy hands every finished bucket to
        ``_early_bucket``; a bucket that carries one of ``ranges`` whole (Whisper: the LM head, the embedding table - the two
        160 MB variables, each the bulk of its bucket) is updated on the optimizer stream as soon as ITS collective is done,
        on the throttled grid, while backward goes on; every other bucket stays with the strategy and is updated by
        ``apply_gradients`` at the end of the step, as without this.  ``early_buckets_ran`` counts the slices of the last step."""
        self.iterations += 1
        self._early = []
        self.early_buckets_ran = 0
        if self.row_sparse and self.weight_decay == 0.0 and hasattr(model, "embedding_tables"):
            for off, rows, _ in model.embedding_tables():
                self._row_flags(model, off, rows)
        if model.device.type == "cuda" and getattr(self, "_os", None) is None:
            self._os = torch.cuda.Stream(device=model.device)
        self._eb = (model, strategy, list(ranges), grad_scale)
        strategy.on_bucket = self._early_bucket

    def _early_bucket(self, lo, hi, works, post):
        model, strategy, ranges, grad_scale = self._eb
        if not any(lo <= r_lo and r_hi <= hi for r_lo, r_hi in ranges):
            strategy._works.extend(works)   # not ours: the strategy waits for it in all_reduce_gradients
            if post is not None:
                strategy._post.append(post)
            return
        if model.device.type != "cuda":     # (host tensors, the gloo tests: no streams, works already complete)
            for w in works:
                w.wait()
            if post is not None:
                post()
            self._update(model, lo, hi, grad_scale, True, 0)
        else:
            self._os.wait_stream(torch.cuda.current_stream(model.device))  # last readers of these weights in this step
            for st in model.gradient_streams():
                self._os.wait_stream(st)
            prev = ops.set_stream(self._os.cuda_stream)
            try:
                with torch.cuda.stream(self._os):
                    for w in works:         # the optimizer stream (not the host) waits for the bucket's collective
                        w.wait()
                    if post is not None:
                        post()
                    self._update(model, lo, hi, grad_scale, True, self.EARLY_BLOCKS)
            finally:
                ops.set_stream(prev)
        self._early.append((lo, hi))
        self.early_buckets_ran += 1

    def finish_early_buckets(self, model, strategy):
        """After ``apply_gradients``: give the bucket hook back and order the compute stream after the optimizer stream."""
        strategy.on_bucket = None
        self._eb = None
        if model.device.type == "cuda":
            torch.cuda.current_stream(model.device).wait_stream(self._os)

    def abort_early(self):
        """The step that called ``begin_early`` raised before ``apply_gradients``: forget its slices and give the step count
        back, so the next step is a whole one (slices that already ran have moved their parameters: the caller's model is
        as undefined as after any half-finished step, but the optimizer's bookkeeping is consistent again)."""
        if self.__dict__.get("_early") is not None:
            self._early = None
            self.iterations -= 1

    def _update(self, model, lo, hi, grad_scale, zero_grad, max_blocks):
        """Adam over arena range [lo, hi).  The parts of it that are embedding tables (``model.embedding_tables()``:
        (offset, rows, row_len)) go through the row-sparse kernel: the decoder's 51865-row table is 27 % of the
        small-ref arena and a batch touches ~100 of its rows."""
        a = model.arena
        mir = model.mirror

        def dense(s0, s1):
            if s1 > s0:
                ops.adam_step(a.p[s0:s1], a.g[s0:s1], a.m[s0:s1], a.v[s0:s1], s1 - s0, self.learning_rate, self.beta_1,
                              self.beta_2, self.epsilon, self.iterations, self.eps_mode, self.weight_decay, grad_scale,
                              mirror=None if mir is None else mir[s0:s1], zero_grad=zero_grad, max_blocks=max_blocks)
        sparse_ok = self.row_sparse and self.weight_decay == 0.0 and hasattr(model, "embedding_tables")
        tables = model.embedding_tables() if sparse_ok else []
        if not sparse_ok and hasattr(model, "embedding_tables") and a.__dict__.get("adam_row_flags"):
            self.invalidate_row_flags(model)   # a dense step moves m / v of rows the flags call idle
        pos = lo
        for off, rows, row_len in tables:
            end = off + rows * row_len
            if off < lo or end > hi:   # (a table cut by a bucket boundary: dense)
                if off < hi and end > lo:
                    self.invalidate_row_flags(model, off)
                continue
            dense(pos, off)
            flags = self._row_flags(model, off, rows)
            ops.adam_step_rows(a.p[off:end], a.g[off:end], a.m[off:end], a.v[off:end], rows, row_len, flags,
                               self.learning_rate, self.beta_1, self.beta_2, self.epsilon, self.iterations, self.eps_mode,
                               grad_scale, mirror=None if mir is None else mir[off:end], zero_grad=zero_grad)
            pos = end
        dense(pos, hi)

    def _row_flags(self, model, off, rows):
        """One byte per table row, "has m / v ever been non-zero".  The flags describe the m / v arenas, so they live
        WITH them (``arena.adam_row_flags``, keyed by the table's offset), not in a table keyed by ``id(model)``:
        whatever overwrites m / v without going through ``tmi_adam_step_rows`` - ``train.load_checkpoint``, a dense
        update of the table (``row_sparse`` False, or a table cut by a bucket boundary) - calls
        ``invalidate_row_flags`` and the next row-sparse step starts from "every row active" (conservative: a row
        with zero m, v and g is left exactly as the dense kernel would leave it, it just is not skipped)."""
        a = model.arena
        store = a.__dict__.setdefault("adam_row_flags", {})
        f = store.get(off)
        if f is None:
            fresh = self.iterations <= 1 and not a.__dict__.get("adam_state_dirty", False)
            f = store[off] = (torch.zeros if fresh else torch.ones)(rows, dtype=torch.uint8, device=model.device)
            # The fill above is enqueued on torch's CURRENT stream; tmi_adam_step_rows reads and writes the flags on the
            # stream ``ops`` is pinned to (the second stream, for an early / late slice).  Nothing else orders the two, so
            # the consumer stream waits for the fill here (a fill landing after the kernel re-marks touched rows
            # idle, and they are skipped for good).
            if model.device.type == "cuda":
                cur = torch.cuda.current_stream(model.device)
                if ops.stream() != cur.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    torch.cuda.ExternalStream(ops.stream(), device=model.device).wait_event(ev)
        return f

    @staticmethod
    def invalidate_row_flags(model, off=None):
        """m / v of the embedding tables (of the one at ``off``) were written by something else than the row-sparse kernel."""
        a = model.arena
        store = a.__dict__.setdefault("adam_row_flags", {})
        if off is None:
            store.clear()
        else:
            store.pop(off, None)
        a.adam_state_dirty = True

    def apply_gradients_clipped(self, model, chunks, sumsq, nseg, clip_global=0.0, clip_each=0.0, grad_scale=1.0,
                                zero_grad=False, late_row=None):
        """Adam with V:1243's global-norm clip and / or V:1274's per-variable clipnorm folded into the kernel as a factor
        of g (``sumsq``: per-variable sums of squares of the raw gradients, ops.segment_sumsq; ``chunks``: ops.segment_chunks of
        the variables' offsets): no clipped copy of the
        gradient arena is written or re-read."""
        a = model.arena
        self.iterations += 1

        def rows(tab, n, blocks):
            ops.adam_step_segments(a.p, a.g, a.m, a.v, n, tab, sumsq, nseg, clip_global, clip_each, self.learning_rate,
                                   self.beta_1, self.beta_2, self.epsilon, self.iterations, self.eps_mode, self.weight_decay,
                                   grad_scale, mirror=model.mirror, zero_grad=zero_grad, max_blocks=blocks)
        if late_row is None:
            rows(chunks, a.numel, 0)
        else:
            # ``late_row`` = (row index, first arena offset of that row): the rows from there on are the LATE slice (see
            # apply_gradients): same launch arguments - the global norm comes from the whole ``sumsq`` - on the model's
            # second stream, not waited for here
            if not zero_grad:
                raise ValueError("a late Adam slice zeroes its gradients behind the caller's back: zero_grad=True")
            k, off = late_row
            rows(chunks[:k], off, 0)
            model.run_late(lambda: rows(chunks[k:], a.numel - off, self.LATE_BLOCKS))
        a.g_clean = bool(zero_grad)

    # -- the update bucket by bucket under backward -------------------------------------------------------------
    # Adam is a 4.4 GB/step HBM stream (small-ref) that needs nothing but final gradients: instead of one launch
    # behind backward it runs slice by slice on its own stream as the strategy releases buckets (after the bucket's
    # all-reduce when there are replicas), on a small grid so that the GEMMs of the remaining backward keep their CUs.
    # A slice's weights are final in this step once its gradients are: every forward and backward read of them has
    # been enqueued before the range is reported (KernelBlocks reports whole layers), so updating them - and their
    # bf16 mirror - under the rest of backward is safe.
    OVERLAP_BLOCKS = int(os.environ.get("TMI_ADAM_OVERLAP_BLOCKS", "48"))

    def begin_overlapped(self, model, strategy, grad_scale=1.0, zero_grad=True):
        self.iterations += 1
        self._ov = (model, grad_scale, zero_grad)
        if model.device.type == "cuda" and getattr(self, "_os", None) is None:
            self._os = torch.cuda.Stream(device=model.device)
        strategy.on_bucket = self._bucket_ready
        strategy.producers = model.gradient_streams
        strategy.pre_launch = None

    def _bucket_ready(self, lo, hi, works, post):
        model, grad_scale, zero_grad = self._ov
        a = model.arena
        main = torch.cuda.current_stream(model.device)
        self._os.wait_stream(main)      # the bucket's producers: the compute stream and the weight-gradient stream
        for st in model.gradient_streams():
            self._os.wait_stream(st)
        prev = ops.set_stream(self._os.cuda_stream)
        try:
            with torch.cuda.stream(self._os):
                for w in works:         # replicas: the optimizer stream (not the host) waits for the bucket's reduce
                    w.wait()
                if post is not None:
                    post()
                self._update(model, lo, hi, grad_scale, zero_grad, self.OVERLAP_BLOCKS)
        finally:
            ops.set_stream(prev)

    def finish_overlapped(self, model, strategy):
        """Release whatever the strategy still holds (the head of the arena), then order the compute stream after the
        optimizer stream: the step's parameters are final for whoever runs next on it."""
        strategy.all_reduce_gradients(model.arena.g)
        strategy.on_bucket = None
        torch.cuda.current_stream(model.device).wait_stream(self._os)
        model.arena.g_clean = bool(self._ov[2])
        self._ov = None

    # -- captured-graph form: the launch reads its step-dependent scalars from device memory
    def scalars(self, step=None):
        return ops.adam_scalars(self.learning_rate, self.beta_1, self.beta_2, self.iterations if step is None else step,
                                self.eps_mode, self.weight_decay)

    def apply_gradients_dev(self, model, dev_scalars, grad_scale=1.0):
        """The update of ``apply_gradients`` with [step_size, vcorr, decay] taken from ``dev_scalars``
        (a 3-float device tensor the caller refreshes from ``scalars()`` before every replay)."""
        a = model.arena
        if hasattr(model, "embedding_tables"):
            self.invalidate_row_flags(model)  # dense kernel: the row-activity flags are not kept
        ops.adam_step_dev(a.p, a.g, a.m, a.v, a.numel, self.beta_1, self.beta_2, self.epsilon, dev_scalars,
                          self.eps_mode, grad_scale, mirror=model.mirror)
