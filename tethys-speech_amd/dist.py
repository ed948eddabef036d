"""Data-parallel strategy: one process per GPU, gradient SUM all-reduce over RCCL/xGMI.

Stands in for ``tf.distribute.MultiWorkerMirroredStrategy`` (W:1047): C1 = the implicit
cross-replica sum of every gradient inside ``optimizer.apply_gradients`` (W:834), C2 =
``strategy.reduce(SUM, loss)`` (W:848), C4 = broadcast of the initial weights from the
chief.  Whisper does not divide by the replica count (W:829-836), so the applied gradient
is the SUM and the printed loss is the SUM of per-replica mean losses.

The gradient arena is one flat fp32 buffer filled from its end towards its start by
backward, so buckets are contiguous slices.  Each bucket is handed to
``torch.distributed.all_reduce(async_op=True)`` (backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests): RCCL runs it on its own HIP stream, ordered after the compute stream's
work so far, and ``wait()`` makes the compute stream wait for it — the Adam launch is the
first consumer.  Rank/world size come from TF_CONFIG (the reference's harness) or from
RANK / WORLD_SIZE (torchrun).
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .plan import WorkSlot, host_call


def task_from_env(env=None) -> Tuple[str, int, int, int]:
    """-> (task_type, task_index, rank, world).  TF_CONFIG wins when it names a cluster
    (W:1037-1040, job_name.py:3-14); chief ranks first, then workers."""
    env = os.environ if env is None else env
    tf_config = json.loads(env.get("TF_CONFIG") or "{}")
    task = tf_config.get("task", {})
    cluster = tf_config.get("cluster", {})
    if cluster:
        n_chief = len(cluster.get("chief", []))
        n_worker = len(cluster.get("worker", []))
        ttype, tidx = task.get("type", "worker"), int(task.get("index", 0))
        rank = tidx if ttype == "chief" else n_chief + tidx
        return ttype, tidx, rank, n_chief + n_worker
    if "RANK" in env and "WORLD_SIZE" in env:
        rank, world = int(env["RANK"]), int(env["WORLD_SIZE"])
        return "worker", rank, rank, world
    return task.get("type") or "worker", int(task.get("index") or 0), 0, 1


def rendezvous_from_env(env=None) -> Tuple[Optional[str], Optional[str]]:
    """(MASTER_ADDR, MASTER_PORT) for torch.distributed's rendezvous.  Explicit MASTER_ADDR / MASTER_PORT win;
    otherwise, when TF_CONFIG names a cluster, rank 0's "host:port" entry (chief[0], or worker[0] without a
    chief) — the address every pod of the reference's TFJob already knows (sample_tfjobs/*.yaml run chief and
    worker in separate pods, so a loopback default would have each pod rendezvous with itself)."""
    env = os.environ if env is None else env
    addr, port = env.get("MASTER_ADDR"), env.get("MASTER_PORT")
    cluster = json.loads(env.get("TF_CONFIG") or "{}").get("cluster", {})
    first = (cluster.get("chief") or cluster.get("worker") or [None])[0]
    if first:
        host, _, p = str(first).rpartition(":")
        if not host:  # no port in the entry
            host, p = str(first), ""
        addr = addr or host
        port = port or (p if p.isdigit() else None)
    return addr, port


class DataParallelStrategy:
    """``exchange``: how a bucket is summed over the replicas -
        "allreduce"  one ``all_reduce`` per bucket (RCCL picks its rings / trees);
        "rs_ag"      ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` over the same bucket (the two halves of
                     a ring all-reduce as separate collectives: the form an optimizer working on 1/N shards needs);
        "mesh"       xGMI is a full point-to-point mesh (7 links per GPU), which a single ring leaves mostly idle:
                     ``all_to_all_single`` sends piece j of the bucket straight to rank j over the direct link (all
                     links at once), a local kernel sums the N received pieces (tmi_grad_unpack), and an all-gather
                     returns the reduced pieces.
    ``grad_dtype``: "fp32" moves the arena slices as they are (the reference's arithmetic); "bf16" packs each
    bucket to bf16 when it is launched (half the bytes on the links) and widens the reduced bucket back to fp32
    before the optimizer reads it; sums across ranks are then bf16 sums (RCCL) - or, with "mesh", fp32 sums of
    bf16 pieces."""

    def __init__(self, rank: int = 0, world: int = 1, backend: Optional[str] = None,
                 bucket_bytes: int = 24 << 20, init: bool = True, grad_dtype: Optional[str] = None,
                 exchange: Optional[str] = None, force_collectives: bool = False):
        self.rank, self.world = rank, world
        # ``force_collectives``: issue every collective even with ONE replica (the world-1 short-circuits below are
        # skipped).  A sum over one rank is the identity, so a step must come out bit for bit as without it - which
        # lets a one-GPU box run the real RCCL path (nccl backend, Work objects, exchange-stream ordering against both
        # producer streams, staging buffers) end to end: tests/test_rccl_world1_gpu.py.
        self.force_collectives = bool(force_collectives)
        # ``exchange_off``: run the step with the gradient exchange left out (bench.py's N > 1 line times the same ranks
        # both ways: the difference is the exposed cost of the exchange)
        self.exchange_off = False
        # launch threshold of the overlapped exchange.  24 MiB: every Whisper small-ref layer (28-38 MB of
        # fp32 gradients) goes out as soon as it is final, so what remains exposed after backward is
        # the conv stem's 8 MB, not "last layer + stem" (the big tensors, lm_head / embeddings at 160 MB,
        # are single collectives either way; xGMI rings are per-link bound, large messages are fine)
        self.bucket_bytes = bucket_bytes
        self.grad_dtype = grad_dtype or os.environ.get("TMI_GRAD_DTYPE", "fp32")
        self.exchange = exchange or os.environ.get("TMI_EXCHANGE", "allreduce")
        if self.grad_dtype not in ("fp32", "bf16") or self.exchange not in ("allreduce", "rs_ag", "mesh"):
            raise ValueError("grad_dtype must be fp32 | bf16 and exchange allreduce | rs_ag | mesh")
        self._g = None
        self._works: List = []
        self._post: List = []          # per launched bucket: callable run after its works completed (widen / copy back)
        self._stage = {}               # (lo, hi, kind) -> staging tensor, allocated once
        self._pend_lo = self._pend_hi = 0
        self._serial = False           # gloo runs queued works on a thread pool: dependent collectives must be waited for
        if (world > 1 or self.force_collectives) and init and not dist.is_initialized():
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            addr, port = rendezvous_from_env()
            os.environ["MASTER_ADDR"] = addr or "127.0.0.1"
            os.environ["MASTER_PORT"] = port or "29531"
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if (world > 1 or self.force_collectives) and dist.is_initialized():
            self._serial = dist.get_backend() != "nccl"

    @property
    def _collective(self) -> bool:
        """Does this strategy talk to other replicas (or pretend to, ``force_collectives``) in the current step?"""
        return (self.world > 1 or self.force_collectives) and not self.exchange_off

    @property
    def num_replicas_in_sync(self) -> int:
        return self.world

    def buckets(self, n: int) -> List[Tuple[int, int]]:
        """[start, end) element ranges of the flat arena, LAST bucket first (the order in
        which backward completes them)."""
        per = max(1, self.bucket_bytes // 4)
        out = []
        end = n
        while end > 0:
            start = max(0, end - per)
            out.append((start, end))
            end = start
        return out

    # -- C1, overlapped with backward ---------------------------------------------------
    def begin_gradients(self, g: torch.Tensor):
        """Start a step's gradient exchange.  The model then reports, through
        ``gradients_ready(lo, hi)``, each arena range whose gradients are final (every kernel
        that writes it has been enqueued); ranges arrive contiguous and descending, because the
        arena is in forward order and backward fills it from the end."""
        self._g = g
        self._works = []
        self._post = []
        self._pend_lo = self._pend_hi = g.numel()

    on_bucket = None   # optional consumer of finished buckets: on_bucket(lo, hi, works, post) takes over waiting for
                       # the bucket's collectives and running its post-step (optim.Adam updates the slice under backward)

    def gradients_ready(self, lo: int, hi: int):
        if self._g is None or (not self._collective and self.on_bucket is None):
            return
        if hi != self._pend_lo:
            raise RuntimeError(f"gradient ranges must be contiguous and descending: got [{lo},{hi}) after {self._pend_lo}")
        self._pend_lo = lo
        if (self._pend_hi - self._pend_lo) * 4 >= self.bucket_bytes:
            self._launch()

    pre_launch = None  # optional hook run on the compute stream before a bucket is released
    producers = None   # optional callable -> extra streams that also write gradients (the model's weight-gradient
                       # stream): the exchange / optimizer streams wait for them directly, the compute stream does not

    def _buf(self, key, n, dtype, like):
        t = self._stage.get(key)
        if t is None or t.numel() != n or t.dtype != dtype or t.device != like.device:
            t = self._stage[key] = torch.empty(n, dtype=dtype, device=like.device)
        return t

    # fills and copies of the staging buffers: library launches on device tensors (a launch plan records them; torch's
    # own fill_ / copy_ would run while recording and be missing from every replay)
    @staticmethod
    def _fill_zero(t):
        if t.is_cuda:
            from . import ops
            ops.fill_zero(t)
        else:
            t.zero_()

    @staticmethod
    def _copy(dst, src):
        if dst.is_cuda:
            from . import ops
            ops.copy(dst, src)
        else:
            dst.copy_(src)

    def _pack(self, src, dst):
        if src.is_cuda:
            from . import ops
            ops.grad_pack(src, dst, src.numel())
        else:  # (the gloo tests run the strategy on host tensors)
            dst.copy_(src)

    def _unpack(self, src, dst, nparts=1, part_stride=0):
        n = dst.numel()
        if dst.is_cuda:
            from . import ops
            ops.grad_unpack(src, dst, n, nparts, part_stride)
        elif nparts == 1:
            dst.copy_(src[:n])
        else:
            dst.copy_(src[:nparts * part_stride].view(nparts, part_stride)[:, :n].float().sum(0))

    def _on_exchange_stream(self, g, fn):
        """Run ``fn`` with the exchange stream current (device tensors): staging kernels and collectives are issued
        there, ordered after everything the compute stream has enqueued so far, so backward never waits for them."""
        if not g.is_cuda:
            return fn()
        from . import ops
        if getattr(self, "_xs", None) is None:
            self._xs = torch.cuda.Stream(device=g.device)
        self._xs.wait_stream(torch.cuda.current_stream(g.device))
        for st in (self.producers() if self.producers is not None else ()):
            self._xs.wait_stream(st)
        prev = ops.set_stream(self._xs.cuda_stream)
        try:
            with torch.cuda.stream(self._xs):
                return fn()
        finally:
            ops.set_stream(prev)

    def _exchange(self, g: torch.Tensor, lo: int, hi: int):
        if not self._collective:
            works, post = [], None
        else:
            works, post = self._on_exchange_stream(g, lambda: self._exchange_body(g, lo, hi))
        if self.on_bucket is not None:
            self.on_bucket(lo, hi, works, post)
        else:
            self._works.extend(works)
            if post is not None:
                self._post.append(post)

    def _exchange_body(self, g: torch.Tensor, lo: int, hi: int):
        """Launch the SUM of g[lo:hi] over the replicas; appends the async works and, if the result does not land in
        place, the post-step that moves it there."""
        N, n = self.world, hi - lo
        sl = g[lo:hi]
        wire_dt = torch.bfloat16 if self.grad_dtype == "bf16" else torch.float32
        works = []

        def issue(fn, *a, **kw):
            # through plan.host_call: issued now and - while a launch plan records this step - a callback node of the plan;
            # the Work lives in a slot that every replay refills (plan.WorkSlot)
            slot = WorkSlot()

            def call():
                slot.work = fn(*a, async_op=True, **kw)
            host_call(call)
            if self._serial:
                slot.wait()
            else:
                works.append(slot)
            return slot

        if self.exchange == "allreduce":
            if wire_dt == torch.float32:
                issue(dist.all_reduce, sl, op=dist.ReduceOp.SUM)
                post = None
            else:
                wire = self._buf((lo, hi, "wire"), n, wire_dt, g)
                self._pack(sl, wire)
                issue(dist.all_reduce, wire, op=dist.ReduceOp.SUM)
                post = lambda: self._unpack(wire, sl)
        else:
            per = -(-n // N)
            per = -(-per // 8) * 8            # 16-byte pieces in either dtype
            padded = per * N
            wire = self._buf((lo, hi, "wire"), padded, wire_dt, g)
            if padded > n:
                self._fill_zero(wire[n:])
            if wire_dt == torch.float32:
                self._copy(wire[:n], sl)
            else:
                self._pack(sl, wire[:n])
            shard = self._buf((lo, hi, "shard"), per, wire_dt, g)
            if self.exchange == "rs_ag":
                issue(dist.reduce_scatter_tensor, shard, wire, op=dist.ReduceOp.SUM)
                issue(dist.all_gather_into_tensor, wire, shard)
                post = lambda: self._unpack(wire, sl)
            else:  # mesh: piece j goes straight to rank j; fp32 sum of the N received pieces here
                recv = self._buf((lo, hi, "recv"), padded, wire_dt, g)
                w = issue(dist.all_to_all_single, recv, wire)
                # The local fold below reads ``recv``: the exchange stream has to be ordered after the all-to-all.  Under
                # RCCL ``Work.wait()`` IS that stream-order dependency (hipStreamWaitEvent on the current stream - here the
                # exchange stream - against the collective's own stream; the host returns at once:
                # tests/test_rccl_world1_gpu.py times the issue path behind 50 ms of queued device work).  gloo has no
                # streams: there it blocks, as every gloo collective of this class does (``_serial``).
                if not self._serial:
                    works.remove(w)
                    w.wait()
                red = self._buf((lo, hi, "red"), per, torch.float32, g)
                self._unpack(recv, red, nparts=N, part_stride=per)
                full = self._buf((lo, hi, "full"), padded, torch.float32, g)
                issue(dist.all_gather_into_tensor, full, red)
                post = lambda: self._copy(sl, full[:n])
        return works, post

    def _launch(self):
        if self._pend_hi > self._pend_lo:
            if self.pre_launch is not None:
                self.pre_launch()
            # RCCL orders this after everything already enqueued on the compute stream and runs it
            # on its own stream, under the rest of backward
            self._exchange(self._g, self._pend_lo, self._pend_hi)
            self._pend_hi = self._pend_lo

    def all_reduce_gradients(self, g: torch.Tensor):
        """C1: SUM over replicas of the whole gradient arena, bucketed.  If ``begin_gradients``
        opened an overlapped exchange for ``g``, only the not-yet-launched head of the arena is
        sent now; in every case this returns with the compute stream ordered after all buckets."""
        if not self._collective and self.on_bucket is None:
            self._g = None
            return
        if getattr(self, "_g", None) is g:
            self._pend_lo = 0
            self._launch()
        else:
            self._works, self._post = [], []
            for s_, e in self.buckets(g.numel()):
                self._exchange(g, s_, e)
        def finish():
            for w in self._works:
                w.wait()
            for fn in self._post:
                fn()
        if self._collective:
            self._on_exchange_stream(g, finish)
            if g.is_cuda:  # the optimizer (compute stream) is the first consumer
                torch.cuda.current_stream(g.device).wait_stream(self._xs)
        self._g = None
        self._works, self._post = [], []

    def reduce_sum(self, x: torch.Tensor) -> torch.Tensor:
        """C2: strategy.reduce(SUM, per_replica_losses, axis=None) (W:848)."""
        if self.world > 1 or self.force_collectives:
            dist.all_reduce(x, op=dist.ReduceOp.SUM)
        return x

    def broadcast_parameters(self, p: torch.Tensor):
        """C4: replicas start from the chief's initial values."""
        if self.world > 1 or self.force_collectives:
            dist.broadcast(p, src=0)

    def barrier(self):
        if self.world > 1 or self.force_collectives:
            dist.barrier()
