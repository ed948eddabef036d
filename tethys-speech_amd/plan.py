"""Launch plans: one training step recorded once, then replayed from a single C call.

The reference's step is a ``@tf.function`` (speech_jobs/whisper_dist.py:818-819): traced on its first call, replayed
afterwards with no Python between its ops.  The step here is ~255 (Whisper) / ~340 (Wav2Vec2) launches issued one
``ctypes`` call at a time; ``LaunchPlan`` records the calls of one real step inside the library (``tmi_plan_*`` in
include/tethys_mi.h) and ``PlannedStep`` replays them, patching what changes from step to step (the dropout seeds, the
Adam step number) through two scalars.  A hipGraph does the same job worse on ROCm 7.2 (a replay costs the host as much
as the eager launches, tools/graph_check.py) and cannot take the per-step seeds.

What a plan contains besides the library's launches: every ``torch.cuda.Event.record`` / ``Event.wait`` made while
recording (the step orders its weight-gradient stream and its early / late Adam slices with events; ``Stream.wait_event``,
``wait_stream`` and ``record_event`` all go through those two methods), reported to the library by patching the two
methods for the duration of the recording.  The plan keeps the event objects alive; the replay re-records and re-waits
the same handles.  What a plan must NOT contain: device work issued by anything else (a torch kernel between two
launches would run while recording and be missing from every replay) - the step code uses ``ops.fill_zero`` /
``ops.copy`` for its fills and copies, and tests/test_plan_gpu.py compares replayed steps with eager ones bit for bit.

Replicas: the gradient exchange is issued through ``torch.distributed`` (RCCL), which the library cannot record.  Those
calls go through ``host_call(fn)``: ``fn`` runs at once and, while a plan is recording, becomes a CALLBACK node - the
replay calls it at the same place of the launch sequence, with the torch stream that was current when it was recorded
(a collective is ordered after the current stream, ``Work.wait()`` orders the current stream after the collective).  A
replayed step of an N-replica job is then the recorded launches plus one Python call per collective / wait
(``DataParallelStrategy._exchange_body``), instead of ~255 ``ctypes`` launches.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from ._lib import check, lib

SEED_STEP = 0x9E3779B97F4A7C15  # KernelBlocks._site_seed: the per-step stride of every dropout site's seed

_recording = None   # the LaunchPlan recording on this thread (one at a time: tmi_plan_begin refuses a second)
_CALLBACK = C.CFUNCTYPE(None)


def host_call(fn):
    """Run ``fn()`` now; while a plan is recording also make it a callback node of that plan (see the module docstring).
    ``fn`` must be replayable: everything it touches lives in objects that outlive the step (buffers allocated once,
    ``WorkSlot``s), and it returns nothing."""
    plan = _recording
    if plan is not None:
        plan.add_callback(fn)
    fn()


class WorkSlot:
    """The ``Work`` of a collective issued through ``host_call``: a replay puts its own Work object in the slot, and
    ``wait()`` (itself a host call) waits for whichever is there."""
    __slots__ = ("work",)

    def __init__(self):
        self.work = None

    def wait(self):
        host_call(lambda: self.work.wait())


class LaunchPlan:
    def __init__(self):
        h = C.c_void_p()
        check(lib().tmi_plan_create(C.byref(h)), "tmi_plan_create")
        self._h = h
        self._keep = []       # events (and anything else) whose handles the plan refers to
        self.recorded = False
        self.callbacks = 0
        self._error = None    # an exception raised inside a callback node during the last replay

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().tmi_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def add_callback(self, fn):
        """``fn()`` becomes a node of the plan being recorded: called by every replay at this place of the sequence, with
        the torch stream that is current NOW made current around it."""
        if torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            ctx = lambda: torch.cuda.stream(cur)
        else:
            ctx = contextlib.nullcontext

        def node():
            if self._error is not None:
                return       # a previous node of this replay failed: issue nothing more (the caller raises after the replay)
            try:
                with ctx():
                    fn()
            except BaseException as e:   # (an exception cannot cross the C frames of the replay loop)
                self._error = e

        cb = _CALLBACK(node)
        self._keep.append(cb)
        self.callbacks += 1
        check(lib().tmi_plan_note_callback(C.cast(cb, C.c_void_p)), "tmi_plan_note_callback")

    @contextlib.contextmanager
    def recording(self):
        """Everything the library launches inside this block (on this thread) is executed AND recorded, together with the
        torch event traffic between the launches."""
        L = lib()
        Event = torch.cuda.Event
        rec0, wait0 = Event.record, Event.wait
        keep = self._keep

        def record(ev, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            rec0(ev, stream)
            keep.append(ev)
            L.tmi_plan_note_event_record(ev.cuda_event, stream.cuda_stream)

        def wait(ev, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            wait0(ev, stream)
            keep.append(ev)
            L.tmi_plan_note_stream_wait(stream.cuda_stream, ev.cuda_event)

        global _recording
        check(L.tmi_plan_begin(self._h), "tmi_plan_begin")
        Event.record, Event.wait = record, wait
        _recording = self
        try:
            yield self
        finally:
            _recording = None
            Event.record, Event.wait = rec0, wait0
            check(L.tmi_plan_end(self._h), "tmi_plan_end")
        self.recorded = True

    def replay(self, seed_delta=0, step_delta=0):
        self._error = None
        rc = lib().tmi_plan_replay(self._h, seed_delta & 0xFFFFFFFFFFFFFFFF, step_delta)
        if self._error is not None:
            e, self._error = self._error, None
            raise e
        check(rc, "tmi_plan_replay")

    @property
    def nodes(self):
        return int(lib().tmi_plan_size(self._h, 0))

    @property
    def launches(self):
        return int(lib().tmi_plan_size(self._h, 1))


class PlannedStep:
    """``step(*inputs) -> loss`` for ONE replica, replayed from a launch plan.

    ``eager(*inputs)`` is the step as the host code issues it (``train.distributed_train_step`` / ``wav2vec2_train_step``
    bound to their strategy, model and optimizer); it must return a fresh device tensor derived from ``loss_buffer()``,
    the model's own (static) loss scalar.  Per input signature (shapes + dtypes): the first ``warm`` calls run eagerly
    (workspaces, streams and the steady state of the early / late Adam slices come into being), the next call runs
    eagerly on STATIC copies of the inputs while the plan records it, every later call copies its inputs into those
    buffers and replays.  ``finish(loss_tensor)`` turns the static loss scalar into the step's return value (clone +
    the strategy's reduce).  Python-side counters the eager step advances are advanced here too (``model._drop_step``,
    ``optimizer.iterations``); everything else the step leaves behind on the host (event bookkeeping, the ``g_clean``
    flag) is the same after every steady-state step, which is what makes the recording replayable."""

    def __init__(self, eager, model, optimizer, loss_buffer, finish=None, warm=2):
        self.eager, self.model, self.opt = eager, model, optimizer
        self.loss_buffer, self.finish, self.warm = loss_buffer, finish or (lambda t: t.clone()), warm
        self._by_sig = {}
        self.replays = 0

    def __call__(self, *inputs):
        from . import ops
        sig = tuple((tuple(t.shape), t.dtype) for t in inputs)
        st = self._by_sig.get(sig)
        if st is None:
            st = self._by_sig[sig] = {"seen": 0, "plan": None}
        if st["plan"] is None:
            if st["seen"] < self.warm or any(t.numel() == 0 for t in inputs):
                st["seen"] += 1
                return self.eager(*inputs)
            # record: a real step, on static input buffers
            st["static"] = [torch.empty_like(t) for t in inputs]
            for s_, t in zip(st["static"], inputs):
                s_.copy_(t)
            plan = LaunchPlan()
            st["drop0"], st["it0"] = int(getattr(self.model, "_drop_step", 0)), int(self.opt.iterations)
            with plan.recording():
                out = self.eager(*st["static"])
            st["loss"] = self.loss_buffer()
            st["ws"] = getattr(self.model, "ws", None)  # the workspace set whose addresses the plan bakes in
            st["plan"] = plan
            return out
        for s_, t in zip(st["static"], inputs):
            ops.copy(s_, t.contiguous())
        m = self.model
        drop_now = int(getattr(m, "_drop_step", 0))
        st["plan"].replay((drop_now - st["drop0"]) * SEED_STEP, int(self.opt.iterations) - st["it0"])
        if hasattr(m, "_drop_step"):
            m._drop_step = drop_now + 1
        self.opt.iterations += 1
        self.replays += 1
        return self.finish(st["loss"])
