"""Training step and loop with the reference's surface (W:819-848, W:894-958)."""
from __future__ import annotations

import os
import time

import torch

from .data import create_dummy_dataset
from .dist import DataParallelStrategy
from .optim import Adam
from .whisper import create_whisper_model


# Adam slice by slice under backward (optim.Adam.begin_overlapped) is OFF by default: measured on MI355X
# (profiles/r02_adam_under_backward.txt) the step does not get shorter - the concurrent HBM stream slows the GEMMs it
# runs beside by as much as it saves (p8 weight gradients 130 -> 147 us, fc2 dgrad 101 -> 123 us: 9.55 vs 9.56-9.77
# ms/step for grids of 16-512 workgroups).  What it does buy everywhere is the zeroing of the gradient arena inside
# the Adam kernel instead of a separate fill pass.
ADAM_UNDER_BACKWARD = os.environ.get("TMI_ADAM_UNDER_BACKWARD", "0") != "0"
# What does pay (round 3): only the slices whose gradients are final while the chip is idle anyway - the LM head and the
# embedding table under the decoder's backward chain (optim.Adam.begin_early).  With replicas the gradients have to be
# exchanged first: the same two slices run as soon as the bucket that carries them has been summed
# (optim.Adam.begin_early_buckets).  TMI_ADAM_EARLY=0 restores the single update at the end of the step.
ADAM_EARLY = os.environ.get("TMI_ADAM_EARLY", "1") != "0"


# The decoder layers' share of the update (26 % of small-ref) on the second stream UNDER THE NEXT STEP's encoder forward
# (optim.Adam.apply_gradients(late=...)).  Only for callers that say the next thing they do is another step
# (``pipelined=True``: the training loops and the bench); TMI_ADAM_LATE=0 switches it off.
ADAM_LATE = os.environ.get("TMI_ADAM_LATE", "1") != "0"


def distributed_train_step(strategy, model, dist_inputs, optimizer, pipelined=False):
    """W:819-848.  Per replica: forward, backward, apply_gradients (all-reduce SUM, then
    Adam); returns ``strategy.reduce(SUM, per_replica_loss)`` as a 1-element device tensor.
    A replica whose slice of a short final batch is empty contributes zero gradients.
    ``pipelined``: the caller's next access to the model is another step or ``model.finish_late()`` - the decoder layers'
    Adam slice may then still be running on the second stream when this returns."""
    features, labels = dist_inputs
    if hasattr(model, "finish_late") and features.shape[0] == 0:
        model.finish_late()  # (no forward to do the waiting)
    strategy.begin_gradients(model.arena.g)
    # weight gradients run on the model's second stream: the exchange waits for it directly (the compute stream,
    # busy with the dgrad chain, is not joined to it)
    strategy.pre_launch, strategy.producers = None, model.gradient_streams
    overlapped = ADAM_UNDER_BACKWARD and model.device.type == "cuda"
    if overlapped:  # Adam slice by slice as buckets become final (and reduced), under the rest of backward
        optimizer.begin_overlapped(model, strategy)
    # the early / late Adam slices: one schedule for every run of the same job.  Whether there are replicas is a property
    # of the job (world, force_collectives), NOT of ``exchange_off`` - bench.py's "step without the exchange" must be the
    # same schedule minus the collectives
    replicas = strategy.world > 1 or strategy.force_collectives
    can_early = (ADAM_EARLY and not overlapped and model.device.type == "cuda" and features.shape[0] > 0
                 and model._side is not None)
    early, early_buckets = None, False
    if can_early and not replicas:
        early = optimizer.begin_early(model)
    elif can_early and hasattr(model, "early_adam_ranges"):
        # replicas: the same two slices, each as soon as the bucket that carries it has been summed (optim.begin_early_buckets)
        optimizer.begin_early_buckets(model, strategy, model.early_adam_ranges())
        early_buckets = True
    try:
        if features.shape[0] > 0:
            loss = model.forward_backward(features, labels, grad_ready=strategy.gradients_ready, early_update=early)
        else:
            # same reports as a real backward, so this rank's bucket launches match its peers' one for one
            model.report_zero_gradients(strategy.gradients_ready)
            loss = torch.zeros(1, dtype=torch.float32, device=model.device)
    except BaseException:
        optimizer.abort_early()   # (a stale slice list would make the next apply_gradients skip those ranges)
        if early_buckets:
            strategy.on_bucket = None
        raise
    if overlapped:
        optimizer.finish_overlapped(model, strategy)
    else:
        late = None
        if pipelined and ADAM_LATE and (early is not None or early_buckets) and hasattr(model, "late_adam_range"):
            late = model.late_adam_range()
        optimizer.apply_gradients(model, strategy, zero_grad=True, late=late)
        if early_buckets:
            optimizer.finish_early_buckets(model, strategy)
    return strategy.reduce_sum(loss.clone())


# Launch plans (plan.py): the step recorded once per batch shape and replayed from one C call.  On by default on a GPU
# (TMI_PLAN=0: every step issued from Python).  With replicas the collectives are issued through torch.distributed between
# the launches: they are callback nodes of the plan (plan.host_call in DataParallelStrategy._exchange_body), so a replayed
# step of an N-replica job is one C call that comes back to Python once per collective / wait.  TMI_PLAN_REPLICAS=0 keeps
# jobs with replicas eager.
USE_PLAN = os.environ.get("TMI_PLAN", "1") != "0"
PLAN_REPLICAS = os.environ.get("TMI_PLAN_REPLICAS", "1") != "0"


def plan_ok(strategy, model):
    replicas = strategy.world > 1 or strategy.force_collectives
    return (USE_PLAN and (PLAN_REPLICAS or not replicas) and model.device.type == "cuda" and not ADAM_UNDER_BACKWARD)


def planned_step(strategy, model, optimizer, kind="whisper", pipelined=True):
    """``step(*inputs) -> loss`` for the training loops and the bench: ``distributed_train_step`` (kind "whisper": inputs
    features, labels), ``wav2vec2_train_step`` ("wav2vec2": audio, neg_indices) or ``single_train_step`` ("single":
    audio, neg_indices_t), through a launch plan when ``plan_ok`` and eagerly otherwise.  ``step.planned`` is the
    PlannedStep (or None)."""
    if kind == "whisper":
        eager = lambda f, l: distributed_train_step(strategy, model, (f, l), optimizer, pipelined=pipelined)
    elif kind == "wav2vec2":
        eager = lambda audio, negs: wav2vec2_train_step(strategy, model, audio, negs, optimizer, pipelined=pipelined)
    elif kind == "single":
        eager = lambda audio, negs: single_train_step(model, audio, negs, optimizer)
    else:
        raise ValueError(kind)
    if not plan_ok(strategy, model):
        eager.planned = None
        return eager
    from .plan import PlannedStep
    ps = PlannedStep(eager, model, optimizer, loss_buffer=lambda: model.ws["loss"],
                     finish=lambda t: strategy.reduce_sum(t.clone()))
    step = lambda *inputs: ps(*inputs)
    step.planned = ps
    return step


class GraphedTrainStep:
    """``distributed_train_step`` for ONE replica, captured once as a HIP graph and replayed.

    The eager step costs ~8 ms of host time for ~430 launches (ctypes + Python per launch) against
    ~9.5 ms of device time, so the decoder-sized kernels wait for the host; a replay costs the host
    two small copies and one hipGraphLaunch.  Everything the captured launches read that changes per
    step lives in device memory: the batch is copied into static buffers, Adam takes its bias-corrected
    scalars from a 3-float device tensor (``Adam.apply_gradients_dev``).  The capture includes the
    model's weight-gradient stream (forked and joined through events inside the capture).
    Requirements: world == 1, the model already warmed up with the same batch shapes (all buffers,
    workspaces and kernel attributes exist before capture).  A batch of another shape falls back to
    the eager step."""

    def __init__(self, strategy, model, optimizer, example_inputs):
        if strategy.world != 1:
            raise ValueError("GraphedTrainStep is for a single replica (RCCL is not captured)")
        self.strategy, self.model, self.opt = strategy, model, optimizer
        f, l = example_inputs
        self.feats = torch.empty_like(f)
        self.labels = torch.empty_like(l)
        self.scal = torch.zeros(3, dtype=torch.float32, device=model.device)
        self._host = torch.zeros(3, dtype=torch.float32).pin_memory()
        self.feats.copy_(f)
        self.labels.copy_(l)
        self._set_scalars(optimizer.iterations + 1)
        torch.cuda.synchronize()
        model.arena.g_clean = False  # the captured step must contain its own fill of the gradient arena
        optimizer.row_sparse = False  # the captured update is the dense kernel: it does not keep the row-activity flags
        self._ws = model.ws  # the workspace set whose addresses the capture bakes in: keep it alive (model._ws_sets may evict it)
        self.graph = torch.cuda.CUDAGraph()
        m0 = (model.arena.p.clone(), model.arena.m.clone(), model.arena.v.clone())  # capture must not train
        with torch.cuda.graph(self.graph):
            self.loss = model.forward_backward(self.feats, self.labels)
            optimizer.apply_gradients_dev(model, self.scal)
        # (capturing does not execute, but restore anyway in case a runtime ever runs the work eagerly)
        model.arena.p.copy_(m0[0]); model.arena.m.copy_(m0[1]); model.arena.v.copy_(m0[2])

    def _set_scalars(self, step):
        vals = self.opt.scalars(step)
        self._host[0], self._host[1], self._host[2] = vals
        self.scal.copy_(self._host, non_blocking=True)

    def __call__(self, inputs):
        f, l = inputs
        if f.shape != self.feats.shape or l.shape != self.labels.shape:
            return distributed_train_step(self.strategy, self.model, inputs, self.opt)
        self.feats.copy_(f)
        self.labels.copy_(l)
        self.opt.iterations += 1
        self._set_scalars(self.opt.iterations)
        self.graph.replay()
        self.model.arena.g_clean = False  # the replayed step left its gradients in the arena
        return self.loss


def make_train_step(strategy, model, optimizer, example_inputs, warmup=2):
    """Step function ``inputs -> loss``: runs ``warmup`` eager steps on ``example_inputs`` (real
    training steps), then returns the captured graph if TMI_HIP_GRAPH=1 (single replica on a GPU), the
    eager step otherwise.  Opt-in: on ROCm 7.2 a replay of the ~430-node graph costs the host as much
    as the eager launches (7.8 ms) and the device slightly more (10.2 vs 9.5 ms/step), measured with
    tools/graph_check.py."""
    eager = lambda inputs: distributed_train_step(strategy, model, inputs, optimizer)
    if example_inputs[0].shape[0] == 0:
        return eager
    for _ in range(warmup):
        eager(example_inputs)
    if strategy.world != 1 or model.device.type != "cuda" or os.environ.get("TMI_HIP_GRAPH", "0") != "1":
        return eager
    if model._drop_p > 0.0 or model._drop_attn_p > 0.0 or model._drop_act_p > 0.0:
        # the dropout seeds are kernel arguments chosen per step on the host: a captured graph would replay one mask
        return eager
    try:
        return GraphedTrainStep(strategy, model, optimizer, example_inputs)
    except Exception as e:  # capture is an optimisation: report and carry on eagerly
        print(f"[tethys] HIP graph capture failed ({type(e).__name__}: {e}); running the eager step", flush=True)
        torch.cuda.synchronize()
        return eager


def train_whisper(strategy, model_type="small", num_epochs=1, learning_rate=1e-4, *, batch_size=1,
                  num_batches=40, precision="bf16", device="cuda:0", checkpoint_dir=None, log=print, seed=1234,
                  model_overrides=None, seq_len=3000, max_target_length=100, tensor_log_dir=None,
                  resume_from=None, dropout=None, loss_fetch_depth=2):
    """W:894-958: model + Adam(1e-4), dummy dataset, per-step log line, checkpoint at epoch end.
    ``tensor_log_dir``: also write the tensor-size / skewness report of the reference's
    ``whisper_dist_tensorsize.py`` there (computed from shapes, see tensorsize.py).
    ``resume_from``: checkpoint to restore before training (the reference has no restore)."""
    model = create_whisper_model(model_type, device=device, precision=precision, seed=seed,
                                 **(model_overrides or {}))
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    # the reference trains with its Dropout layers active (W:29-30, training=True): on by default on the bf16
    # path, every replica with its own mask stream; ``dropout=False`` is the parity setting
    if dropout is None:
        dropout = precision == "bf16"
    if dropout:
        model.enable_dropout(model.config.dropout, model.config.attention_dropout, seed=seed * 1000003 + strategy.rank)
    optimizer = Adam(learning_rate=learning_rate)
    report = None
    if tensor_log_dir and strategy.rank == 0:
        from .tensorsize import TensorSizeReport
        report = TensorSizeReport(model, batch_size, seq_len, max_target_length, log_dir=tensor_log_dir,
                                  model_label=f"whisper_{model_type}")
        report.log_parameters(0)
    ds = create_dummy_dataset(batch_size, n_mels=model.config.n_mels, seq_len=seq_len,
                              max_target_length=max_target_length, device=device, rank=strategy.rank,
                              world=strategy.world, seed=seed, drop_remainder=strategy.world > 1)
    it = iter(ds)
    step = load_checkpoint(model, optimizer, resume_from, dataset=ds) if resume_from else 0
    losses = []
    start_time = time.time()
    fetch = LossFetcher(device, depth=loss_fetch_depth)

    def emit(final):
        for lv, (i, t0) in final:
            losses.append(lv)
            log(_step_line(i, lv, start_time, t0, time.time()))
            if report is not None:
                report.log_step(i)
    # (one replica on a GPU: the step is recorded once per batch shape and replayed from a launch plan, planned_step)
    train_step = planned_step(strategy, model, optimizer, "whisper", pipelined=True)
    for epoch in range(num_epochs):
        log(f"Epoch {epoch + 1}/{num_epochs}")
        for _ in range(num_batches):
            inputs = next(it)
            step_start = time.time()
            loss = train_step(*inputs)
            # the reference's loss.numpy() (W:951), fetched behind an event so the next step is enqueued meanwhile
            emit(fetch.push(loss, (step, step_start)))
            step += 1
        emit(fetch.drain())
        if checkpoint_dir and strategy.rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"whisper_{model_type}_epoch_{epoch + 1}.pt"), background=True,
                            dataset=ds, step=step)
    if report is not None:
        summ = report.save_final_results()
        report.close()
        log(f"Tiresias tensorsize: {summ['tiresias_tensorsize_mb']:.2f} MB, skewness: {summ['model_skewness']:.3f}")
    # the reference's checkpoint.save sits inside this function (W:916-919, V:1341-1342), i.e. inside the job's JCT window:
    # mid-run writes overlap training, the last one is waited for here, before the caller reads its clock
    wait_for_checkpoints()
    if hasattr(model, "finish_late"):
        model.finish_late()  # whoever looks at the model next does so after its last update
    model.losses = losses
    return model


def save_weights(model, path):
    """model.save_weights (W:1024-1025, V:1437-1439): the parameters only, keyed by the reference's variable paths."""
    torch.save({k: v.detach().cpu().clone() for k, v in model.arena.ref_views(model.arena.p).items()}, path)


class LossFetcher:
    """The per-step ``loss.numpy()`` of the reference's log line (W:951) without stalling the launch queue: each
    step's device scalar is copied to its own pinned slot behind an event, and the line of step i is printed once
    its event has completed - normally while step i+1 is already being enqueued - instead of the host blocking on
    every step.  ``depth`` = how many steps may be in flight before the oldest is waited for (1 = the reference's
    synchronous behaviour)."""

    def __init__(self, device, depth=2):
        self.depth = max(1, int(depth))
        self.cuda = torch.device(device).type == "cuda"
        n = self.depth + 1
        self.slots = [torch.zeros(1, dtype=torch.float32).pin_memory() if self.cuda else torch.zeros(1) for _ in range(n)]
        self.events = [torch.cuda.Event() if self.cuda else None for _ in range(n)]
        self.pending = []  # (slot, payload)
        self._i = 0

    def push(self, loss, payload):
        """Queue ``loss`` (1-element device tensor); returns the (value, payload) pairs that became final."""
        i = self._i
        self._i = (self._i + 1) % len(self.slots)
        self.slots[i].copy_(loss.detach().reshape(1), non_blocking=True)
        if self.cuda:
            self.events[i].record()
        self.pending.append((i, payload))
        out = []
        while self.pending and (len(self.pending) >= self.depth or self._done(self.pending[0][0])):
            out.append(self._pop())
        return out

    def _done(self, i):
        return (not self.cuda) or self.events[i].query()

    def _pop(self):
        i, payload = self.pending.pop(0)
        if self.cuda:
            self.events[i].synchronize()
        return float(self.slots[i].item()), payload

    def drain(self):
        out = []
        while self.pending:
            out.append(self._pop())
        return out


def _step_line(step, lv, start_time, step_start, step_end):
    return (f"Step {step}, Loss: {lv:.4f}, Time: {time.strftime('%H:%M:%S')} "
            f"(경과: {step_end - start_time:.2f}초, 스텝 시간: {step_end - step_start:.2f}초)")


_CKPT_THREADS = []


def wait_for_checkpoints():
    """Block until every background checkpoint write has reached the disk.  Every ``train_*`` loop calls this before it
    returns, so the JCT a job shim measures around it covers the write, as the reference's does (W:1001-1008)."""
    while _CKPT_THREADS:
        _CKPT_THREADS.pop().join()


def save_checkpoint(model, optimizer, path, dataset=None, step=None, background=False):
    """tf.train.Checkpoint(model, optimizer).save (W:919,956): flat-arena dump, plus what a resumed run needs to
    continue rather than replay: the dropout step counter (mask stream), the dataset cursor and the step index.
    ``background``: snapshot p / m / v on the device (stream-ordered after the last update, ~2 ms) and let a thread do the
    device-to-host copy and the file write: the 1.8 GB dump otherwise is 0.7 s of a short job's JCT (``wait_for_checkpoints``)."""
    a = model.arena
    if hasattr(model, "finish_late"):
        model.finish_late()  # a decoder-layer Adam slice left running by a pipelined step: the snapshot comes after it
    meta = {"iterations": optimizer.iterations, "names": a.names, "offsets": a.offsets, "shapes": a.shapes,
            "drop_step": int(getattr(model, "_drop_step", 0)),
            "data_pos": None if dataset is None else int(dataset._pos),
            "step": optimizer.iterations if step is None else int(step)}
    if not (background and a.p.is_cuda):
        torch.save({"p": a.p.cpu(), "m": a.m.cpu(), "v": a.v.cpu(), **meta}, path)
        return
    import threading
    snap = {"p": a.p.clone(), "m": a.m.clone(), "v": a.v.clone()}
    ready = torch.cuda.Event()
    ready.record()

    def write():
        ready.synchronize()
        torch.save({**{k: t.cpu() for k, t in snap.items()}, **meta}, path)
        snap.clear()
    wait_for_checkpoints()  # one writer at a time (and one spare copy of the arenas on the device)
    t = threading.Thread(target=write, name="tethys-checkpoint")
    t.start()
    _CKPT_THREADS.append(t)


def load_checkpoint(model, optimizer, path, dataset=None):
    """The restore path the reference lacks.  Returns the step index to continue from."""
    ck = torch.load(path, map_location="cpu")
    a = model.arena
    if hasattr(model, "finish_late"):
        model.finish_late()
    if ck["names"] != a.names or ck["offsets"] != a.offsets or ck["shapes"] != a.shapes or ck["p"].numel() != a.p.numel():
        raise ValueError("checkpoint layout does not match the model")
    a.p.copy_(ck["p"]); a.m.copy_(ck["m"]); a.v.copy_(ck["v"])
    optimizer.iterations = int(ck["iterations"])
    optimizer.invalidate_row_flags(model)  # the loaded m / v are not the ones the row-activity flags were kept for
    model._drop_step = int(ck.get("drop_step", optimizer.iterations))
    if dataset is not None and ck.get("data_pos") is not None:
        dataset._pos = int(ck["data_pos"])
    model.refresh_shadows()
    return int(ck.get("step", optimizer.iterations))


# ---------------------------------------------------------------------------------------
# Wav2Vec2 (speech_jobs/wav2vec2_dist.py, "V:")
# ---------------------------------------------------------------------------------------
def wav2vec2_train_step(strategy, model, audio, neg_indices, optimizer, pipelined=False, forced_codes=None):
    """V:1186-1260.  Per replica: forward, loss / num_replicas, backward, LOCAL
    clip_by_global_norm(1.0) (V:1243, before the exchange), gradient all-reduce SUM (=> mean),
    Keras clipnorm(1.0) per variable (V:1274, after aggregation), Adam; returns
    strategy.reduce(SUM, scaled_loss)."""
    from . import ops
    a = model.arena
    if audio.shape[0] > 0:
        loss = model.forward_backward(audio, neg_indices, num_replicas=strategy.num_replicas_in_sync, forced_codes=forced_codes)
    else:  # the reference's empty-batch branch (V:1196-1198, V:1250-1254)
        model.finish_late()
        a.g.zero_()
        loss = torch.zeros(1, dtype=torch.float32, device=model.device)
    model._prepare_clip()
    ws = model.ws
    if strategy.num_replicas_in_sync == 1:
        # one replica: both clips are factors of the same raw gradients, so ONE sum-of-squares pass (per variable; the
        # global norm is their sum) feeds an Adam launch that applies c_global * c_variable on the fly
        if audio.shape[0] > 0:
            ops.segment_sumsq_chunks(a.g, model.seg_chunks, ws["clip_vars"], model.n_var)
            # ``pipelined`` (the caller's next access to the model is another step or model.finish_late()): the slice of the
            # update from encoder layer L/6 on runs on the second stream under the next step's conv stack and first layers
            late = model._late_row if (pipelined and ADAM_LATE and model._side is not None) else None
            optimizer.apply_gradients_clipped(model, model.seg_chunks, ws["clip_vars"], model.n_var, clip_global=1.0,
                                              clip_each=1.0, zero_grad=True, late_row=late)
        else:
            optimizer.apply_gradients(model, None, zero_grad=True)
    else:
        if audio.shape[0] > 0:  # V:1243: local, before the exchange (the clipped gradients are what is summed)
            ops.segment_sumsq(a.g, model.seg_all, ws["clip_all"], 1)
            ops.segment_clip(a.g, model.seg_all, ws["clip_all"], 1, 1.0)
        strategy.all_reduce_gradients(a.g)
        ops.segment_sumsq(a.g, model.seg_vars, ws["clip_vars"], model.n_var)  # V:1274 on the aggregated gradients
        optimizer.apply_gradients_clipped(model, model.seg_chunks, ws["clip_vars"], model.n_var, clip_each=1.0, zero_grad=True)
    model._pack_pos()
    return strategy.reduce_sum(loss.clone())


def train_wav2vec2(strategy, model_type="pretraining", model_size="small", num_epochs=1, learning_rate=3e-5, *,
                   batch_size=1, num_batches=5, precision="bf16", device="cuda:0", checkpoint_dir=None, log=print,
                   seed=1234, clip_samples=32000, model_overrides=None, dropout=None, resume_from=None,
                   loss_fetch_depth=2, epoch_label="Epoch", init_batches=0, checkpoint_stem=None):
    """V:1263-1376: model + Adam(3e-5, eps 1e-8, clipnorm 1), 50 x 2 s dummy clips, per-step log
    line, checkpoint every 50 steps and at the end.  speech_jobs/wav2vec2_single.py ("U:") is this loop on one
    replica (U:1118-1175 is V:1186-1260 without the strategy) with ``epoch_label`` "에포크" (U:1240),
    ``init_batches`` 1 (the first batch is consumed by the weight-building call, U:1188-1196) and checkpoints
    named model_step_N / model_epoch_N (U:1272-1275)."""
    import numpy as np
    from .data import W2VDummyDataset
    from .wav2vec2 import create_full_model, sample_negative_indices
    model = create_full_model(model_type, model_size, device=device, precision=precision, seed=seed,
                              **(model_overrides or {}))
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    if dropout is None:  # the reference trains with its Dropout layers active (V:69-71); rates 0 is the parity setting
        dropout = precision == "bf16"
    if dropout:
        c = model.config
        model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=seed * 1000003 + strategy.rank, act_p=c.activation_dropout)
    optimizer = Adam(learning_rate=learning_rate, epsilon=1e-8)
    ds = W2VDummyDataset(batch_size, length=clip_samples, device=device, rank=strategy.rank, world=strategy.world, seed=seed)
    rng = np.random.default_rng(seed + 1)
    it = iter(ds)
    step, losses = 0, []
    fetch = LossFetcher(device, depth=loss_fetch_depth)
    for _ in range(init_batches):
        next(it)
    stem = checkpoint_stem or f"wav2vec2_{model_type}"

    def emit(final):
        for lv, (i, t0) in final:
            losses.append(lv)
            log(_step_line(i, lv, start_time, t0, time.time()))
    if resume_from:
        step = load_checkpoint(model, optimizer, resume_from, dataset=ds)
        model._prepare(batch_size, clip_samples)
        for _ in range(step):  # the negative-index stream continues where the saved run stopped
            sample_negative_indices(rng, ds.global_batch, model.T, model.config.num_negatives)
    start_time = time.time()
    train_step = planned_step(strategy, model, optimizer, "wav2vec2", pipelined=True)
    for epoch in range(num_epochs):
        log(f"{epoch_label} {epoch + 1}/{num_epochs}")
        for _ in range(num_batches):
            audio = next(it)
            model._prepare(audio.shape[0], audio.shape[1])
            # one draw per GLOBAL batch row, sliced per replica, so N replicas see what 1 would
            neg_all = sample_negative_indices(rng, ds.global_batch, model.T, model.config.num_negatives)
            neg = torch.from_numpy(neg_all[strategy.rank * batch_size:(strategy.rank + 1) * batch_size]).to(device)
            step_start = time.time()
            loss = train_step(audio, neg)
            emit(fetch.push(loss, (step, step_start)))
            step += 1
            if checkpoint_dir and strategy.rank == 0 and step % 50 == 0:
                emit(fetch.drain())
                os.makedirs(checkpoint_dir, exist_ok=True)
                save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"{stem}_step_{step}.pt"), background=True,
                                dataset=ds, step=step)
        emit(fetch.drain())
        if checkpoint_dir and strategy.rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"{stem}_epoch_{epoch + 1}.pt"), background=True,
                            dataset=ds, step=step)
    # the reference's checkpoint.save sits inside this function (W:916-919, V:1341-1342), i.e. inside the job's JCT window:
    # mid-run writes overlap training, the last one is waited for here, before the caller reads its clock
    wait_for_checkpoints()
    model.finish_late()  # whoever looks at the model next does so after its last update
    model.losses = losses
    return model


# ---------------------------------------------------------------------------------------
# speech_jobs/whisper_single.py ("S:") - BASELINE config #1 as the file is named: a single-device Wav2Vec2-base
# pre-training job.  Same model as V: base; the step has no replica scaling, no clipping, Adam's
# default epsilon, roll-based negatives and 5 s clips.
# ---------------------------------------------------------------------------------------
def single_train_step(model, audio, neg_indices_t, optimizer):
    """S:1143-1180: forward, loss = contrastive + 0.1 * (-perplexity), gradients, Adam.  ``neg_indices_t`` [T, N]
    (wav2vec2.sample_negative_indices_roll)."""
    model.neg_per_time = True
    loss = model.forward_backward(audio, neg_indices_t, num_replicas=1)
    optimizer.apply_gradients(model, None, zero_grad=True)
    model._pack_pos()
    return loss


def train_wav2vec2_single(model_type="pretraining", num_epochs=1, learning_rate=3e-5, *, batch_size=4, num_batches=40,
                          precision="bf16", device="cuda:0", checkpoint_dir=None, log=print, seed=1234,
                          clip_samples=80000, model_overrides=None, dropout=None, model_size="base", loss_fetch_depth=2):
    """S:1183-1263: Wav2Vec2-base + Adam(3e-5) (epsilon 1e-7, no clipnorm: S:1189), 50 x 5 s dummy clips batched
    WITHOUT drop_remainder (S:1094-1111), per-step log line, checkpoint at the end of the epoch."""
    import numpy as np
    from .data import W2VDummyDataset
    from .wav2vec2 import create_full_model, sample_negative_indices_roll
    model = create_full_model(model_type, model_size, device=device, precision=precision, seed=seed,
                              **(model_overrides or {}))
    model.refresh_shadows()
    if dropout is None:
        dropout = precision == "bf16"
    if dropout:
        c = model.config
        model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=seed * 1000003, act_p=c.activation_dropout)
    optimizer = Adam(learning_rate=learning_rate)  # Keras default epsilon 1e-7
    ds = W2VDummyDataset(batch_size, length=clip_samples, device=device, seed=seed, drop_remainder=False)
    rng = np.random.default_rng(42)  # the reference asks tf.random.shuffle for seed 42 (S:799)
    it = iter(ds)
    step, losses = 0, []
    start_time = time.time()
    fetch = LossFetcher(device, depth=loss_fetch_depth)

    def emit(final):
        for lv, (i, t0) in final:
            losses.append(lv)
            log(_step_line(i, lv, start_time, t0, time.time()))
    for epoch in range(num_epochs):
        log(f"Epoch {epoch + 1}/{num_epochs}")
        for _ in range(num_batches):
            audio = next(it)
            model._prepare(audio.shape[0], audio.shape[1])
            neg = torch.from_numpy(sample_negative_indices_roll(rng, model.T, model.config.num_negatives)).to(device)
            step_start = time.time()
            loss = single_train_step(model, audio, neg, optimizer)
            emit(fetch.push(loss, (step, step_start)))
            step += 1
        emit(fetch.drain())
        if checkpoint_dir:
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"model_epoch_{epoch + 1}.pt"), dataset=ds, step=step,
                            background=True)
    # the reference's checkpoint.save sits inside this function (W:916-919, V:1341-1342), i.e. inside the job's JCT window:
    # mid-run writes overlap training, the last one is waited for here, before the caller reads its clock
    wait_for_checkpoints()
    model.losses = losses
    return model


# ---------------------------------------------------------------------------------------
# stable_jobs/wav2vec2_dist.py ("T:"): the S: model and step under MultiWorkerMirroredStrategy.  (stable_jobs/whisper_dist.py
# is byte-identical to speech_jobs/whisper_dist.py.)
# ---------------------------------------------------------------------------------------
def stable_wav2vec2_train_step(strategy, model, audio, neg_indices_t, optimizer):
    """T:1143-1190.  Per replica: forward, loss = contrastive + 0.1 * (-perplexity) as the mean over the LOCAL rows,
    gradients; ``apply_gradients`` under the strategy all-reduces them with SUM (no 1/N, nothing clipped), Adam; returns
    strategy.reduce(SUM, loss) (T:1190).  A replica whose slice of a short final batch is empty sends zeros through the
    same collectives."""
    model.neg_per_time = True
    if audio.shape[0] > 0:
        loss = model.forward_backward(audio, neg_indices_t, num_replicas=1)
    else:
        model.arena.g.zero_()
        loss = torch.zeros(1, dtype=torch.float32, device=model.device)
    strategy.all_reduce_gradients(model.arena.g)
    optimizer.apply_gradients(model, None, zero_grad=True)
    model._pack_pos()
    return strategy.reduce_sum(loss.clone())


def train_wav2vec2_stable(strategy, model_type="pretraining", num_epochs=1, learning_rate=3e-5, *, batch_size=1,
                          num_batches=40, precision="bf16", device="cuda:0", checkpoint_dir=None, log=print, seed=1234,
                          clip_samples=80000, model_overrides=None, dropout=None, model_size="base", loss_fetch_depth=2):
    """T:1193-1268: Wav2Vec2-base + Adam(3e-5) (Keras default epsilon, no clipnorm: T:1200), 50 x 5 s dummy clips batched
    by the GLOBAL batch without drop_remainder (T:1094-1111, T:1226), per-step log line, checkpoint at the end of the epoch."""
    import numpy as np
    from .data import W2VDummyDataset
    from .wav2vec2 import create_full_model, sample_negative_indices_roll
    model = create_full_model(model_type, model_size, device=device, precision=precision, seed=seed,
                              **(model_overrides or {}))
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    if dropout is None:
        dropout = precision == "bf16"
    if dropout:
        c = model.config
        model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=seed * 1000003 + strategy.rank, act_p=c.activation_dropout)
    optimizer = Adam(learning_rate=learning_rate)  # Keras default epsilon 1e-7
    ds = W2VDummyDataset(batch_size, length=clip_samples, device=device, rank=strategy.rank, world=strategy.world, seed=seed,
                         drop_remainder=False)
    rng = np.random.default_rng(42)  # the reference asks tf.random.shuffle for seed 42 (T:799)
    it = iter(ds)
    step, losses = 0, []
    start_time = time.time()
    fetch = LossFetcher(device, depth=loss_fetch_depth)

    def emit(final):
        for lv, (i, t0) in final:
            losses.append(lv)
            log(_step_line(i, lv, start_time, t0, time.time()))
    for epoch in range(num_epochs):
        log(f"Epoch {epoch + 1}/{num_epochs}")
        for _ in range(num_batches):
            audio = next(it)
            if audio.shape[0] > 0 or model._ws_key is None:  # (an empty slice runs no kernels; T only depends on the clip length)
                model._prepare(max(1, audio.shape[0]), clip_samples)
            # one permutation per replica per step, in rank order (every rank advances the same stream)
            draws = [sample_negative_indices_roll(rng, model.T, model.config.num_negatives) for _ in range(strategy.world)]
            neg = torch.from_numpy(draws[strategy.rank]).to(device)
            step_start = time.time()
            loss = stable_wav2vec2_train_step(strategy, model, audio, neg, optimizer)
            emit(fetch.push(loss, (step, step_start)))
            step += 1
        emit(fetch.drain())
        if checkpoint_dir and strategy.rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"model_epoch_{epoch + 1}.pt"), dataset=ds, step=step,
                            background=True)
    # the reference's checkpoint.save sits inside this function (W:916-919, V:1341-1342), i.e. inside the job's JCT window:
    # mid-run writes overlap training, the last one is waited for here, before the caller reads its clock
    wait_for_checkpoints()
    model.losses = losses
    return model

