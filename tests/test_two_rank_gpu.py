"""The N > 1 training step on a GPU: two ranks (gloo moves the CUDA gradient buckets; RCCL cannot put two
ranks on one device) share cuda:0 and run ``distributed_train_step`` with small buckets, so the overlapped
exchange (ranges reported during backward, weight-gradient stream joined before every launch, async works
waited before Adam) is exercised end to end.  Reference semantics (W:829-836): gradients are SUMMED over
replicas, no 1/N; both ranks must end with identical parameters, equal to one process that accumulates both
shards' gradients and applies Adam once per step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
          encoder_layers=2, decoder_layers=2, n_mels=16, n_ctx=32, decoder_start_token_id=150, max_target_positions=32)
STEPS = 3


def _batches():
    rng = np.random.default_rng(7)
    return [[(rng.standard_normal((2, 16, 48)).astype(np.float32), rng.integers(0, 150, (2, 12)).astype(np.int32))
             for _ in range(STEPS)] for _ in range(2)]  # [rank][step]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, whisper
    torch.cuda.set_device(0)
    dev = "cuda:0"
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=256 * 1024)
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=11 + rank, **KW)  # differ per rank
    strat.broadcast_parameters(model.arena.p)  # C4: everyone starts from rank 0's values
    model.refresh_shadows()
    opt = optim.Adam(1e-3)
    losses = []
    launched = 0
    for f, l in _batches()[rank]:
        out = train.distributed_train_step(strat, model, (torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)), opt)
        losses.append(float(out.item()))
    torch.cuda.synchronize()
    q.put((rank, model.arena.p.cpu().numpy(), losses))
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_accumulated_single_process(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, p0, l0), (_, p1, l1) = res
    assert np.array_equal(p0, p1), "replicas diverged"
    assert l0 == l1  # C2: the reduced (summed) loss is the same on every rank

    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import optim, whisper
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=11, **KW)
    opt = optim.Adam(1e-3)
    b = _batches()
    ref_losses = []
    for s in range(STEPS):
        tot = torch.zeros_like(model.arena.g)
        lsum = 0.0
        for r in range(2):
            f, l = b[r][s]
            loss = model.forward_backward(torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev))
            tot += model.arena.g
            lsum += float(loss.item())
        model.arena.g.copy_(tot)
        opt.apply_gradients(model)
        ref_losses.append(lsum)
    ref = model.arena.p.cpu().numpy()
    assert np.allclose(l0, ref_losses, rtol=1e-5, atol=1e-6), (l0, ref_losses)
    err = np.abs(p0 - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err


def _six_worker(rank, port, q, planned):
    """Six same-shape steps: issued from Python, or through ``train.planned_step`` (two eager, one recorded, three replayed
    with the gloo collectives and their waits as callback nodes of the plan)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, whisper
    torch.cuda.set_device(0)
    dev = "cuda:0"
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=256 * 1024)
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=11, **KW)
    strat.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    opt = optim.Adam(1e-3)
    if planned:
        step = train.planned_step(strat, model, opt, "whisper", pipelined=True)
        assert step.planned is not None
    else:
        step = lambda f, l: train.distributed_train_step(strat, model, (f, l), opt, pipelined=True)
    rng = np.random.default_rng(70 + rank)
    losses = []
    for _ in range(6):
        f = rng.standard_normal((2, 16, 48)).astype(np.float32)
        l = rng.integers(0, 150, (2, 12)).astype(np.int32)
        losses.append(float(step(torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)).item()))
    model.finish_late()
    torch.cuda.synchronize()
    info = None
    if planned:
        pl = [v["plan"] for v in step.planned._by_sig.values() if v.get("plan") is not None]
        info = (step.planned.replays, [p_.callbacks for p_ in pl])
    q.put((rank, model.arena.p.cpu().numpy(), losses, info))
    torch.distributed.destroy_process_group()


def test_two_rank_launch_plan_replays_the_exchange(dev):
    """plan.host_call: a replayed step of a job with replicas issues its collectives from callback nodes.  Both ranks of the
    planned job stay identical, and equal to the eager job (same data) up to the step's own atomics noise."""
    res = {}
    for planned in (False, True):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_six_worker, args=(r, port, q, planned)) for r in range(2)]
        for p_ in procs:
            p_.start()
        res[planned] = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
        for p_ in procs:
            p_.join(60)
    (_, e0, le0, _), (_, e1, le1, _) = res[False]
    (_, p0, lp0, i0), (_, p1, lp1, i1) = res[True]
    assert np.array_equal(e0, e1) and np.array_equal(p0, p1), "replicas diverged"
    assert le0 == le1 and lp0 == lp1
    for replays, callbacks in (i0, i1):
        assert replays == 3 and len(callbacks) == 1 and callbacks[0] >= 2, (replays, callbacks)
    assert np.allclose(lp0, le0, rtol=1e-5, atol=1e-6), (lp0, le0)
    assert np.abs(p0 - e0).max() / np.abs(e0).max() <= 1e-5


def _ragged_worker(rank, port, q):
    """Step 1 is the short final batch of a pass (W:812-815 has no drop_remainder): rank 0 gets one sample,
    rank 1 none.  ADVICE r1 (train.py:22): the empty replica must issue the same bucket collectives."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, whisper
    torch.cuda.set_device(0)
    dev = "cuda:0"
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=256 * 1024)
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=11, **KW)
    strat.broadcast_parameters(model.arena.p)
    opt = optim.Adam(1e-3)
    b = _batches()
    losses = []
    for s in range(STEPS):
        f, l = b[rank][s]
        if s == 1:
            f, l = (f[:1], l[:1]) if rank == 0 else (f[:0], l[:0])
        out = train.distributed_train_step(strat, model, (torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)), opt)
        losses.append(float(out.item()))
    torch.cuda.synchronize()
    q.put((rank, model.arena.p.cpu().numpy(), losses))
    torch.distributed.destroy_process_group()


def test_two_rank_ragged_final_batch_with_an_empty_replica(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, p0, l0), (_, p1, l1) = res
    assert np.array_equal(p0, p1) and l0 == l1

    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import optim, whisper
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=11, **KW)
    opt = optim.Adam(1e-3)
    b = _batches()
    ref_losses = []
    for s in range(STEPS):
        tot = torch.zeros_like(model.arena.g)
        lsum = 0.0
        for r in range(2):
            f, l = b[r][s]
            if s == 1:
                if r == 1:
                    continue
                f, l = f[:1], l[:1]
            loss = model.forward_backward(torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev))
            tot += model.arena.g
            lsum += float(loss.item())
        model.arena.g.copy_(tot)
        opt.apply_gradients(model)
        ref_losses.append(lsum)
    assert np.allclose(l0, ref_losses, rtol=1e-5, atol=1e-6), (l0, ref_losses)
    ref = model.arena.p.cpu().numpy()
    assert np.abs(p0 - ref).max() / np.abs(ref).max() <= 1e-5


# ---- stable_jobs/wav2vec2_dist.py ("T:"): the whisper_single step under the strategy, against the fp64 oracle
def _stable_setup():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_wav2vec2_gpu as TW
    return TW


def _stable_worker(rank, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    TW = _stable_setup()
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train
    from tethys_speech_amd.data import W2VDummyDataset
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=64 * 1024)
    model, ocfg, _ = TW.build("fp32", dev)
    strat.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    opt = optim.Adam(learning_rate=1e-3)
    ds = W2VDummyDataset(2, length=400, device=dev, rank=rank, world=2, seed=3, num_samples=5, drop_remainder=False)
    ds.audio = torch.from_numpy(TW.V.create_dummy_pool(seed=3, num_samples=5, length=400)).to(dev)  # the oracle's pool
    T = TW.V.feature_lengths(ocfg, 400)[-1]
    rng = np.random.default_rng(42)
    losses, sizes = [], []
    it = iter(ds)
    for _ in range(4):
        a = next(it)
        sizes.append(int(a.shape[0]))
        draws = [TW.V.sample_negative_indices_roll(rng, T, ocfg.num_negatives) for _ in range(2)]
        out = train.stable_wav2vec2_train_step(strat, model, a, torch.from_numpy(draws[rank]).to(dev), opt)
        losses.append(float(out.item()))
    torch.cuda.synchronize()
    q.put((rank, model.arena.p.cpu().numpy(), losses, sizes))
    torch.distributed.destroy_process_group()


def test_two_rank_stable_wav2vec2_step_matches_oracle(dev):
    """T:1143-1190 on two replicas, global batch 4 over a pool of 5 clips (4, then a short batch of 1: rank 0 one row,
    rank 1 none), against oracle.train_steps_stable in fp64."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stable_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, p0, l0, s0), (_, p1, l1, s1) = res
    assert np.array_equal(p0, p1) and l0 == l1
    assert s0 == [2, 1, 2, 1] and s1 == [2, 0, 2, 0]
    TW = _stable_setup()
    _, ocfg, params = TW.build("fp32", dev)
    pool = TW.V.create_dummy_pool(seed=3, num_samples=5, length=400)
    ref, _ = TW.V.train_steps_stable(ocfg, params, pool, 2, 2, 4, seed=42, lr=1e-3)
    assert max(abs(x - y) / max(1.0, abs(y)) for x, y in zip(l0, ref)) <= 2e-4, (l0, ref)


# ---- speech_jobs/wav2vec2_dist.py ("V:") on two replicas: loss / N, local global-norm clip before the exchange,
# gradient SUM, per-variable clipnorm after it, Adam eps 1e-8 - against oracle.train_steps(n_replicas=2) in fp64
def _v_worker(rank, port, q, planned=None, steps=4):
    """``planned``: None = the step function called directly (the oracle comparison below); False / True = ``steps`` steps
    through ``train.planned_step`` with launch plans off / on (plan.host_call: the collectives as callback nodes)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    TW = _stable_setup()
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=64 * 1024)
    model, ocfg, _ = TW.build("fp32", dev)
    strat.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    opt = optim.Adam(learning_rate=1e-3, epsilon=1e-8)
    pool = TW.V.create_dummy_pool(seed=3, num_samples=8, length=400)
    T = TW.V.feature_lengths(ocfg, 400)[-1]
    rng = np.random.default_rng(77)
    it = TW.V.batches(pool, 4)
    losses = []
    if planned is None:
        step = lambda a_, n_: train.wav2vec2_train_step(strat, model, a_, n_, opt)
    else:
        train.USE_PLAN = bool(planned)
        step = train.planned_step(strat, model, opt, "wav2vec2", pipelined=False)
        assert (step.planned is not None) == bool(planned)
    for _ in range(steps):
        a = next(it)
        neg = TW.V.sample_negative_indices(rng, 4, T, ocfg.num_negatives)
        sl = slice(2 * rank, 2 * rank + 2)
        out = step(torch.from_numpy(np.ascontiguousarray(a[sl])).to(dev), torch.from_numpy(np.ascontiguousarray(neg[sl])).to(dev))
        losses.append(float(out.item()))
    torch.cuda.synchronize()
    if planned:
        pl = [v["plan"] for v in step.planned._by_sig.values() if v.get("plan") is not None]
        q.put((rank, model.arena.p.cpu().numpy(), losses, (step.planned.replays, [p_.callbacks for p_ in pl])))
    else:
        q.put((rank, model.arena.p.cpu().numpy(), losses))
    torch.distributed.destroy_process_group()


def test_two_rank_wav2vec2_launch_plan_replays_the_exchange(dev):
    """The V: step with replicas (local clip, bucketed SUM, per-variable clipnorm, Adam) replayed from a launch plan."""
    res = {}
    for planned in (False, True):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_v_worker, args=(r, port, q, planned, 6)) for r in range(2)]
        for p_ in procs:
            p_.start()
        res[planned] = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
        for p_ in procs:
            p_.join(60)
    (_, e0, le0), (_, e1, le1) = res[False]
    (_, p0, lp0, i0), (_, p1, lp1, i1) = res[True]
    assert np.array_equal(e0, e1) and np.array_equal(p0, p1), "replicas diverged"
    assert le0 == le1 and lp0 == lp1
    for replays, callbacks in (i0, i1):
        assert replays == 3 and len(callbacks) == 1 and callbacks[0] >= 2, (replays, callbacks)
    assert np.allclose(lp0, le0, rtol=1e-4, atol=1e-5), (lp0, le0)
    dp = np.abs(p0 - e0)   # (the step's own fp32 atomics make two eager runs differ in the last bits too)
    assert float(np.median(dp)) <= 1e-6 and float((dp > 1e-4).mean()) <= 2e-3, (float(np.median(dp)), float(dp.max()))


def test_two_rank_wav2vec2_step_matches_oracle(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_v_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, p0, l0), (_, p1, l1) = res
    assert np.array_equal(p0, p1) and l0 == l1
    TW = _stable_setup()
    _, ocfg, params = TW.build("fp32", dev)
    pool = TW.V.create_dummy_pool(seed=3, num_samples=8, length=400)
    ref, _ = TW.V.train_steps(ocfg, params, pool, 2, 4, seed=77, n_replicas=2, lr=1e-3)  # (updates ``params`` in place)
    assert max(abs(x - y) / max(1.0, abs(y)) for x, y in zip(l0, ref)) <= 2e-4, (l0, ref)
    import tethys_speech_amd  # noqa: F401
    model, _, _ = TW.build("fp32", dev)
    model.arena.p.copy_(torch.from_numpy(p0).to(dev))
    worst = 0.0
    for k, v in model.arena.ref_views(model.arena.p).items():
        r = params[k]
        worst = max(worst, float((v.double().cpu() - r).abs().max() / max(float(r.abs().max()), 1e-3)))
    # Adam divides by sqrt(v): a parameter whose true gradient is ~0 (k_proj.bias, project_q beta) moves by ~lr either way
    assert worst <= 4 * 4 * 1e-3, worst


# ---- speech_jobs/whisper_dist.py (W:819-848) on two replicas against the fp64 oracle's multi-replica loop, the dataset's
# own slicing (pool of 6 clips, global batch 4: 4, then a short batch of 2 - rank 0 two rows, rank 1 none)
def _whisper_oracle_worker(rank, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import test_whisper_step_gpu as TWH
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train
    from tethys_speech_amd.data import create_dummy_dataset
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=128 * 1024)
    cfg_kw = TWH.small_cfg()
    model, _, _ = TWH.build("fp32", cfg_kw, dev)
    strat.broadcast_parameters(model.arena.p)
    opt = optim.Adam(1e-3)
    ds = create_dummy_dataset(2, n_mels=cfg_kw["n_mels"], seq_len=48, max_target_length=12, device=dev, rank=rank, world=2,
                              seed=21, num_samples=6)
    feats, labels = TWH.O.create_dummy_pool(seed=21, n_mels=cfg_kw["n_mels"], seq_len=48, max_target_length=12, num_samples=6)
    ds.features, ds.labels = torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)  # the oracle's pool
    it = iter(ds)
    losses, sizes = [], []
    for _ in range(4):
        f, l = next(it)
        sizes.append(int(f.shape[0]))
        losses.append(float(train.distributed_train_step(strat, model, (f, l), opt).item()))
    torch.cuda.synchronize()
    q.put((rank, model.arena.p.cpu().numpy(), losses, sizes))
    torch.distributed.destroy_process_group()


def test_two_rank_whisper_step_matches_oracle(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_whisper_oracle_worker, args=(r, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, p0, l0, s0), (_, p1, l1, s1) = res
    assert np.array_equal(p0, p1) and l0 == l1
    assert s0 == [2, 2, 2, 2] and s1 == [2, 0, 2, 0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_whisper_step_gpu as TWH
    cfg_kw = TWH.small_cfg()
    _, ocfg, params = TWH.build("fp32", cfg_kw, dev)
    feats, labels = TWH.O.create_dummy_pool(seed=21, n_mels=cfg_kw["n_mels"], seq_len=48, max_target_length=12, num_samples=6)
    ref, _ = TWH.O.train_steps(ocfg, params, feats, labels, 2, 4, lr=1e-3, n_replicas=2)
    assert max(abs(x - y) for x, y in zip(l0, ref)) <= 2e-4, (l0, ref)



def _early_worker(rank, port, q, early, grad_dtype, exchange):
    """ADVICE r4 (optim.py:126): the early Adam slices WITH replicas - the LM-head and embedding buckets updated on the
    optimizer stream as soon as their own collective is done (optim.Adam.begin_early_buckets) - on two ranks with
    DIFFERENT shards, against the same job with the early slices off (one update at the end of the step)."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import tethys_speech_amd  # noqa: F401
        from tethys_speech_amd import dist as D, optim, train, whisper
        torch.cuda.set_device(0)
        dev = "cuda:0"
        train.ADAM_EARLY = bool(early)
        strat = D.DataParallelStrategy(rank, 2, backend="gloo", bucket_bytes=64 * 1024, grad_dtype=grad_dtype, exchange=exchange)
        model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=11, **KW)
        strat.broadcast_parameters(model.arena.p)
        model.refresh_shadows()
        opt = optim.Adam(1e-3)
        losses, ran = [], []
        for f, l in _batches()[rank]:
            out = train.distributed_train_step(strat, model, (torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)), opt,
                                               pipelined=True)
            losses.append(float(out.item()))
            ran.append(int(getattr(opt, "early_buckets_ran", 0)))
        model.finish_late()
        torch.cuda.synchronize()
        q.put((rank, "ok", model.arena.p.cpu().numpy(), losses, ran))
        torch.distributed.destroy_process_group()
    except BaseException as e:
        import traceback
        q.put((rank, "error", f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None, None))


@pytest.mark.parametrize("grad_dtype,exchange", [("fp32", "allreduce"), ("bf16", "mesh")])
def test_two_rank_early_adam_buckets_leave_the_same_model(dev, grad_dtype, exchange):
    ctx = mp.get_context("spawn")

    def job(early):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_early_worker, args=(r, port, q, early, grad_dtype, exchange)) for r in range(2)]
        for p_ in procs:
            p_.start()
        res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
        for p_ in procs:
            p_.join(60)
        for r in res:
            assert r[1] == "ok", r[2]
        assert np.array_equal(res[0][2], res[1][2]), "replicas diverged"
        return res[0]

    _, _, p_on, l_on, ran_on = job(True)
    _, _, p_off, l_off, ran_off = job(False)
    assert all(n == 2 for n in ran_on), ("the early slices (LM head, embedding table) did not both run every step", ran_on)
    assert all(n == 0 for n in ran_off), ran_off
    # same per-parameter arithmetic either way (Adam is elementwise): equal up to the step's own fp32-atomic noise
    assert np.allclose(l_on, l_off, rtol=1e-5, atol=1e-5), (l_on, l_off)
    dp = np.abs(p_on - p_off)
    assert float((dp > 1e-5).mean()) <= 2e-3 and float(np.median(dp)) <= 1e-7, (float(dp.max()), float((dp > 1e-5).mean()))
