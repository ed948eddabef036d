"""The gradient exchange on the REAL backend (nccl = RCCL) on a one-GPU box: a process group of ONE rank, with the
strategy's world-1 short-circuits disabled (``DataParallelStrategy(force_collectives=True)``).  A sum over one replica is
the identity, so every exchange form must leave the step what the plain step computes (to the run-to-run noise of the
step's own fp32 atomics, measured here by running the plain step twice) - while the
machinery that only RCCL exercises runs end to end: asynchronous ``Work`` objects (gloo runs them synchronously,
``_serial``), the dedicated exchange stream ordered after BOTH producer streams (compute + weight-gradient), staging
buffers reused across steps, ``all_to_all_single`` followed by the local fold (mesh), ``Work.wait()`` on the exchange
and optimizer streams, the post lambdas, the on_bucket path (Adam under backward), C2 / C4 and the dmabuf IPC
environment (``HSA_ENABLE_IPC_MODE_LEGACY=0``).  Reference: W:834 (implicit all-reduce inside apply_gradients), W:848,
W:1047, V:1468-1475 (NCCL forced).  What a one-rank group cannot show is link behaviour; that needs the 8-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# two encoder + two decoder layers: the early-decoder block, the per-layer hand-offs and several buckets are all in play
KW = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
          encoder_layers=2, decoder_layers=2, n_mels=16, n_ctx=32, decoder_start_token_id=150, max_target_positions=32)
STEPS = 3
FORMS = [(ex, dt) for ex in ("allreduce", "rs_ag", "mesh") for dt in ("fp32", "bf16")]


def _batches(steps=STEPS):
    rng = np.random.default_rng(21)
    return [(rng.standard_normal((2, 16, 48)).astype(np.float32), rng.integers(0, 150, (2, 12)).astype(np.int32))
            for _ in range(steps)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import tethys_speech_amd  # noqa: F401
        from tethys_speech_amd import dist as D, optim, train, whisper
        torch.cuda.set_device(0)
        dev = "cuda:0"

        ran = {}

        def run(strat, precision, adam_under_backward=False, pipelined=False, tag=None):
            model = whisper.create_whisper_model("small", device=dev, precision=precision, seed=11, **KW)
            assert model._side is not None, "the overlapped step (weight-gradient stream) is what this test is about"
            strat.broadcast_parameters(model.arena.p)
            model.refresh_shadows()
            opt = optim.Adam(1e-3)
            keep, train.ADAM_UNDER_BACKWARD = train.ADAM_UNDER_BACKWARD, adam_under_backward
            try:
                losses = []
                for f, l in _batches():
                    losses.append(train.distributed_train_step(strat, model, (torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)),
                                                               opt, pipelined=pipelined))
                    if tag is not None:
                        ran.setdefault(tag, []).append((getattr(opt, "early_buckets_ran", 0), bool(model._late_ev)))
                losses = [float(x.item()) for x in losses]
            finally:
                train.ADAM_UNDER_BACKWARD = keep
            model.finish_late()
            torch.cuda.synchronize()
            return losses, model.arena.p.cpu().numpy(), model.arena.m.cpu().numpy()

        out = {}
        for precision in ("fp32", "bf16"):
            out[("plain", precision)] = run(D.DataParallelStrategy(0, 1), precision)
            out[("plain2", precision)] = run(D.DataParallelStrategy(0, 1), precision)  # the step's own run-to-run noise
        first = True
        for ex, dt in FORMS:
            # 256 KiB buckets: ~10 buckets per step on this model, launched from inside backward
            strat = D.DataParallelStrategy(0, 1, backend="nccl", bucket_bytes=256 * 1024, exchange=ex, grad_dtype=dt,
                                           force_collectives=True)
            if first:
                assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
                ones = torch.ones(8, device=dev)
                torch.distributed.all_reduce(ones)
                assert torch.distributed.get_world_size() == 1 and float(ones.sum().item()) == 8.0
                assert not strat._serial, "RCCL works must stay asynchronous"
                first = False
            for precision in ("fp32", "bf16"):
                out[(ex, dt, precision)] = run(strat, precision)
        # Adam slice by slice under backward (on_bucket): the optimizer stream waits for each bucket's works
        strat = D.DataParallelStrategy(0, 1, backend="nccl", bucket_bytes=256 * 1024, force_collectives=True)
        out[("under_backward", "fp32")] = run(strat, "fp32", adam_under_backward=True)
        # the early Adam slices WITH replicas (optim.Adam.begin_early_buckets: the LM-head and embedding
        # buckets updated on the optimizer stream as soon as their own collective is done) and the late slice behind the
        # exchange, in the pipelined step the training loops and the bench run
        for precision in ("fp32", "bf16"):
            strat = D.DataParallelStrategy(0, 1, backend="nccl", bucket_bytes=64 * 1024, force_collectives=True)
            out[("early_buckets", precision)] = run(strat, precision, pipelined=True, tag=("early_buckets", precision))
        # the bench's "same ranks without the exchange" run must keep that schedule
        strat = D.DataParallelStrategy(0, 1, backend="nccl", bucket_bytes=64 * 1024, force_collectives=True)
        strat.exchange_off = True
        out[("early_buckets_off", "fp32")] = run(strat, "fp32", pipelined=True, tag=("early_buckets_off", "fp32"))
        # Launch plans with replicas (plan.host_call): the collectives and their Work.wait()s are callback nodes of the
        # recorded step.  Six pipelined steps - two eager, one recorded, three replayed - against the same six issued
        # from Python, per exchange form (RCCL Work objects refilled by every replay, waits on the exchange and optimizer
        # streams, the early-bucket Adam slices, the staging fills / copies of rs_ag and mesh as library launches).
        def run6(strat, precision, planned):
            model = whisper.create_whisper_model("small", device=dev, precision=precision, seed=11, **KW)
            strat.broadcast_parameters(model.arena.p)
            model.refresh_shadows()
            opt = optim.Adam(1e-3)
            if planned:
                step = train.planned_step(strat, model, opt, "whisper", pipelined=True)
                assert step.planned is not None, "train.plan_ok refused a job with replicas"
            else:
                step = lambda f, l: train.distributed_train_step(strat, model, (f, l), opt, pipelined=True)
            losses = [step(torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)) for f, l in _batches(6)]
            losses = [float(x.item()) for x in losses]
            model.finish_late()
            torch.cuda.synchronize()
            info = None
            if planned:
                pl = [v["plan"] for v in step.planned._by_sig.values() if v.get("plan") is not None]
                info = (step.planned.replays, [p_.callbacks for p_ in pl], [p_.launches for p_ in pl])
            return (losses, model.arena.p.cpu().numpy(), model.arena.m.cpu().numpy()), info

        for ex, dt, precision in (("allreduce", "fp32", "fp32"), ("allreduce", "fp32", "bf16"), ("rs_ag", "fp32", "fp32"),
                                  ("mesh", "bf16", "fp32")):
            for planned in (False, True):
                strat = D.DataParallelStrategy(0, 1, backend="nccl", bucket_bytes=64 * 1024, exchange=ex, grad_dtype=dt,
                                               force_collectives=True)
                out[("six", ex, dt, precision, planned)] = run6(strat, precision, planned)
        # the mesh form's all-to-all must not stall the ISSUING THREAD.  Its ``w.wait()`` (dist.py) is a
        # stream-order dependency under RCCL (the exchange stream waits for the collective before the local fold; the host
        # returns at once) - shown here by queueing ~50 ms of device work in front of the exchange: the issue path must be
        # back long before that work, let alone the collective behind it, has finished.
        import time
        strat = D.DataParallelStrategy(0, 1, backend="nccl", exchange="mesh", grad_dtype="fp32", force_collectives=True)
        g = torch.randn(1 << 20, device=dev)
        g0 = g.clone()
        strat._exchange(g, 0, g.numel())          # (staging buffers, exchange stream: first use)
        torch.cuda.synchronize()
        torch.cuda._sleep(100_000_000)            # ~50 ms on the compute stream; the exchange stream is ordered behind it
        t0 = time.perf_counter()
        strat.begin_gradients(g)
        strat._exchange(g, 0, g.numel())
        t_issue = time.perf_counter() - t0
        done = torch.cuda.Event()
        done.record(strat._xs)
        still_running = not done.query()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        for w in strat._works:
            w.wait()
        for fn in strat._post:
            fn()
        torch.cuda.synchronize()
        out["mesh_issue"] = (t_issue, t_all, still_running, bool(torch.equal(g, g0)))
        out["ran"] = ran
        torch.distributed.destroy_process_group()
        q.put(("ok", out))
    except BaseException as e:  # report instead of hanging the parent on q.get
        import traceback
        q.put(("error", f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


def test_rccl_one_rank_exchange_is_the_identity(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=600)
    p.join(60)
    assert status == "ok", out
    def close(a, b, what):
        """Same trajectory up to the step's own run-to-run noise: the weight-gradient / bias-sum kernels add their split-K
        and per-block partials with fp32 atomics, whose order varies (|dg| ~ 1e-7 |g|), so two PLAIN runs already differ in
        the last bits.  A missed wait or a stale staging buffer would lose a whole bucket: an Adam step of lr = 1e-3 on
        every element of it, three orders of magnitude above this bound."""
        (la, pa, ma), (lb, pb, mb) = a, b
        assert np.allclose(la, lb, rtol=1e-6, atol=1e-6), (what, la, lb)
        dp, dm = np.abs(pa - pb), np.abs(ma - mb)
        # (a gradient that is zero in exact arithmetic - k_proj.bias - is pure summation noise, and Adam turns noise of
        # the size of its epsilon into a visible step: allow isolated elements, not regions)
        frac = float((dp > 1e-5).mean())
        assert frac <= 2e-3 and float(np.median(dp)) <= 1e-7 and float(dm.max()) <= 1e-3 * float(np.abs(mb).max()), \
            (what, frac, float(dp.max()), float(dm.max()))
        return float(dp.max())

    t_issue, t_all, still_running, same = out["mesh_issue"]
    print(f"mesh exchange behind ~50 ms of queued work: issue path {t_issue * 1e3:.2f} ms, everything drained after {t_all * 1e3:.1f} ms")
    assert still_running and t_issue < 0.25 * t_all, ("the mesh issue path waited for the device", t_issue, t_all, still_running)
    assert same, "a one-rank mesh exchange must be the identity"
    for precision in ("fp32", "bf16"):
        plain = out[("plain", precision)]
        noise = close(out[("plain2", precision)], plain, ("plain twice", precision))
        print(f"{precision}: two plain runs differ by max |dp| = {noise:.1e}")
        l0, p0, m0 = plain
        for ex, dt in FORMS:
            l1, p1, m1 = out[(ex, dt, precision)]
            if dt == "fp32":
                # fp32 wire: every form moves the bytes unchanged -> the plain step's trajectory
                d = close(out[(ex, dt, precision)], plain, (ex, dt, precision))
                print(f"{precision} {ex}/fp32 wire: max |dp| vs the plain step = {d:.1e}")
            else:
                # bf16 wire: gradients are rounded to bf16 once on the way (|dg| <= 2^-8 |g|); Adam's first moment after
                # 3 steps moves by at most that fraction
                assert np.allclose(l1, l0, rtol=2e-3, atol=2e-3), (ex, dt, precision, l1, l0)
                den = np.abs(m0).max()
                assert np.abs(m1 - m0).max() <= 1e-2 * den, (ex, dt, precision, np.abs(m1 - m0).max(), den)
    for ex, dt, precision in (("allreduce", "fp32", "fp32"), ("allreduce", "fp32", "bf16"), ("rs_ag", "fp32", "fp32"),
                              ("mesh", "bf16", "fp32")):
        eager, _ = out[("six", ex, dt, precision, False)]
        planned, (replays, callbacks, launches) = out[("six", ex, dt, precision, True)]
        print(f"launch plan with replicas, {ex}/{dt} {precision}: {replays} replays, {callbacks} callback nodes, {launches} launches")
        assert replays == 3 and len(callbacks) == 1 and callbacks[0] >= 2, (ex, dt, precision, replays, callbacks)
        close(planned, eager, ("planned vs eager", ex, dt, precision))
    close(out[("under_backward", "fp32")], out[("plain", "fp32")], "Adam under backward")
    for key in (("early_buckets", "fp32"), ("early_buckets", "bf16"), ("early_buckets_off", "fp32")):
        close(out[key], out[("plain", key[1])], key)
        per_step = out["ran"][key]
        # both slices (LM head, embedding table) ran early in every step, and the late slice was left running behind the step
        assert all(n == 2 for n, _ in per_step) and all(late for _, late in per_step), (key, per_step)
