"""Whole-step parity: the HIP forward+backward(+Adam) against the oracle on the same
parameters and the same synthetic batch (dropout 0).

Tolerances: fp32 path vs fp64 oracle — loss |d| <= 1e-5, every gradient
tensor max|err| <= 1e-4 * max|ref|; bf16 path — loss |d| <= 2e-2, gradients compared by
relative L2 error <= 6e-2 (bf16 compute is never called "reference parity")."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import whisper_oracle as O  # noqa: E402  (checker only)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def small_cfg(**kw):
    base = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
                encoder_layers=2, decoder_layers=2, n_mels=16, n_ctx=32, decoder_start_token_id=150,
                max_target_positions=32)
    base.update(kw)
    return base


def build(precision, cfg_kw, dev, seed=7):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper
    ocfg = O.make_config("small", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, **cfg_kw)
    params = O.init_params(ocfg, seed=seed, dtype=torch.float64)
    # non-trivial biases / LayerNorm affine so their gradients and use are exercised
    g = torch.Generator().manual_seed(seed)
    for k, v in params.items():
        if k.endswith(".bias") or k.endswith(".beta"):
            v.copy_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
        if k.endswith(".gamma"):
            v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
    model = whisper.create_whisper_model("small", device=dev, precision=precision, **cfg_kw)
    model.arena.load_ref(params)
    model.refresh_shadows()
    return model, ocfg, params


@pytest.mark.parametrize("T_in", [48, 47])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_step_gradients_match_oracle(dev, precision, T_in):
    cfg_kw = small_cfg()
    model, ocfg, params = build(precision, cfg_kw, dev)
    S, B = 12, 3
    feats, labels = O.create_dummy_pool(seed=11, n_mels=cfg_kw["n_mels"], seq_len=T_in, max_target_length=S, num_samples=B)
    if precision == "bf16":  # evaluate the oracle on the bf16-rounded master weights the kernels see
        for k in params:
            if k.endswith(".kernel"):
                params[k] = params[k].to(torch.bfloat16).double()
    loss_ref, grads_ref = O.loss_and_grads(params, torch.from_numpy(feats), torch.from_numpy(labels), ocfg)
    loss = model.forward_backward(torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev))
    torch.cuda.synchronize()
    lv = float(loss.item())
    from _margins import within
    within(f"whisper step {precision} |dloss|", abs(lv - float(loss_ref)), 1e-6 if precision == "fp32" else 2e-3)  # measured 3e-8 / 8e-4
    got = model.arena.ref_views(model.arena.g)
    worst = {}
    for k, gr in grads_ref.items():
        gg = got[k].double().cpu()
        # k_proj.bias has an exactly-zero true gradient (softmax is shift-invariant per row), so
        # errors are measured against max(|ref|, floor) with a floor far below any live gradient
        if precision == "fp32":
            err = float((gg - gr).abs().max() / max(float(gr.abs().max()), 1e-4))
        else:
            err = float((gg - gr).norm() / max(float(gr.norm()), 1e-2))
        worst[k] = err
    within(f"whisper step {precision} worst gradient (fp32: max-norm, bf16: rel L2)", max(worst.values()),
           5e-5 if precision == "fp32" else 6e-2, sorted(worst.items(), key=lambda kv: -kv[1])[:4])  # measured 2.2e-5 / 4.1e-2


def test_step_with_dropout_matches_oracle_fed_the_same_masks(dev):
    """Training-mode dropout (W:160 attention probabilities, W:205 FFN output, W:342 / W:411 stem and embedding outputs):
    the bf16 path draws counter-based masks inside its kernels; the oracle, fed the same generator restated on the
    host (oracle/dropout.py), must give the same loss and gradients — for two consecutive steps, whose masks differ."""
    from oracle import dropout as DO
    cfg_kw = small_cfg()
    model, ocfg, params = build("bf16", cfg_kw, dev)
    model.enable_dropout(0.1, 0.1, seed=0xC0FFEE)
    ocfg = O.make_config_like(ocfg, dropout=0.1, attention_dropout=0.1)
    S, B, T_in = 12, 3, 47
    feats, labels = O.create_dummy_pool(seed=13, n_mels=cfg_kw["n_mels"], seq_len=T_in, max_target_length=S, num_samples=B)
    for k in params:
        if k.endswith(".kernel"):
            params[k] = params[k].to(torch.bfloat16).double()
    losses = []
    try:
        for step in range(2):
            O.DROPOUT_PROVIDER = DO.HostDropout(0xC0FFEE, step)
            loss_ref, grads_ref = O.loss_and_grads(params, torch.from_numpy(feats), torch.from_numpy(labels), ocfg)
            loss = model.forward_backward(torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev))
            torch.cuda.synchronize()
            lv = float(loss.item())
            losses.append((lv, float(loss_ref)))
            from _margins import within
            within("whisper step bf16 + dropout |dloss|", abs(lv - float(loss_ref)), 2e-3, losses)  # measured 6e-4
            got = model.arena.ref_views(model.arena.g)
            errs = {k: float((got[k].double().cpu() - gr).norm() / max(float(gr.norm()), 1e-2)) for k, gr in grads_ref.items()}
            within("whisper step bf16 + dropout worst gradient rel L2", max(errs.values()), 6e-2,
                   sorted(errs.items(), key=lambda kv: -kv[1])[:4])
    finally:
        O.DROPOUT_PROVIDER = None
    # the two steps drew different masks (same batch, same weights: the losses differ) and dropout is really on
    assert abs(losses[0][1] - losses[1][1]) > 1e-4
    O_nodrop, _ = O.loss_and_grads(params, torch.from_numpy(feats), torch.from_numpy(labels), O.make_config_like(ocfg, dropout=0.0, attention_dropout=0.0))
    assert abs(float(O_nodrop) - losses[0][1]) > 1e-4


def test_ten_step_loss_curve_fp32_small_dims(dev):
    """10 Adam steps (lr 1e-4, TF epsilon placement), fp32 path, against the oracle run in
    fp64 on the same pool: |dloss| <= 1e-3 per step is the BASELINE target; we hold 1e-4."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import optim, dist, train
    cfg_kw = small_cfg()
    model, ocfg, params = build("fp32", cfg_kw, dev)
    S, B, T_in = 12, 2, 48
    feats, labels = O.create_dummy_pool(seed=5, n_mels=cfg_kw["n_mels"], seq_len=T_in, max_target_length=S, num_samples=7)
    ref_losses, _ = O.train_steps(ocfg, params, feats, labels, B, 10, lr=1e-3)
    opt = optim.Adam(learning_rate=1e-3)
    strat = dist.DataParallelStrategy(0, 1)
    it = O.batches(feats, labels, B)
    got = []
    for _ in range(10):
        f, l = next(it)
        loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                           torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt)
        got.append(float(loss.item()))
    assert max(abs(a - b) for a, b in zip(got, ref_losses)) <= 1e-4, (got, ref_losses)
    assert got[-1] < got[0]


def test_ragged_final_batches_bf16(dev):
    """The reference's dataset has no drop_remainder (W:812): a pass over the pool ends in a short batch.
    bf16 path, pool of 7 in batches of 3 (3, 3, 1, 3, 3, 1): the workspaces are re-laid-out on every change of
    batch size and the losses must stay on the oracle's curve (bf16 tolerance 2e-2)."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import optim, dist, train
    cfg_kw = small_cfg()
    model, ocfg, params = build("bf16", cfg_kw, dev)
    S, B, T_in = 12, 3, 47
    feats, labels = O.create_dummy_pool(seed=9, n_mels=cfg_kw["n_mels"], seq_len=T_in, max_target_length=S, num_samples=7)
    ref_losses, _ = O.train_steps(ocfg, params, feats, labels, B, 6, lr=1e-3)
    opt = optim.Adam(learning_rate=1e-3)
    strat = dist.DataParallelStrategy(0, 1)
    it = O.batches(feats, labels, B)
    got, sizes = [], []
    for _ in range(6):
        f, l = next(it)
        sizes.append(len(f))
        loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                           torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt)
        got.append(float(loss.item()))
    assert sizes == [3, 3, 1, 3, 3, 1]
    assert max(abs(a - b) for a, b in zip(got, ref_losses)) <= 2e-2, (got, ref_losses)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 1e-3)])  # measured 6.5e-7 / 3.9e-4
def test_whisper_tiny_loss_curve_golden(dev, precision, tol):
    """BASELINE config #1(b): Whisper-tiny (384/6h/1536/4+4), B=2, 10 steps, 30 s clips,
    Adam 1e-4, against the committed fp64-oracle loss curve (tests/golden/make_golden.py)."""
    path = os.path.join(GOLD, "whisper_tiny_b2_10steps.json")
    if not os.path.exists(path):
        pytest.skip("golden curve not generated")
    gold = json.load(open(path))
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper, optim, dist, train
    ocfg = O.make_config("tiny")
    params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
    model = whisper.create_whisper_model("tiny", device=dev, precision=precision)
    model.arena.load_ref(params)
    model.refresh_shadows()
    feats, labels = O.create_dummy_pool(seed=gold["seed"])
    opt = optim.Adam(learning_rate=1e-4)
    strat = dist.DataParallelStrategy(0, 1)
    it = O.batches(feats, labels, 2)
    got = []
    for _ in range(len(gold["losses"])):
        f, l = next(it)
        loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                           torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt)
        got.append(float(loss.item()))
    err = max(abs(a - b) for a, b in zip(got, gold["losses"]))
    from _margins import within
    within(f"whisper-tiny B=2 10-step golden {precision} max |dloss|", err, tol, (got, gold["losses"]))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 1e-3)])  # measured 1.2e-6 / 2.6e-4: both held to the north star's 1e-3
def test_whisper_small_ref_b8_loss_curve_golden(dev, precision, tol):
    """BASELINE.json's headline model at its headline batch (configs[1]): Whisper small-ref (768/12h/3072/4+4,
    W:13-18), per-GPU batch 8, 30 s clips, Adam 1e-4 (W:901), dropout 0, the bench.py pool (seed 1234), 10 steps
    against the committed fp64-oracle curve.  This is the north-star's "loss curve matching to 1e-3" on the model
    it names, held by BOTH paths: fp32 (the parity mode, ~1e-6 measured) and bf16 (the perf mode the bench times,
    2.6e-4 measured since tmi_linear_xent reads the loss's target logit in fp32 - 7.7e-4 before, all of the difference the
    bf16 rounding of that one logit per row: a breach of the 1e-3 contract fails here rather than hiding under a looser bound).  ``dataset.batch(8).repeat()`` keeps the remainder (W:812-815): step 6 is the
    two-sample batch."""
    path = os.path.join(GOLD, "whisper_small_ref_b8_10steps.json")
    if not os.path.exists(path):
        pytest.skip("golden curve not generated")
    gold = json.load(open(path))
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper, optim, dist, train, ops
    ocfg = O.make_config("small")
    params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
    feats, labels = O.create_dummy_pool(seed=gold["seed"])

    def curve():
        model = whisper.create_whisper_model("small", device=dev, precision=precision)
        assert model.arena.n_params == 147_781_632
        model.arena.load_ref(params)
        model.refresh_shadows()
        opt = optim.Adam(learning_rate=gold["lr"])
        strat = dist.DataParallelStrategy(0, 1)
        it = O.batches(feats, labels, gold["batch_size"])
        got, sizes = [], []
        for _ in range(len(gold["losses"])):
            f, l = next(it)
            sizes.append(len(f))
            loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                               torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt)
            got.append(float(loss.item()))
        assert sizes == [8, 8, 8, 8, 8, 8, 2, 8, 8, 8]
        return got

    # The curve is measured on the REPRODUCIBLE reductions (ops.set_deterministic: LayerNorm backward through its fixed-order
    # fold, column sums with one workgroup per column group).  With the default fp32 atomics the bias / LayerNorm gradients
    # differ by ~1e-7 relative between two runs, and ten Adam steps turn that into +-1e-4 of loss: over 30 bf16 runs the
    # maximum error ranged 7.5e-4 .. 9.7e-4 (profiles/r04_bf16_margin.txt) - a bound of 1e-3 on ONE such run is a coin with a
    # thin edge.  The reproducible form gives one number per build, and the second run below must repeat it bit for bit
    # (which also holds the forward to the round-4 attention fix: an inline-asm v_max3_f32 inside the MFMA hazard window).
    was = ops.set_deterministic(True)
    try:
        got = curve()
        again = curve() if precision == "bf16" else got
    finally:
        ops.set_deterministic(was)
    assert got == again, ("two runs of the same ten steps differ", got, again)
    err = [abs(a - b) for a, b in zip(got, gold["losses"])]
    print(f"small-ref B=8 {precision}: max |dloss| = {max(err):.2e} (bound {tol:g}); per step {['%.1e' % e for e in err]}")
    from _margins import within
    within(f"whisper small-ref B=8 10-step golden {precision} max |dloss|", max(err), tol, (err, got, gold["losses"]))
    if precision == "fp32":  # the north star's 1e-3 is the contract; what the fp32 path actually holds is ~1e-6
        assert max(err) <= 1e-5, err
    else:  # (<= 6e-4 with the contract's bound left at 1e-3; 2.6e-4 measured, reproducible)
        assert max(err) <= 6e-4, err


def test_whisper_small_ref_b8_loss_curve_golden_on_the_path_the_bench_times(dev):
    """The same ten steps as above, bf16, ONCE, on what ``bench.py`` times: the default reductions (fp32 atomics in the bias /
    LayerNorm-parameter gradients, not ``ops.set_deterministic``), the pipelined step (the decoder layers' Adam slice under
    the next step's encoder) and the launch plan (steps 3 onward of the batch-8 shape are replays, train.planned_step).
    Bound: the north star's 1e-3 (W:585-600 loss; the deterministic pair above pins the tighter 6e-4)."""
    path = os.path.join(GOLD, "whisper_small_ref_b8_10steps.json")
    if not os.path.exists(path):
        pytest.skip("golden curve not generated")
    gold = json.load(open(path))
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper, optim, dist, train
    ocfg = O.make_config("small")
    params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
    feats, labels = O.create_dummy_pool(seed=gold["seed"])
    model = whisper.create_whisper_model("small", device=dev, precision="bf16")
    model.arena.load_ref(params)
    model.refresh_shadows()
    opt = optim.Adam(learning_rate=gold["lr"])
    strat = dist.DataParallelStrategy(0, 1)
    step = train.planned_step(strat, model, opt, "whisper", pipelined=True)
    it = O.batches(feats, labels, gold["batch_size"])
    got = []
    for _ in range(len(gold["losses"])):
        f, l = next(it)
        got.append(step(torch.from_numpy(np.ascontiguousarray(f)).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)))
    model.finish_late()
    got = [float(x.item()) for x in got]
    if train.plan_ok(strat, model):
        assert step.planned is not None and step.planned.replays >= 5, "the launch plan never replayed"
    err = [abs(a - b) for a, b in zip(got, gold["losses"])]
    print(f"small-ref B=8 bf16, default reductions + pipelined + launch plan: max |dloss| = {max(err):.2e}")
    from _margins import within
    within("whisper small-ref B=8 10-step golden bf16 (bench path) max |dloss|", max(err), 1e-3, (err, got, gold["losses"]))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pipelined_steps_leave_the_same_model(dev, precision):
    """``distributed_train_step(..., pipelined=True)`` returns while the decoder layers' Adam slice is still running on the
    second stream (train.ADAM_LATE); the next step's decoder waits for it, ``finish_late`` / ``save_checkpoint`` order
    other readers behind it.  Per-parameter arithmetic is unchanged, so N pipelined steps must leave the model N plain
    steps leave - up to the run-to-run noise of the step's fp32 atomics, measured by running the plain steps twice.  A
    missing wait shows as a decoder that trained on stale weights: losses apart by far more than that noise."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, whisper
    cfg_kw = small_cfg()
    rng = np.random.default_rng(5)
    batches = [(rng.standard_normal((3, 16, 48)).astype(np.float32), rng.integers(0, 150, (3, 12)).astype(np.int32))
               for _ in range(6)]

    def run(pipelined):
        model = whisper.create_whisper_model("small", device=dev, precision=precision, seed=3, **cfg_kw)
        assert model._side is not None and model.late_adam_range() is not None
        opt = optim.Adam(1e-3)
        strat = D.DataParallelStrategy(0, 1)
        losses = []
        for f, l in batches:
            losses.append(train.distributed_train_step(strat, model, (torch.from_numpy(f).to(dev), torch.from_numpy(l).to(dev)), opt,
                                                       pipelined=pipelined))
        if pipelined:
            assert model._late_ev is not None, "the late slice never ran: nothing was tested"
            model.finish_late()
        # (a reader on the default stream after finish_late: exactly what a caller would do)
        p, m = model.arena.p.cpu().numpy(), model.arena.m.cpu().numpy()
        return [float(x.item()) for x in losses], p, m

    assert train.ADAM_LATE and train.ADAM_EARLY
    l0, p0, m0 = run(False)
    l1, p1, m1 = run(False)
    l2, p2, m2 = run(True)
    noise = float(np.abs(p1 - p0).max())
    assert np.allclose(l2, l0, rtol=2e-6 if precision == "fp32" else 2e-5, atol=1e-6), (l2, l0)
    dp = np.abs(p2 - p0)
    # isolated elements whose gradient is pure summation noise may move by an Adam step; regions may not
    assert float((dp > 1e-5).mean()) <= 2e-3 and float(np.median(dp)) <= 1e-7, (float(dp.max()), noise)
    assert float(np.abs(m2 - m0).max()) <= 1e-3 * float(np.abs(m0).max())
    print(f"{precision}: plain twice max |dp| {noise:.1e}; pipelined vs plain {float(dp.max()):.1e}")
