"""Launch plans (include/tethys_mi.h tmi_plan_*, tethys_speech_amd/plan.py): a step replayed from a plan must be the step
the host code issues - same losses, same parameters, bit for bit - including what changes per step (dropout masks, the
Adam step number) and what sits between the launches (events across the weight-gradient stream, the early / late Adam
slices).  The reference's counterpart is the traced @tf.function of speech_jobs/whisper_dist.py:818-819."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_TINY = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
             encoder_layers=2, decoder_layers=2, n_mels=16, n_ctx=64, decoder_start_token_id=150, max_target_positions=32)


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _whisper_run(dev, planned, dropout, steps=9):
    from tethys_speech_amd import whisper, optim, train, ops
    from tethys_speech_amd.dist import DataParallelStrategy
    from tethys_speech_amd.data import create_dummy_dataset
    was = ops.set_deterministic(True)
    try:
        strategy = DataParallelStrategy(0, 1, init=False)
        model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=5, **_TINY)
        model.refresh_shadows()
        if dropout:
            model.enable_dropout(0.1, 0.1, seed=77)
        opt = optim.Adam(1e-3)
        # batches of 3 out of a pool of 8: the short batch of 2 comes round every third step (its own plan / eager steps)
        it = iter(create_dummy_dataset(3, n_mels=16, seq_len=96, max_target_length=12, device=dev, seed=9, num_samples=8))
        old = train.USE_PLAN
        train.USE_PLAN = planned
        try:
            step = train.planned_step(strategy, model, opt, "whisper", pipelined=True)
            losses = []
            for _ in range(steps):
                losses.append(step(*next(it)))
            model.finish_late()
            torch.cuda.synchronize()
            info = step.planned
        finally:
            train.USE_PLAN = old
        return [float(x.item()) for x in losses], model.arena.p.clone(), model.arena.m.clone(), info
    finally:
        ops.set_deterministic(was)


@pytest.mark.parametrize("dropout", [False, True])
def test_whisper_planned_steps_equal_eager_steps_bit_for_bit(dev, dropout):
    le, pe, me, _ = _whisper_run(dev, False, dropout)
    lp, pp, mp, info = _whisper_run(dev, True, dropout)
    assert info is not None and info.replays >= 3, "the plan path did not replay"
    plans = [v["plan"] for v in info._by_sig.values() if v.get("plan") is not None]
    assert plans and all(p.launches > 50 for p in plans)
    assert le == lp, (le, lp)
    assert torch.equal(pe, pp) and torch.equal(me, mp)
    if dropout:  # masks really change from step to step under replay: the same batch comes round with another loss
        assert len(set(lp)) == len(lp)


def test_wav2vec2_planned_steps_equal_eager_steps_bit_for_bit(dev):
    from tethys_speech_amd import wav2vec2, optim, train, ops
    from tethys_speech_amd.dist import DataParallelStrategy

    def run(planned):
        strategy = DataParallelStrategy(0, 1, init=False)
        over = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    conv_dim=(64, 64, 64), conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2), num_conv_pos_embeddings=8,
                    num_conv_pos_embedding_groups=4, num_codevectors_per_group=16, codevector_dim=32,
                    proj_codevector_dim=64, num_negatives=10)  # (the small model of tests/test_wav2vec2_gpu.py)
        model = wav2vec2.create_full_model("pretraining", "base", device=dev, precision="bf16", seed=3, **over)
        model.refresh_shadows()
        c = model.config
        model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=11, act_p=c.activation_dropout)
        opt = optim.Adam(3e-4, epsilon=1e-8)
        g = torch.Generator(device="cpu").manual_seed(4)
        audio = [torch.randn(3, 400, generator=g).to(dev) for _ in range(3)]
        model._prepare(3, 400)
        rng = np.random.default_rng(8)
        negs = [torch.from_numpy(wav2vec2.sample_negative_indices(rng, 3, model.T, c.num_negatives)).to(dev) for _ in range(3)]
        old = train.USE_PLAN
        train.USE_PLAN = planned
        try:
            step = train.planned_step(strategy, model, opt, "wav2vec2", pipelined=True)
            losses = [step(audio[i % 3], negs[i % 3]) for i in range(8)]
            model.finish_late()
            torch.cuda.synchronize()
            return [float(x.item()) for x in losses], model.arena.p.clone(), step.planned
        finally:
            train.USE_PLAN = old

    was = ops.set_deterministic(True)
    try:
        le, pe, _ = run(False)
        le2, pe2, _ = run(False)
        lp, pp, info = run(True)
    finally:
        ops.set_deterministic(was)
    assert info is not None and info.replays >= 4
    # The Wav2Vec2 step keeps fp32 atomics (GroupNorm / codebook gradients: tmi_set_deterministic covers the Whisper step
    # only), so two EAGER runs already differ in the last bits; the planned run must sit inside that spread, and a wrong
    # mask or Adam step number would be off by orders of magnitude more (the masks of step k decide its loss)
    spread = float((pe - pe2).abs().max())
    tol = max(4.0 * spread, 1e-7)
    assert float((pe - pp).abs().max()) <= tol, (float((pe - pp).abs().max()), spread)
    assert max(abs(a - b) for a, b in zip(le, lp)) <= max(4.0 * max(abs(a - b) for a, b in zip(le, le2)), 1e-6 * abs(le[0])), (le, lp)
