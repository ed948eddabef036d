"""Memory discipline of the step's workspace: every buffer gets a guard zone behind it (TMI_WS_GUARD) and the
torch.empty ones start as NaN (TMI_WS_POISON), then real training steps run - batch shapes changing, dropout on and
off, with and without the clipping Adam.  A kernel that writes past the end of a workspace buffer (the r2 bug:
tmi_fir_gn_workspace_floats sized for fewer chunks than the backward pass wrote) or reads one before writing it fails
here instead of corrupting whatever the allocator placed next to it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wav2vec2_oracle as V  # noqa: E402  (checker only: pools and index draws)
from oracle import whisper_oracle as O  # noqa: E402


@pytest.fixture
def guarded(monkeypatch):
    monkeypatch.setenv("TMI_WS_GUARD", "4096")
    monkeypatch.setenv("TMI_WS_POISON", "1")


def _finite(model, loss):
    assert np.isfinite(float(loss.item()))
    bad = [k for k, v in model.arena.ref_views(model.arena.g).items() if not bool(torch.isfinite(v).all())]
    assert not bad, bad[:8]


@pytest.mark.parametrize("precision,dropout", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_wav2vec2_steps_stay_inside_their_workspace(dev, guarded, precision, dropout):
    import test_wav2vec2_gpu as TW
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train
    model, ocfg, _ = TW.build(precision, dev)
    if dropout:
        model.enable_dropout(0.1, 0.1, seed=77, act_p=0.1)
    strat = D.DataParallelStrategy(0, 1)
    opt = optim.Adam(learning_rate=1e-4, epsilon=1e-8)
    rng = np.random.default_rng(0)
    # (400, 2) -> (400, 1) are the shapes of tests/test_wav2vec2_gpu.py::test_whisper_single_step_curve_fp32, the test that
    # aborted in round 2 (gpurun_out/ab14_tests.log): conv0 output 80 frames = 1 FIR chunk but 2 GroupNorm chunks per sample,
    # so the backward apply pass wrote B*2*12*C floats into a B*1*12*C buffer (fixed in 8db8fe5, DESIGN (f))
    for T_in, B in ((400, 2), (400, 1), (1600, 3), (400, 2)):  # (T = 20 and 80 frames: one and two chunks per sample)
        pool = V.create_dummy_pool(seed=B, num_samples=B, length=T_in)
        T = V.feature_lengths(ocfg, T_in)[-1]
        model.neg_per_time = False
        neg = V.sample_negative_indices(rng, B, T, ocfg.num_negatives)
        loss = train.wav2vec2_train_step(strat, model, torch.from_numpy(pool).to(dev), torch.from_numpy(neg).to(dev), opt)
        assert np.isfinite(float(loss.item()))
        assert model.check_workspace_guards() == []
        neg_t = V.sample_negative_indices_roll(rng, T, ocfg.num_negatives)
        loss = train.single_train_step(model, torch.from_numpy(pool).to(dev), torch.from_numpy(neg_t).to(dev), opt)
        assert np.isfinite(float(loss.item()))
        assert model.check_workspace_guards() == []
        # stable_jobs/wav2vec2_dist.py's step (T:1143-1190): the same kernels under the strategy
        loss = train.stable_wav2vec2_train_step(strat, model, torch.from_numpy(pool).to(dev), torch.from_numpy(neg_t).to(dev), opt)
        assert np.isfinite(float(loss.item()))
        assert model.check_workspace_guards() == []
    model.neg_per_time = False
    loss = model.forward_backward(torch.from_numpy(pool).to(dev), torch.from_numpy(neg).to(dev))
    _finite(model, loss)
    assert model.check_workspace_guards() == []


@pytest.mark.parametrize("precision,dropout", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_whisper_steps_stay_inside_their_workspace(dev, guarded, precision, dropout):
    import test_whisper_step_gpu as TWH
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train
    cfg_kw = TWH.small_cfg()
    model, ocfg, _ = TWH.build(precision, cfg_kw, dev)
    if dropout:
        model.enable_dropout(0.1, 0.1, seed=5)
    strat = D.DataParallelStrategy(0, 1)
    opt = optim.Adam(1e-4)
    for T_in, S, B in ((48, 12, 3), (47, 9, 1), (64, 12, 2), (48, 12, 3)):
        feats, labels = O.create_dummy_pool(seed=B, n_mels=cfg_kw["n_mels"], seq_len=T_in, max_target_length=S, num_samples=B)
        loss = train.distributed_train_step(strat, model, (torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)), opt)
        assert np.isfinite(float(loss.item()))
        assert model.check_workspace_guards() == []
    loss = model.forward_backward(torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev))
    _finite(model, loss)
    assert model.check_workspace_guards() == []
