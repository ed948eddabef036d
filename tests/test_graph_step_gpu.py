"""Captured-graph training step (opt-in, TMI_HIP_GRAPH=1): replays must follow the eager trajectory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_step_matches_eager(dev, monkeypatch):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, optim, train, whisper
    kw = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=32, decoder_start_token_id=150,
              max_target_positions=32)
    rng = np.random.default_rng(1)
    batches = [(torch.from_numpy(rng.standard_normal((2, 16, 48)).astype(np.float32)).to(dev),
                torch.from_numpy(rng.integers(0, 150, (2, 12)).astype(np.int32)).to(dev)) for _ in range(6)]
    strat = dist.DataParallelStrategy(0, 1)

    def run(graph):
        monkeypatch.setenv("TMI_HIP_GRAPH", "1" if graph else "0")
        model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=5, **kw)
        opt = optim.Adam(1e-3)
        step = train.make_train_step(strat, model, opt, batches[0], warmup=2)
        assert isinstance(step, train.GraphedTrainStep) == graph
        out = [float(step(b).item()) for b in batches[1:]]
        return out, opt.iterations

    eager, it_e = run(False)
    graphed, it_g = run(True)
    assert it_e == it_g == 7
    assert np.allclose(graphed, eager, rtol=2e-5, atol=1e-6), (graphed, eager)


def test_graph_replay_survives_a_ragged_batch_in_between(dev, monkeypatch):
    """ADVICE r1 (train.py:73): a batch of another shape falls back to the eager step, which lays out a second
    workspace set; the captured set must stay alive and intact, so the next replay continues the eager trajectory."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, optim, train, whisper
    kw = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=32, decoder_start_token_id=150,
              max_target_positions=32)
    rng = np.random.default_rng(2)

    def mk(B):
        return (torch.from_numpy(rng.standard_normal((B, 16, 48)).astype(np.float32)).to(dev),
                torch.from_numpy(rng.integers(0, 150, (B, 12)).astype(np.int32)).to(dev))
    batches = [mk(2), mk(2), mk(1), mk(2), mk(1), mk(2)]
    strat = dist.DataParallelStrategy(0, 1)

    def run(graph):
        monkeypatch.setenv("TMI_HIP_GRAPH", "1" if graph else "0")
        model = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=5, **kw)
        opt = optim.Adam(1e-3)
        step = train.make_train_step(strat, model, opt, batches[0], warmup=1)
        assert isinstance(step, train.GraphedTrainStep) == graph
        return [float(step(b).item()) for b in batches[1:]]

    eager, graphed = run(False), run(True)
    assert np.allclose(graphed, eager, rtol=2e-5, atol=1e-6), (graphed, eager)
