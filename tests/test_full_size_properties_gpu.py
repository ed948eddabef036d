"""Size-independent properties at BASELINE.json's full headline size (Whisper small-ref, per-GPU batch 8, 30 s clips,
bf16 path), where the oracle is too slow to be the checker: init-time loss, exactly-dead gradients, and the
data-parallel identity  grad(batch of 8) = (grad(first 4) + grad(last 4)) / 2  (W:829-836 sums replica gradients of
per-replica means).  Dropout off: these are parity-mode properties."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_small_ref_full_size_properties(dev):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper
    from tethys_speech_amd.data import create_dummy_dataset
    model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
    feats, labels = next(iter(create_dummy_dataset(8, device=dev, seed=1234, drop_remainder=True)))
    assert tuple(feats.shape) == (8, 80, 3000) and tuple(labels.shape) == (8, 100)

    loss8 = float(model.forward_backward(feats, labels).item())
    g8 = model.arena.g.clone()
    views = model.arena.ref_views(g8)
    # at Keras-default init the loss is ln(vocab) + O(0.02)
    assert abs(loss8 - math.log(51865)) < 0.1, loss8
    # softmax is shift-invariant per row: every k_proj bias has an exactly-zero true gradient
    scale = max(float(v.abs().max()) for k, v in views.items() if k.endswith("q_proj.bias"))
    worst_k = max(float(v.abs().max()) for k, v in views.items() if k.endswith("k_proj.bias"))
    assert worst_k <= 2e-2 * scale, (worst_k, scale)
    # the zero pad columns of the stored LM head never receive a gradient (they must stay zero under Adam)
    full = model.arena.view(g8, "lm_head.kernel")
    assert float(full[:, 51865:].abs().max()) == 0.0
    assert all(bool(torch.isfinite(v).all()) for v in views.values())

    la = float(model.forward_backward(feats[:4].contiguous(), labels[:4].contiguous()).item())
    ga = model.arena.g.clone()
    lb = float(model.forward_backward(feats[4:].contiguous(), labels[4:].contiguous()).item())
    gb = model.arena.g.clone()
    assert abs(0.5 * (la + lb) - loss8) <= 1e-4 * abs(loss8)   # measured 4e-7 .. 6e-6
    half = 0.5 * (ga + gb)
    rel = float((half - g8).norm() / g8.norm())
    assert rel <= 5e-3, rel                                     # measured 8e-4 (bf16 rounding of different batch shapes)


def test_wav2vec2_base_full_size_properties(dev):
    """BASELINE configs[3] size (Wav2Vec2-base, per-GPU batch 8, 2 s clips), rates 0: finite loss assembled as
    contrastive - 0.1 * perplexity, the quantiser projection and every parameter reached only through the diversity
    term get an exactly-zero gradient (V:631-638), and the replica scaling of V:1231: the gradients
    with num_replicas = 2 are half of those with 1."""
    import numpy as np
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import wav2vec2
    from tethys_speech_amd.data import W2VDummyDataset
    model = wav2vec2.create_full_model("pretraining", "base", device=dev, precision="bf16", seed=1234)
    assert model.arena.n_params == 92_297_728
    audio = next(iter(W2VDummyDataset(8, device=dev, seed=1234)))
    assert tuple(audio.shape) == (8, 32000)
    neg = torch.from_numpy(wav2vec2.sample_negative_indices(np.random.default_rng(3), 8, 100, 100)).to(dev)
    l1 = float(model.forward_backward(audio, neg, num_replicas=1).item())
    g1 = model.arena.g.clone()
    closs, perp = float(model.ws["closs"].item()), float(model.ws["perplexity"].item())
    assert np.isfinite(l1) and abs(l1 - (closs - 0.1 * perp)) <= 1e-3 * abs(l1)
    assert 1.0 <= perp <= 320.0 and closs > 0.0
    views = model.arena.ref_views(g1)
    assert float(views["quantizer.projection.kernel"].abs().max()) == 0.0
    assert float(views["quantizer.projection.bias"].abs().max()) == 0.0
    assert all(bool(torch.isfinite(v).all()) for v in views.values())
    l2 = float(model.forward_backward(audio, neg, num_replicas=2).item())
    g2 = model.arena.g.clone()
    assert abs(2.0 * l2 - l1) <= 2e-3 * abs(l1)
    assert float((2.0 * g2 - g1).norm() / g1.norm()) <= 2e-2
