"""Log-mel front end (W:739-766): the HIP path against the numpy restatement of tf.signal's
behaviour, plus closed-form checks of the restatement itself (parity unpinned: TensorFlow is absent)."""
import numpy as np
import pytest

from oracle import logmel_oracle as L  # checker only


def test_mel_matrix_closed_form():
    m = L.linear_to_mel_weight_matrix()
    assert m.shape == (201, 80)
    assert np.all(m[0] == 0.0)                       # DC bin excluded
    assert np.all(m >= 0.0) and m.max() <= 1.0
    # every interior spectrogram bin is covered by exactly its two neighbouring triangles, which sum to 1
    # between the first and the last centre (slaney-free HTK triangles evaluated in mel)
    mel = lambda f: 1127.0 * np.log1p(f / 700.0)
    edges = np.linspace(mel(0.0), mel(8000.0), 82)
    f = np.linspace(0, 8000, 201)
    inside = (mel(f) >= edges[1]) & (mel(f) <= edges[-2])
    assert np.allclose(m[inside].sum(1), 1.0, atol=1e-12)
    assert np.all((m > 0).sum(1)[inside] <= 2)


def test_oracle_tone_and_shapes():
    sr, n = 16000, 16000
    t = np.arange(n) / sr
    x = np.sin(2 * np.pi * 1000.0 * t)
    feat = L.extract_fbank_features(x)
    assert feat.shape == (1 + (n - 400) // 160, 80) == (98, 80)
    # the 1 kHz tone lands in the mel band whose centre is nearest 1 kHz, in every frame
    mel = lambda f: 1127.0 * np.log1p(f / 700.0)
    centres = np.linspace(mel(0.0), mel(8000.0), 82)[1:-1]
    want = int(np.argmin(np.abs(centres - mel(1000.0))))
    assert np.all(np.abs(feat.argmax(1) - want) <= 1)
    # silence -> log(1e-6) everywhere
    assert np.allclose(L.extract_fbank_features(np.zeros(4000)), np.log(1e-6))
    # Parseval on one frame: sum of the two-sided power = n_fft * sum (w x)^2
    fr = x[:400] * L.hann_periodic(400)
    p = np.abs(np.fft.rfft(fr)) ** 2
    assert np.isclose(p[0] + p[-1] + 2 * p[1:-1].sum(), 400 * np.sum(fr ** 2))


def test_host_constants_match_the_restatement():
    """The product's host-side constants (built once in fp64) against the oracle's, without a GPU."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import frontend as F
    assert np.allclose(F.mel_weight_matrix(), L.linear_to_mel_weight_matrix(), rtol=0, atol=1e-15)
    assert np.allclose(F.hann_periodic(400), L.hann_periodic(400), rtol=0, atol=1e-15)
    assert F.mel_weight_matrix(40, 129, 8000, 0.0, 4000.0).shape == (129, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 16000), (3, 480000 // 10), (2, 401)])
def test_frontend_matches_oracle(dev, B, N):
    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd.frontend import LogMelFrontend
    rng = np.random.default_rng(B * 1000 + N)
    x = (rng.standard_normal((B, N)) * 0.1).astype(np.float32)
    fe = LogMelFrontend(device=dev)
    ref = L.extract_fbank_features(x.astype(np.float64))          # [B, F, 80]
    got = fe(torch.from_numpy(x).to(dev)).cpu().numpy()           # [B, 80, F]
    assert got.shape == (B, 80, fe.num_frames(N))
    # fp32 DFT of K = 400 terms: power relative error ~1e-5; compare in the linear domain
    err = np.abs(np.exp(got.transpose(0, 2, 1)) - np.exp(ref)).max() / np.exp(ref).max()
    assert err <= 1e-4, err
    assert np.abs(got.transpose(0, 2, 1) - ref).max() <= 5e-3
    got_ref_layout = fe(torch.from_numpy(x).to(dev), reference_layout=True).cpu().numpy()
    assert np.array_equal(got_ref_layout, got.transpose(0, 2, 1))


@pytest.mark.gpu
def test_frontend_feeds_the_model(dev):
    """30 s of audio -> [1, 80, 2998] features -> one training step runs (T_in need not be 3000)."""
    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper
    from tethys_speech_amd.frontend import LogMelFrontend
    fe = LogMelFrontend(device=dev)
    x = torch.randn(2, 16000 * 3, device=dev) * 0.05
    feats = fe(x)
    assert feats.shape == (2, 80, 298)
    kw = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_ctx=160, decoder_start_token_id=150, max_target_positions=32)
    model = whisper.create_whisper_model("small", device=dev, precision="bf16", **kw)
    labels = torch.randint(0, 150, (2, 12), dtype=torch.int32, device=dev)
    loss = model.forward_backward(feats.contiguous(), labels)
    assert torch.isfinite(loss).all()
