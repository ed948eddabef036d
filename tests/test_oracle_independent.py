"""Second, independent pins for the oracle's ops.

The reference ships no fixtures and TensorFlow is not installed, so the oracle stays "parity unpinned"
against a live TF.  What CAN be checked here is that each restated op agrees with an implementation of the
same published op semantics that shares no code with ``oracle/``: PyTorch's own functional ops
(``F.layer_norm``, ``F.group_norm``, ``F.gelu``, ``F.scaled_dot_product_attention``, ``F.conv1d(padding="same")``,
``torch.optim.Adam``, ``torch.nn.utils.clip_grad_norm_``, ``torch.stft``) and, for the HTK mel matrix,
``transformers.audio_utils.mel_filter_bank``.  Each test names the reference call site the op restates.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import logmel_oracle as LM
from oracle import wav2vec2_oracle as V
from oracle import whisper_oracle as O


def test_layer_norm_matches_torch():
    """tf.keras.layers.LayerNormalization(epsilon=1e-5), W:214 etc."""
    g = torch.Generator().manual_seed(1)
    for shape in ((3, 7, 768), (5, 384), (2, 3, 4, 32)):
        x = torch.randn(shape, generator=g, dtype=torch.float64) * 3 + 0.7
        gamma = torch.randn(shape[-1], generator=g, dtype=torch.float64)
        beta = torch.randn(shape[-1], generator=g, dtype=torch.float64)
        ref = F.layer_norm(x, (shape[-1],), gamma, beta, eps=1e-5)
        assert torch.allclose(O.layer_norm(x, gamma, beta, 1e-5), ref, rtol=1e-12, atol=1e-12)


def test_group_norm_matches_torch():
    """V:140-196 (channels-last, contiguous channel groups) vs F.group_norm (channels-first)."""
    g = torch.Generator().manual_seed(2)
    for B, T, C, G in ((2, 50, 512, 16), (1, 7, 32, 4), (3, 11, 256, 8)):
        x = torch.randn(B, T, C, generator=g, dtype=torch.float64) * 2 - 0.3
        gamma = torch.randn(C, generator=g, dtype=torch.float64)
        beta = torch.randn(C, generator=g, dtype=torch.float64)
        ref = F.group_norm(x.transpose(1, 2), G, gamma, beta, eps=1e-5).transpose(1, 2)
        assert torch.allclose(V.group_norm(x, gamma, beta, G, 1e-5), ref, rtol=1e-11, atol=1e-11)


def test_gelu_matches_torch_exact_erf():
    """tf.keras.activations.gelu(approximate=False), W:195; V:132-136."""
    x = torch.linspace(-8, 8, 4001, dtype=torch.float64)
    assert torch.allclose(O.gelu_erf(x), F.gelu(x, approximate="none"), rtol=0, atol=1e-15)
    assert torch.allclose(V.gelu_erf(x), F.gelu(x, approximate="none"), rtol=0, atol=1e-15)
    # and it is NOT the tanh approximation (max gap ~5e-4)
    assert float((O.gelu_erf(x) - F.gelu(x, approximate="tanh")).abs().max()) > 1e-4


def test_conv1d_same_stride1_matches_torch_same():
    """Keras Conv1D(padding="same"), stride 1 (W:311; V:271-277 with an even kernel: left 63 / right 64):
    torch's padding="same" uses the same split (left = total // 2)."""
    g = torch.Generator().manual_seed(3)
    for k, cin, cout, groups in ((3, 5, 7, 1), (128, 32, 32, 4), (4, 6, 6, 2)):
        x = torch.randn(2, 40, cin, generator=g, dtype=torch.float64)
        w = torch.randn(k, cin // groups, cout, generator=g, dtype=torch.float64)
        b = torch.randn(cout, generator=g, dtype=torch.float64)
        ref = F.conv1d(x.transpose(1, 2), w.permute(2, 1, 0), b, stride=1, padding="same", groups=groups).transpose(1, 2)
        got = V.conv1d_same(x, w, b, 1, groups=groups)
        assert torch.allclose(got, ref, rtol=1e-11, atol=1e-11)
        if groups == 1:
            assert torch.allclose(O.conv1d_same(x, w, b, 1), ref, rtol=1e-11, atol=1e-11)


def test_conv1d_same_strided_matches_explicit_windows():
    """TF SAME for stride s: out = ceil(T / s), pad_total = max((out-1)*s + k - T, 0), left = total // 2
    (W:312 k3 s2 on T = 3000 -> (0, 1)); checked against explicit window dot products."""
    g = torch.Generator().manual_seed(4)
    for T, k, s in ((30, 3, 2), (31, 3, 2), (40, 10, 5), (17, 2, 2)):
        x = torch.randn(1, T, 3, generator=g, dtype=torch.float64)
        w = torch.randn(k, 3, 2, generator=g, dtype=torch.float64)
        out = -(-T // s)
        total = max((out - 1) * s + k - T, 0)
        left = total // 2
        xp = torch.zeros(1, T + total, 3, dtype=torch.float64)
        xp[:, left:left + T] = x
        ref = torch.stack([torch.einsum("kc,kcd->d", xp[0, j * s:j * s + k], w) for j in range(out)])[None]
        assert torch.allclose(O.conv1d_same(x, w, None, s), ref, rtol=1e-12, atol=1e-12)


def _unit_attention_params(d, prefix="a"):
    eye = torch.eye(d, dtype=torch.float64)
    p = {f"{prefix}.{n}.kernel": eye.clone() for n in ("q_proj", "k_proj", "v_proj", "out_proj")}
    p.update({f"{prefix}.{n}.bias": torch.zeros(d, dtype=torch.float64) for n in ("q_proj", "k_proj", "v_proj", "out_proj")})
    return p


def test_attention_core_matches_sdpa_with_additive_mask():
    """W:141-167: q scaled by hd^-0.5, additive (1 - mask) * -1e9, softmax, probs @ v.  Identity projections
    isolate the core, which must equal F.scaled_dot_product_attention with the same additive mask."""
    g = torch.Generator().manual_seed(5)
    B, H, S, hd = 2, 3, 9, 8
    d = H * hd
    x = torch.randn(B, S, d, generator=g, dtype=torch.float64)
    enc = torch.randn(B, 14, d, generator=g, dtype=torch.float64)
    p = _unit_attention_params(d)

    def heads(t):
        return t.reshape(B, -1, H, hd).permute(0, 2, 1, 3)

    # encoder self-attention (no mask) and cross-attention
    for kv in (None, enc):
        src = x if kv is None else kv
        ref = F.scaled_dot_product_attention(heads(x), heads(src), heads(src), scale=hd ** -0.5)
        ref = ref.permute(0, 2, 1, 3).reshape(B, S, d)
        got = O.mha(p, "a", x, kv, None, H, 0.0, True)
        assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    # decoder self-attention with the reference's inverted mask: rows 0..S-2 only (the fully-masked last row is
    # pinned separately: -1e9 absorbs the scores in fp32 and the row is exactly uniform)
    mask = torch.from_numpy(O.decoder_mask(S))[None]
    add = ((1.0 - mask) * -1e9).to(torch.float64)
    ref = F.scaled_dot_product_attention(heads(x), heads(x), heads(x), attn_mask=add[None], scale=hd ** -0.5)
    ref = ref.permute(0, 2, 1, 3).reshape(B, S, d)
    got = O.mha(p, "a", x, None, mask, H, 0.0, True)
    assert torch.allclose(got[:, :-1], ref[:, :-1], rtol=1e-9, atol=1e-9)
    assert torch.allclose(got[:, -1], x.mean(dim=1), rtol=1e-6, atol=1e-6)  # uniform over all S keys (v = x)


def test_w2v_attention_scale_after_scores_matches_sdpa():
    """V:348-349: scores / sqrt(hd) AFTER q.k^T, no mask."""
    g = torch.Generator().manual_seed(6)
    B, H, T, hd = 2, 4, 10, 8
    d = H * hd
    x = torch.randn(B, T, d, generator=g, dtype=torch.float64)
    p = _unit_attention_params(d)

    def heads(t):
        return t.reshape(B, T, H, hd).permute(0, 2, 1, 3)

    ref = F.scaled_dot_product_attention(heads(x), heads(x), heads(x)).permute(0, 2, 1, 3).reshape(B, T, d)
    assert torch.allclose(V.attention(p, "a", x, H), ref, rtol=1e-10, atol=1e-10)


def test_double_shift_cross_entropy_matches_manual_logsumexp():
    """W:585-600: SparseCategoricalCrossentropy(from_logits=True) over logits[:, :-1] / labels[:, 1:], mean."""
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(3, 6, 11, generator=g, dtype=torch.float64)
    labels = torch.randint(0, 11, (3, 6), generator=g)
    sl, sy = logits[:, :-1], labels[:, 1:]
    manual = (torch.logsumexp(sl, -1) - sl.gather(-1, sy[..., None])[..., 0]).mean()
    got = F.cross_entropy(sl.reshape(-1, 11), sy.reshape(-1), reduction="mean")
    assert float(got) == pytest.approx(float(manual), rel=1e-13)


def test_adam_torch_eps_mode_matches_torch_optim():
    """eps_mode="torch" of the oracle's Adam is torch.optim.Adam; eps_mode="tf" (Keras V2, W:901) is NOT."""
    g = torch.Generator().manual_seed(8)
    w0 = torch.randn(37, generator=g, dtype=torch.float64)
    grads = [torch.randn(37, generator=g, dtype=torch.float64) for _ in range(6)]
    w = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adam([w], lr=1e-2, betas=(0.9, 0.999), eps=1e-3)
    p_torch, p_tf = {"w": w0.clone()}, {"w": w0.clone()}
    st_torch, st_tf = O.AdamState(), O.AdamState()
    for gr in grads:
        w.grad = gr.clone()
        opt.step()
        O.adam_step(p_torch, {"w": gr}, st_torch, lr=1e-2, eps=1e-3, eps_mode="torch")
        O.adam_step(p_tf, {"w": gr}, st_tf, lr=1e-2, eps=1e-3, eps_mode="tf")
    assert torch.allclose(p_torch["w"], w.detach(), rtol=1e-12, atol=1e-12)
    assert float((p_tf["w"] - w.detach()).abs().max()) > 1e-4  # the epsilon placements really differ
    # Keras V2 closed form, written independently of oracle.adam_step: theta -= lr_t * m / (sqrt(v) + eps)
    m = v = torch.zeros(37, dtype=torch.float64)
    th = w0.clone()
    for t, gr in enumerate(grads, 1):
        m = 0.9 * m + 0.1 * gr
        v = 0.999 * v + 0.001 * gr * gr
        th = th - 1e-2 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (v.sqrt() + 1e-3)
    assert torch.allclose(p_tf["w"], th, rtol=1e-12, atol=1e-12)


def test_clip_by_global_norm_matches_torch_clip_grad_norm():
    """tf.clip_by_global_norm (V:1243): g * clip / max(norm, clip).  torch's clip_grad_norm_ uses
    clip / (norm + 1e-6) clamped to 1: equal to ~1e-6 relative when norm > clip, identity otherwise."""
    g = torch.Generator().manual_seed(9)
    for scale in (5.0, 0.01):
        grads = {k: torch.randn(n, generator=g, dtype=torch.float64) * scale for k, n in (("a", 13), ("b", 40))}
        params = [torch.nn.Parameter(torch.zeros_like(v)) for v in grads.values()]
        for prm, v in zip(params, grads.values()):
            prm.grad = v.clone()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        out = V.clip_by_global_norm({k: v.clone() for k, v in grads.items()}, 1.0)
        out = out[0] if isinstance(out, tuple) else out
        for prm, k in zip(params, grads):
            assert torch.allclose(out[k], prm.grad, rtol=1e-5, atol=1e-12)


def test_logmel_matches_torch_stft_and_transformers_mel():
    """W:739-766: tf.signal.stft(400, 160, fft 400, periodic Hann, pad_end=False) -> power -> HTK mel -> log(x + 1e-6).
    STFT against torch.stft (center=False), mel matrix against transformers.audio_utils.mel_filter_bank with
    triangles taken in the mel domain (tf.signal.linear_to_mel_weight_matrix's rule)."""
    rng = np.random.default_rng(10)
    wav = rng.standard_normal(16000).astype(np.float64)
    spec = torch.stft(torch.from_numpy(wav), n_fft=400, hop_length=160, win_length=400,
                      window=torch.hann_window(400, periodic=True, dtype=torch.float64), center=False,
                      return_complex=True)  # [201, frames]
    power = (spec.real ** 2 + spec.imag ** 2).T.numpy()
    assert power.shape == (1 + (16000 - 400) // 160, 201)
    au = pytest.importorskip("transformers.audio_utils")
    mel_tf = au.mel_filter_bank(num_frequency_bins=201, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                                sampling_rate=16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)
    mel_or = LM.linear_to_mel_weight_matrix(80, 201, 16000, 0.0, 8000.0)
    assert mel_or.shape == mel_tf.shape == (201, 80)
    assert np.allclose(mel_or, mel_tf, rtol=0, atol=1e-9)
    assert np.all(mel_or[0] == 0.0)  # DC bin excluded
    ref = np.log(power @ mel_tf + 1e-6)
    got = LM.extract_fbank_features(wav)
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-9)
