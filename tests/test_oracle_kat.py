"""Known-answer tests that pin the oracle: the reference has no golden
vectors, so these closed forms, derived from the source text of
speech_jobs/whisper_dist.py, are what anchors the restatement."""
import math

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O


def test_positional_encoding_closed_form():
    for d in (8, 768):
        pe = O.positional_encoding(50, d)
        for p in (0, 1, 7, 49):
            for i in (0, 2, d - 2):
                ang = p * math.exp(-i * math.log(10000.0) / d)
                assert pe[p, i] == pytest.approx(math.sin(ang), abs=1e-6)
                assert pe[p, i + 1] == pytest.approx(math.cos(ang), abs=1e-6)
    assert O.positional_encoding(3, 8).dtype == np.float32


def test_decoder_mask_is_inverted():
    for S in (1, 4, 100):
        m = O.decoder_mask(S)
        add = (1.0 - m) * -1e9
        for i in range(S):
            for j in range(S):
                assert (add[i, j] == 0.0) == (j > i)  # only strictly-future keys stay visible
    # fully masked last row -> exactly uniform probabilities in fp32
    S = 5
    cfg = O.make_config("small", d_model=8, decoder_attention_heads=1)
    p = {f"a.{n}.kernel": torch.randn(8, 8) for n in ("q_proj", "k_proj", "v_proj", "out_proj")}
    p.update({f"a.{n}.bias": torch.zeros(8) for n in ("q_proj", "k_proj", "v_proj", "out_proj")})
    x = torch.randn(1, S, 8)
    q = (x @ p["a.q_proj.kernel"]) * 8 ** -0.5
    k = x @ p["a.k_proj.kernel"]
    s = (q @ k.transpose(-1, -2)) + torch.from_numpy((1.0 - O.decoder_mask(S)) * -1e9)
    pr = torch.softmax(s, -1)
    assert torch.all(pr[0, -1] == pr[0, -1, 0])
    assert float(pr[0, 0, 0]) == 0.0 and float(pr[0, 0, 1:].sum()) == pytest.approx(1.0)


def test_same_padding_lengths_and_splits():
    assert O.same_pad(3000, 3, 1) == (3000, 1, 1)
    assert O.same_pad(3000, 3, 2) == (1500, 0, 1)
    assert O.same_pad(100, 128, 1) == (100, 63, 64)
    T = 32000
    outs = []
    for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        T = O.same_pad(T, k, s)[0]
        outs.append(T)
    assert outs == [6400, 3200, 1600, 800, 400, 200, 100]
    T = 80000
    for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
        T = O.same_pad(T, k, s)[0]
    assert T == 250


def test_conv1d_same_hand_vector():
    # k=3, s=2, T=4: out=2, pad (0,1); y[t] = sum_kk x[2t+kk] * w[kk]
    x = torch.tensor([[[1.0], [2.0], [3.0], [4.0]]])
    w = torch.tensor([[[1.0]], [[10.0]], [[100.0]]])
    y = O.conv1d_same(x, w, None, 2)
    assert y.flatten().tolist() == [321.0, 43.0]
    # k=3, s=1: pad (1,1)
    y = O.conv1d_same(x, w, None, 1)
    assert y.flatten().tolist() == [210.0, 321.0, 432.0, 43.0]


def test_label_recipe():
    f, l = O.create_dummy_pool(seed=1234)
    assert f.shape == (50, 80, 3000) and f.dtype == np.float32
    assert l.shape == (50, 100) and l.dtype == np.int32
    for row in l:
        assert row[0] == 1
        L = int(np.max(np.nonzero(row)[0])) + 1
        assert 50 <= L < 90 and row[L - 1] == 2
        assert np.all((row[1:L - 1] >= 3) & (row[1:L - 1] < 100))
        assert np.all(row[L:] == 0)
    b = O.batches(f, l, 8)
    sizes = [next(b)[0].shape[0] for _ in range(8)]
    assert sizes == [8, 8, 8, 8, 8, 8, 2, 8]  # no drop_remainder (W:815)


def test_parameter_counts():
    assert O.param_count(O.make_config("tiny")) == 56_933_376
    assert O.param_count(O.make_config("small")) == 147_781_632
    assert O.param_count(O.make_config("large")) == 1_607_321_600


def test_decoder_input_shift():
    lab = torch.tensor([[1, 5, 6, 2, 0]], dtype=torch.int32)
    assert O.decoder_input_ids(lab, 50257).tolist() == [[50257, 1, 5, 6, 2]]


def _small():
    cfg = O.make_config("small", d_model=32, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=64,
                        vocab_size=128, encoder_layers=1, decoder_layers=2, n_mels=8, n_ctx=12,
                        decoder_start_token_id=127, dropout=0.0, attention_dropout=0.0)
    f, l = O.create_dummy_pool(seed=2, n_mels=8, seq_len=24, max_target_length=6, num_samples=4)
    return cfg, f, l


def test_init_loss_is_about_log_vocab():
    cfg, f, l = _small()
    p = O.init_params(cfg, dtype=torch.float64)
    loss, _ = O.forward_loss(p, torch.from_numpy(f[:2]), torch.from_numpy(l[:2]), cfg)
    assert abs(float(loss) - math.log(cfg.vocab_size)) < 0.6


def test_double_shift_scoring():
    """logits[:, t] is scored against labels[:, t+1] (W:585-586) and the last position is unused."""
    cfg, f, l = _small()
    p = O.init_params(cfg, dtype=torch.float64)
    ft, lt = torch.from_numpy(f[:2]), torch.from_numpy(l[:2])
    loss, logits = O.forward_loss(p, ft, lt, cfg)
    lp = torch.log_softmax(logits, -1)
    manual = -sum(lp[b, t, int(lt[b, t + 1])] for b in range(2) for t in range(5)) / 10
    assert float(loss) == pytest.approx(float(manual), rel=1e-12)


def test_finite_difference_gradients_fp64(monkeypatch):
    monkeypatch.setattr(O, "FP32_MASK_ROUNDING", False)  # see the flag's comment in the oracle
    monkeypatch.setattr(O, "MASK_VALUE", -1e4)  # -1e9 + score quantises the score even in fp64
    cfg, f, l = _small()
    p = O.init_params(cfg, dtype=torch.float64)
    ft, lt = torch.from_numpy(f[:2]), torch.from_numpy(l[:2])
    _, g = O.loss_and_grads(p, ft, lt, cfg)
    rng = np.random.default_rng(0)
    for name in ["encoder.conv1.kernel", "encoder.conv2.bias", "encoder.layers.0.self_attn.k_proj.kernel",
                 "encoder.layers.0.final_layer_norm.gamma", "decoder.layers.1.encoder_attn.v_proj.kernel",
                 "decoder.layers.0.feed_forward.fc1.bias", "decoder.embed_tokens.embeddings", "lm_head.kernel"]:
        w = p[name]
        gi = g[name]
        flat = torch.nonzero(gi.abs() > gi.abs().max() * 0.2)
        idx = tuple(flat[int(rng.integers(0, len(flat)))].tolist())
        old = float(w[idx])
        eps = 1e-6
        w[idx] = old + eps
        lp = float(O.forward_loss(p, ft, lt, cfg)[0])
        w[idx] = old - eps
        lm = float(O.forward_loss(p, ft, lt, cfg)[0])
        w[idx] = old
        assert (lp - lm) / (2 * eps) == pytest.approx(float(gi[idx]), rel=2e-5, abs=1e-9), name


def test_adam_three_hand_computed_steps():
    g = [torch.tensor([0.1, -0.2, 0.0, 0.3], dtype=torch.float64)] * 3
    for mode in ("tf", "torch"):
        p = {"w": torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)}
        st = O.AdamState()
        w = np.array([1.0, 2.0, 3.0, 4.0])
        m = np.zeros(4)
        v = np.zeros(4)
        for t in range(1, 4):
            O.adam_step(p, {"w": g[t - 1]}, st, lr=1e-2, eps=1e-7, eps_mode=mode)
            gn = g[t - 1].numpy()
            m = 0.9 * m + 0.1 * gn
            v = 0.999 * v + 0.001 * gn * gn
            if mode == "tf":
                w = w - 1e-2 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
            else:
                w = w - 1e-2 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-7)
            assert np.allclose(p["w"].numpy(), w, rtol=0, atol=1e-15)
    # constant gradient => first TF step moves by ~lr (sign), zero gradient does not move
    p = {"w": torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)}
    O.adam_step(p, {"w": g[0]}, O.AdamState(), lr=1e-2, eps=1e-7)
    assert p["w"][2] == 3.0 and float(p["w"][0]) == pytest.approx(1.0 - 1e-2, abs=1e-6)  # eps shifts it by lr*eps/(sqrt(1-b2)|g|)


def test_data_parallel_sum_semantics():
    """N replicas x B == gradient SUM and loss SUM (no 1/N), W:829-836 + W:848."""
    cfg, f, l = _small()
    p = O.init_params(cfg, dtype=torch.float64)
    l2, _ = O.train_steps(cfg, {k: v.clone() for k, v in p.items()}, f, l, 2, 1, n_replicas=2)
    la, ga = O.loss_and_grads(p, torch.from_numpy(f[:2]), torch.from_numpy(l[:2]), cfg)
    lb, gb = O.loss_and_grads(p, torch.from_numpy(f[2:4]), torch.from_numpy(l[2:4]), cfg)
    assert l2[0] == pytest.approx(float(la) + float(lb), rel=1e-12)
    l1, g1 = O.loss_and_grads(p, torch.from_numpy(f[:4]), torch.from_numpy(l[:4]), cfg)
    # sum of two half-batch mean-loss gradients = 2 x gradient of the full-batch mean loss
    for k in ga:
        assert torch.allclose(ga[k] + gb[k], 2 * g1[k], rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------ Wav2Vec2 oracle (V:)
from oracle import wav2vec2_oracle as V2  # noqa: E402


def test_w2v_parameter_counts_and_lengths():
    assert V2.param_count(V2.make_config("base")) == 92_297_728
    assert V2.param_count(V2.make_config("small")) == 20_466_816
    assert V2.feature_lengths(V2.make_config("base"), 32000) == [6400, 3200, 1600, 800, 400, 200, 100]
    assert V2.feature_lengths(V2.make_config("base"), 80000)[-1] == 250
    assert V2.make_config("anything-else").hidden_size == 768  # V:49: the else branch is base


def test_w2v_group_norm_hand_vector():
    # B=1, T=2, C=4, G=2: group 0 = channels {0,1}, statistics over (time, 2 channels)
    x = torch.tensor([[[1.0, 3.0, 10.0, 10.0], [5.0, 7.0, 10.0, 14.0]]])
    y = V2.group_norm(x, torch.ones(4), torch.zeros(4), 2, eps=0.0)
    g0 = (torch.tensor([1.0, 3.0, 5.0, 7.0]) - 4.0) / math.sqrt(5.0)
    assert torch.allclose(y[0, :, :2].reshape(-1), g0[[0, 1, 2, 3]])
    g1 = (torch.tensor([10.0, 10.0, 10.0, 14.0]) - 11.0) / math.sqrt(3.0)
    assert torch.allclose(y[0, :, 2:].reshape(-1), g1)


def test_w2v_negative_sampling_recipe():
    rng = np.random.default_rng(0)
    neg = V2.sample_negative_indices(rng, 4, 100, 100)
    assert neg.shape == (4, 100) and neg.dtype == np.int32
    for row in neg:
        assert len(set(row[:99].tolist())) == 99          # K = T-1 distinct positions ...
        assert row[99] == row[0]                            # ... tiled, then cut to 100 (V:924-929)
    r = np.random.default_rng(0).integers(0, 100, size=(4, 100))
    first = int(np.flatnonzero(r[0] == r[0].min())[0])      # smallest draw, lowest index on ties
    assert neg[0, 0] == first
    assert V2.sample_negative_indices(np.random.default_rng(1), 2, 250, 100).shape == (2, 100)  # K = 100 < T-1


def _w2v_small():
    cfg = V2.make_config("base", hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128,
                         conv_dim=(32, 32, 32), conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2),
                         num_conv_pos_embeddings=8, num_conv_pos_embedding_groups=4, num_codevectors_per_group=16,
                         codevector_dim=32, proj_codevector_dim=32, num_negatives=10)
    pool = V2.create_dummy_pool(seed=1, num_samples=4, length=400)
    neg = V2.sample_negative_indices(np.random.default_rng(0), 2, 20, 10)
    return cfg, pool, neg


def test_w2v_dead_gradients_and_loss_assembly():
    cfg, pool, neg = _w2v_small()
    p = V2.init_params(cfg, dtype=torch.float64)
    loss, g, out = V2.loss_and_grads(p, torch.from_numpy(pool[:2]), torch.from_numpy(neg), cfg, num_replicas=2)
    assert float(g["quantizer.projection.kernel"].abs().max()) == 0.0   # argmin/one-hot: no gradient (V:631-638)
    assert float(g["quantizer.projection.bias"].abs().max()) == 0.0
    assert float(g["quantizer.codevectors"].abs().max()) > 0.0
    _, cl = V2.contrastive_loss(out["projected_states"], out["projected_quantized_features"], torch.from_numpy(neg), 0.1)
    assert float(loss) == pytest.approx((float(cl) - 0.1 * float(out["codevector_perplexity"])) / 2, rel=1e-12)
    assert 1.0 <= float(out["codevector_perplexity"]) <= 16.0


def test_w2v_finite_differences_fp64():
    cfg, pool, neg = _w2v_small()
    p = V2.init_params(cfg, dtype=torch.float64)
    a, n = torch.from_numpy(pool[:2]), torch.from_numpy(neg)
    _, g, _ = V2.loss_and_grads(p, a, n, cfg)
    for name in ["feature_extractor.conv_layers.0.conv.kernel", "feature_extractor.conv_layers.1.norm.gamma",
                 "feature_extractor.pos_conv_embed.kernel", "encoder.layers.0.attention.k_proj.kernel",
                 "project_q.dense.kernel", "quantizer.codevectors"]:
        w, gi = p[name], g[name]
        idx = tuple(torch.nonzero(gi.abs() > gi.abs().max() * 0.3)[0].tolist())
        old, e = float(w[idx]), 1e-6
        w[idx] = old + e
        lp = float(V2.step_loss(p, a, n, cfg)[0])
        w[idx] = old - e
        lm = float(V2.step_loss(p, a, n, cfg)[0])
        w[idx] = old
        assert (lp - lm) / (2 * e) == pytest.approx(float(gi[idx]), rel=2e-5), name


def test_w2v_clipping_rules():
    g = {"a": torch.tensor([3.0, 4.0]), "b": torch.tensor([0.0, 12.0])}
    c, n = V2.clip_by_global_norm(g, 1.0)
    assert n == pytest.approx(13.0) and torch.allclose(c["a"], g["a"] / 13) and torch.allclose(c["b"], g["b"] / 13)
    c, _ = V2.clip_by_global_norm({"a": torch.tensor([0.3, 0.4])}, 1.0)
    assert torch.allclose(c["a"], torch.tensor([0.3, 0.4]))  # norm below the clip: untouched
    e = V2.clip_by_norm_each(g, 1.0)
    assert torch.allclose(e["a"], g["a"] / 5) and torch.allclose(e["b"], g["b"] / 12)


# ----------------------------------------------------------------------------- dropout generator
def test_dropout_generator_known_answers_and_statistics():
    """The counter-based dropout generator (csrc/tmi_common.h, restated in oracle/dropout.py) is part of the
    step's definition when dropout is on: pin it with frozen vectors, and check that it behaves like independent
    Bernoulli draws (rate, adjacent / cross-row / cross-stream / cross-step correlation)."""
    from oracle import dropout as D
    assert [int(D.pair_hash(D.row_key(k, r), c)) for k, r, c in
            ((0, 0, 0), (0, 1, 0), (0xDEADBEEF, 12345, 77), (0x12345678, 0xFFFFFFFF, 0xFFFFFF))] == \
        [2356614601, 2907385536, 1512887188, 2692573910]
    assert [int(x) for x in D.row_key(0xDEADBEEF, 12345)] == [2238520111, 1453611697]
    assert (int(D.mix32(1)), int(D.mix32(0xDEADBEEF)), int(D.stream_key(0x0123456789ABCDEF, 7))) == \
        (1753845952, 3861431939, 1429204582)
    assert D.drop_thr(0.1) == 6554 and D.drop_thr(0.25) == 16384
    assert abs(D.keep_scale(0.1) - 65536.0 / (65536 - 6554)) < 1e-6
    assert D.keep_flat(42, 4, 8, 0.1).astype(int).tolist() == [[1, 0, 1, 1, 0, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1, 0],
                                                              [0, 1, 0, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1, 1]]
    # the attention generator (tmi_quad_hash / tmi_keep_attn: one evaluation per four keys, signed 16-bit draws); the frozen
    # values come from a scalar restatement in plain Python integers, not from the numpy code they pin
    assert [int(x) for x in D.quad_hash(D.row_key(0xDEADBEEF, 12345), 77)] == [1512887188, 2932179144]
    assert D.keep_attention(42, 1, 2, 3, 6, 0.5).astype(int).tolist() == \
        [[[[0, 1, 0, 1, 0, 1], [0, 1, 0, 0, 0, 1], [1, 0, 0, 1, 1, 1]], [[1, 0, 1, 0, 1, 1], [0, 1, 0, 0, 0, 1], [0, 1, 0, 1, 0, 1]]]]
    assert D.site_seed(0xC0FFEE, 3, 204) == 17806544269414322833
    assert D.site_id("decoder.layers.3.encoder_attn") == 403 and D.w2v_site_id("encoder.layers.11.attention_output") == 211
    m = D.keep_flat(7, 600, 768, 0.1)
    n = m.size
    assert abs(m.mean() - (1 - 6554 / 65536)) < 4 * (0.3 / np.sqrt(n))
    for a, b in ((m[:, :-1], m[:, 1:]), (m[:-1], m[1:]), (m[:, :-2], m[:, 2:])):
        assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 5 / np.sqrt(n)
    att4 = D.keep_attention(11, 2, 4, 96, 160, 0.1)
    att = att4.reshape(8, -1).astype(float)
    c = np.corrcoef(att)
    assert np.abs(c - np.eye(8)).max() < 5 / np.sqrt(att.shape[1])      # streams (batch, head)
    # the attention mask behaves like independent draws too: rate, neighbours along keys and queries, and the four draws
    # that share one generator evaluation (keys 4cq .. 4cq+3) pairwise
    big = D.keep_attention(3, 1, 2, 512, 1024, 0.1)[0]
    nb = big.size
    assert abs(big.mean() - (1 - 6554 / 65536)) < 4 * (0.3 / np.sqrt(nb))
    for a, b in ((big[..., :-1], big[..., 1:]), (big[:, :-1], big[:, 1:]), (big[..., :-4], big[..., 4:])):
        assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 5 / np.sqrt(nb)
    quad = big.reshape(2, 512, 256, 4)
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(np.corrcoef(quad[..., i].ravel(), quad[..., j].ravel())[0, 1]) < 5 / np.sqrt(nb / 4)
    s0 = D.keep_flat(D.site_seed(5, 0, 100), 300, 256, 0.1).ravel().astype(float)
    s1 = D.keep_flat(D.site_seed(5, 1, 100), 300, 256, 0.1).ravel().astype(float)
    s2 = D.keep_flat(D.site_seed(5, 0, 101), 300, 256, 0.1).ravel().astype(float)
    assert abs(np.corrcoef(s0, s1)[0, 1]) < 5 / np.sqrt(s0.size) and abs(np.corrcoef(s0, s2)[0, 1]) < 5 / np.sqrt(s0.size)


def test_dropout_hash_uses_every_counter_and_key_bit():
    """ADVICE r1 (tmi_common.h:89): the pair hash multiplies 24-bit quantities, so whatever must not alias has to reach
    those 24 bits first.  Rows and stream keys go through two full 32-bit avalanches per row (tmi_row_key), columns are
    < 2^17 by contract (TMI_DROP_MAX_COLS, checked by the entry points): rows 2^24 apart, stream keys equal in their low 24 bits, and the two draws of one pair are all
    independent -- agreement of two independent p = 0.1 masks is 0.9^2 + 0.1^2 = 0.82."""
    from oracle import dropout as DO
    thr = DO.drop_thr(0.1)
    key = DO.stream_key(0x1234567890ABCDEF, 7)
    rows = np.arange(512, dtype=np.uint64)[:, None]
    cols = np.arange(512, dtype=np.uint64)[None, :]

    def keep(key, rows):
        h = DO.pair_hash(DO.row_key(key, rows), cols >> np.uint64(1))
        return np.where(cols & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF)) >= np.uint64(thr)

    a = keep(key, rows)
    assert np.array_equal(a, DO.keep_rows(key, 512, 512, thr))
    b = keep(key, rows + np.uint64(1 << 24))
    assert abs(a.mean() - 0.9) < 3e-3 and abs(b.mean() - 0.9) < 3e-3
    assert abs((a == b).mean() - 0.82) < 5e-3
    key2 = (int(key) ^ 0xA5000000) & 0xFFFFFFFF  # differs in the top byte only
    c = keep(np.uint64(key2), rows)
    assert abs((a == c).mean() - 0.82) < 5e-3
    # low and high 16-bit draws of one pair are independent too, and so are columns 2^16 apart (the top of the contract)
    assert abs((a[:, 0::2] == a[:, 1::2]).mean() - 0.82) < 5e-3
    h0 = DO.pair_hash(DO.row_key(key, rows), cols >> np.uint64(1))
    h1 = DO.pair_hash(DO.row_key(key, rows), (cols >> np.uint64(1)) + np.uint64(1 << 15))
    assert abs((((h0 & np.uint64(0xFFFF)) >= np.uint64(thr)) == ((h1 & np.uint64(0xFFFF)) >= np.uint64(thr))).mean() - 0.82) < 5e-3
