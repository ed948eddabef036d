"""CPU oracle for the Whisper data-parallel training step.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, op for op, of the TensorFlow graph that
``/root/reference/speech_jobs/whisper_dist.py`` (cited below as ``W:``) builds for one
training step.  It is the *checker* for the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The
product package (``tethys-speech_amd/``) never imports anything from ``oracle/``.

PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures (SURVEY.md
section 4 / 8c) and its arithmetic lives in TensorFlow 2.10 (NGC 22.12 image,
``Dockerfile:1``), which is absent from ``/root/reference`` and not installed here.  The
oracle is therefore pinned only by closed-form known-answer tests derived from the source
text (``tests/test_oracle_kat.py``) and by finite-difference gradient checks in fp64.  TF
semantics encoded by hand: Keras ``Conv1D`` "same" padding (pad_total = max((ceil(T/s)-1)*s
+ k - T, 0), left = pad_total // 2), kernel layout [k, C_in, C_out]; ``Dense`` kernel
[in, out]; ``LayerNormalization`` biased variance; ``gelu(approximate=False)`` (exact erf);
``SparseCategoricalCrossentropy(from_logits=True)``; Keras-V2 ``Adam`` (epsilon added to
sqrt(v), bias correction folded into lr_t); fp32 rounding of ``scores + (-1e9)``.

The graph is kept *unfused* on purpose (separate q/k/v/out GEMMs, materialised
[B,H,T,T] scores, separate softmax / GELU / LayerNorm ops, dense logits, per-tensor Adam):
it doubles as the "restated reference CPU path" timed by ``bench.py``.

dtype is a parameter: float64 for gradient checks and kernel tolerances, float32 to mirror
the reference's arithmetic.  Dropout rates are a parameter and 0 in parity mode (TF's RNG
stream cannot be reproduced).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# W:10-45  WhisperConfig, W:852-890 create_whisper_model size table
# --------------------------------------------------------------------------------------
@dataclass
class WhisperConfig:
    d_model: int = 768
    encoder_layers: int = 4
    encoder_attention_heads: int = 12
    decoder_layers: int = 4
    decoder_attention_heads: int = 12
    d_ff: int = 3072
    n_mels: int = 80
    n_ctx: int = 1500
    vocab_size: int = 51865
    max_target_positions: int = 448
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.0
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    decoder_start_token_id: int = 50257


_SIZES = {  # W:859-886; "small" keeps the defaults (W:888)
    "tiny": dict(d_model=384, encoder_layers=4, encoder_attention_heads=6,
                 decoder_layers=4, decoder_attention_heads=6, d_ff=1536),
    "base": dict(d_model=512, encoder_layers=6, encoder_attention_heads=8,
                 decoder_layers=6, decoder_attention_heads=8, d_ff=2048),
    "small": dict(),
    "medium": dict(d_model=1024, encoder_layers=24, encoder_attention_heads=16,
                   decoder_layers=24, decoder_attention_heads=16, d_ff=4096),
    "large": dict(d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                  decoder_layers=32, decoder_attention_heads=20, d_ff=5120),
}


def make_config(model_type: str = "small", **overrides) -> WhisperConfig:
    """W:852-890.  Unknown names fall through to the defaults, as in the reference."""
    cfg = WhisperConfig()
    for k, v in _SIZES.get(model_type, {}).items():
        setattr(cfg, k, v)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


# --------------------------------------------------------------------------------------
# TF "SAME" padding (not in /root/reference: TensorFlow's documented rule)
# --------------------------------------------------------------------------------------
def same_pad(T: int, k: int, s: int) -> Tuple[int, int, int]:
    """-> (T_out, pad_left, pad_right) for Conv1D(padding="same")."""
    out = -(-T // s)
    total = max((out - 1) * s + k - T, 0)
    left = total // 2
    return out, left, total - left


# --------------------------------------------------------------------------------------
# W:49-69 PositionalEncoding (fixed sinusoid, interleaved sin/cos, added)
# --------------------------------------------------------------------------------------
def positional_encoding(max_len: int, d_model: int) -> np.ndarray:
    pe = np.zeros((max_len, d_model))
    position = np.arange(0, max_len)[:, np.newaxis]
    div_term = np.exp(np.arange(0, d_model, 2) * -(np.log(10000.0) / d_model))
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe.astype(np.float32)  # W:63 converts to float32


# --------------------------------------------------------------------------------------
# Parameter naming / shapes.  Order = Keras creation order does not matter for parity;
# names follow the reference attribute paths.
# --------------------------------------------------------------------------------------
def param_shapes(cfg: WhisperConfig) -> "Dict[str, Tuple[int, ...]]":
    d, ff, V = cfg.d_model, cfg.d_ff, cfg.vocab_size
    shapes: Dict[str, Tuple[int, ...]] = {}

    def mha(prefix):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):  # W:89-92
            shapes[f"{prefix}.{n}.kernel"] = (d, d)
            shapes[f"{prefix}.{n}.bias"] = (d,)

    def ln(prefix):
        shapes[f"{prefix}.gamma"] = (d,)
        shapes[f"{prefix}.beta"] = (d,)

    def ffn(prefix):  # W:194-197
        shapes[f"{prefix}.fc1.kernel"] = (d, ff)
        shapes[f"{prefix}.fc1.bias"] = (ff,)
        shapes[f"{prefix}.fc2.kernel"] = (ff, d)
        shapes[f"{prefix}.fc2.bias"] = (d,)

    shapes["encoder.conv1.kernel"] = (3, cfg.n_mels, d)  # W:311
    shapes["encoder.conv1.bias"] = (d,)
    shapes["encoder.conv2.kernel"] = (3, d, d)  # W:312
    shapes["encoder.conv2.bias"] = (d,)
    for i in range(cfg.encoder_layers):  # W:210-216
        p = f"encoder.layers.{i}"
        mha(f"{p}.self_attn")
        ln(f"{p}.self_attn_layer_norm")
        ffn(f"{p}.feed_forward")
        ln(f"{p}.final_layer_norm")
    ln("encoder.layer_norm")  # W:322
    shapes["decoder.embed_tokens.embeddings"] = (V, d)  # W:382
    for i in range(cfg.decoder_layers):  # W:240-253
        p = f"decoder.layers.{i}"
        mha(f"{p}.self_attn")
        ln(f"{p}.self_attn_layer_norm")
        mha(f"{p}.encoder_attn")
        ln(f"{p}.encoder_attn_layer_norm")
        ffn(f"{p}.feed_forward")
        ln(f"{p}.final_layer_norm")
    ln("decoder.layer_norm")  # W:392
    shapes["lm_head.kernel"] = (d, V)  # W:545, no bias, untied
    return shapes


def param_count(cfg: WhisperConfig) -> int:
    return sum(int(np.prod(s)) for s in param_shapes(cfg).values())


def init_params(cfg: WhisperConfig, seed: int = 1234, dtype=torch.float32) -> "Dict[str, torch.Tensor]":
    """Keras default initialisers (SURVEY a-14): glorot-uniform kernels, zero biases,
    Embedding U(-0.05, 0.05), LayerNorm gamma=1 beta=0.  The reference is unseeded; the
    seed and the draw order (dict order of ``param_shapes``) are this build's choice."""
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".kernel"):
            if len(shape) == 3:  # Conv1D [k, Cin, Cout]: fan_in = k*Cin, fan_out = k*Cout
                fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, size=shape)
        elif name.endswith(".embeddings"):
            a = rng.uniform(-0.05, 0.05, size=shape)
        elif name.endswith(".gamma"):
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        out[name] = torch.from_numpy(a.astype(np.float32)).to(dtype)
    return out


# --------------------------------------------------------------------------------------
# W:784-815 create_dummy_dataset
# --------------------------------------------------------------------------------------
def create_dummy_pool(seed: int = 1234, n_mels: int = 80, seq_len: int = 3000,
                      max_target_length: int = 100, num_samples: int = 50):
    """Features randn(50, n_mels, seq_len) f32; labels [50, max_target_length] i32 with
    [0]=BOS(1), [1:len-1]=randint(3,100), [len-1]=EOS(2), rest 0; len=randint(50,90).
    The reference uses the unseeded legacy global RNG; a seeded Generator stands in."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((num_samples, n_mels, seq_len)).astype(np.float32)
    labels = np.zeros((num_samples, max_target_length), dtype=np.int32)
    hi = min(90, max_target_length)
    lo = min(50, hi - 1)
    lengths = rng.integers(lo, hi, size=num_samples)
    for i in range(num_samples):
        labels[i, 0] = 1
        L = int(lengths[i])
        labels[i, 1:L - 1] = rng.integers(3, 100, size=L - 2)
        labels[i, L - 1] = 2
    return feats, labels


def batches(pool_feats, pool_labels, global_batch: int):
    """``dataset.batch(global_batch).repeat()`` (W:815): no drop_remainder, so the last
    batch of each pass over the 50-sample pool is short."""
    n = pool_feats.shape[0]
    while True:
        for s in range(0, n, global_batch):
            yield pool_feats[s:s + global_batch], pool_labels[s:s + global_batch]


# --------------------------------------------------------------------------------------
# Layers
# --------------------------------------------------------------------------------------
def gelu_erf(x):
    """tf.keras.activations.gelu(approximate=False), W:195,333,336."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, gamma, beta, eps):
    """tf.keras.layers.LayerNormalization(epsilon=eps): biased variance over last axis."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def dense(x, kernel, bias=None):
    y = x @ kernel
    return y if bias is None else y + bias


def conv1d_same(x, kernel, bias, stride):
    """Keras Conv1D(padding="same") on channels-last x [B,T,Cin], kernel [k,Cin,Cout]."""
    k = kernel.shape[0]
    _, left, right = same_pad(x.shape[1], k, stride)
    xt = F.pad(x.transpose(1, 2), (left, right))  # [B,Cin,T+pad]
    y = F.conv1d(xt, kernel.permute(2, 1, 0), bias=None, stride=stride)  # cross-correlation
    y = y.transpose(1, 2)
    return y if bias is None else y + bias


# When set (oracle.dropout.HostDropout), every Dropout site takes its keep mask from the counter-based
# generator the HIP kernels use, so a step WITH dropout can be compared mask for mask.
DROPOUT_PROVIDER = None


def dropout(x, rate, training, gen=None, site=None):
    """tf.keras.layers.Dropout: inverted dropout.  Only used when rate > 0 (perf mode)."""
    if not training or rate <= 0.0:
        return x
    if DROPOUT_PROVIDER is not None:
        keep, scale = DROPOUT_PROVIDER.mask(site, tuple(x.shape), rate)
        return x * keep.to(x.dtype) * scale
    keep = (torch.rand(x.shape, generator=gen, dtype=torch.float32) >= rate).to(x.dtype)
    return x * keep / (1.0 - rate)


def decoder_mask(S: int) -> np.ndarray:
    """W:416-418: ``1 - band_part(ones(S,S), -1, 0)`` -> 1 strictly above the diagonal."""
    return 1.0 - np.tril(np.ones((S, S), dtype=np.float32))


# True: masked scores take the fp32-rounded value of ``score + (-1e9)`` as in the reference
# (the score is absorbed).  The finite-difference test sets it False: with the rounding the
# forward is insensitive to a fully-masked row's scores while TF's (and autograd's) gradient
# of the add is still 1, so the analytic gradient is deliberately not the FD derivative.
FP32_MASK_ROUNDING = True
MASK_VALUE = -1e9  # W:153; the FD test lowers it so fp64 keeps the masked row's score bits


def mha(p, prefix, hidden, kv_states, mask, num_heads, attn_dropout=0.0, training=True):
    """W:106-176.  ``mask`` is the [1,S,S] tensor of W:416-418 or None.  The additive form
    (1 - mask) * -1e9 (W:152-153) is added in float32 so that, as in the reference, a
    masked score is *exactly* -1e9 and a fully-masked row softmaxes to uniform."""
    B, Tq, d = hidden.shape
    hd = d // num_heads
    scaling = hd ** -0.5
    src = hidden if kv_states is None else kv_states
    k = dense(src, p[f"{prefix}.k_proj.kernel"], p[f"{prefix}.k_proj.bias"])
    v = dense(src, p[f"{prefix}.v_proj.kernel"], p[f"{prefix}.v_proj.bias"])
    q = dense(hidden, p[f"{prefix}.q_proj.kernel"], p[f"{prefix}.q_proj.bias"]) * scaling  # W:141

    def split(t):
        return t.reshape(B, -1, num_heads, hd).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    scores = q @ k.transpose(-1, -2)  # W:147
    if mask is not None:
        add = ((1.0 - mask.to(torch.float32)) * MASK_VALUE)  # W:152-153, float32
        summed32 = scores.to(torch.float32) + add  # fp32 rounding absorbs the score
        masked = (add != 0).expand_as(scores)
        # identity derivative wrt scores, as tf.add has; value = fp32-rounded sum
        if FP32_MASK_ROUNDING:
            scores = torch.where(masked, scores + (summed32.to(scores.dtype) - scores).detach(), scores)
        else:
            scores = scores + add.to(scores.dtype)
    probs = torch.softmax(scores, dim=-1)  # W:157
    probs = dropout(probs, attn_dropout, training, site=prefix)  # W:160
    ctx = probs @ v  # W:167
    ctx = ctx.permute(0, 2, 1, 3).reshape(B, Tq, d)
    return dense(ctx, p[f"{prefix}.out_proj.kernel"], p[f"{prefix}.out_proj.bias"])  # W:174


def feed_forward(p, prefix, x, cfg, training):
    """W:200-206."""
    h = dense(x, p[f"{prefix}.fc1.kernel"], p[f"{prefix}.fc1.bias"])
    h = gelu_erf(h)
    h = dropout(h, cfg.activation_dropout, training)
    h = dense(h, p[f"{prefix}.fc2.kernel"], p[f"{prefix}.fc2.bias"])
    return dropout(h, cfg.dropout, training, site=prefix)


def encoder_layer(p, prefix, x, cfg, training):
    """W:218-236: pre-LN residual block."""
    r = x
    h = layer_norm(x, p[f"{prefix}.self_attn_layer_norm.gamma"], p[f"{prefix}.self_attn_layer_norm.beta"], cfg.layer_norm_eps)
    h = mha(p, f"{prefix}.self_attn", h, None, None, cfg.encoder_attention_heads, cfg.attention_dropout, training)
    x = r + h
    r = x
    h = layer_norm(x, p[f"{prefix}.final_layer_norm.gamma"], p[f"{prefix}.final_layer_norm.beta"], cfg.layer_norm_eps)
    return r + feed_forward(p, f"{prefix}.feed_forward", h, cfg, training)


def decoder_layer(p, prefix, x, enc, mask, cfg, training):
    """W:255-301."""
    H = cfg.decoder_attention_heads
    r = x
    h = layer_norm(x, p[f"{prefix}.self_attn_layer_norm.gamma"], p[f"{prefix}.self_attn_layer_norm.beta"], cfg.layer_norm_eps)
    x = r + mha(p, f"{prefix}.self_attn", h, None, mask, H, cfg.attention_dropout, training)
    r = x
    h = layer_norm(x, p[f"{prefix}.encoder_attn_layer_norm.gamma"], p[f"{prefix}.encoder_attn_layer_norm.beta"], cfg.layer_norm_eps)
    x = r + mha(p, f"{prefix}.encoder_attn", h, enc, None, H, cfg.attention_dropout, training)
    r = x
    h = layer_norm(x, p[f"{prefix}.final_layer_norm.gamma"], p[f"{prefix}.final_layer_norm.beta"], cfg.layer_norm_eps)
    return r + feed_forward(p, f"{prefix}.feed_forward", h, cfg, training)


def encoder(p, feats, cfg, training=True):
    """W:324-372.  feats [B, n_mels, T_in]."""
    x = feats.transpose(1, 2)  # W:329
    x = gelu_erf(conv1d_same(x, p["encoder.conv1.kernel"], p["encoder.conv1.bias"], 1))  # W:332-333
    x = gelu_erf(conv1d_same(x, p["encoder.conv2.kernel"], p["encoder.conv2.bias"], 2))  # W:335-336
    pe = torch.from_numpy(positional_encoding(cfg.n_ctx, cfg.d_model)).to(x.dtype)
    x = x + pe[: x.shape[1]]  # W:339
    x = dropout(x, cfg.dropout, training, site="encoder.stem")
    for i in range(cfg.encoder_layers):
        x = encoder_layer(p, f"encoder.layers.{i}", x, cfg, training)
    return layer_norm(x, p["encoder.layer_norm.gamma"], p["encoder.layer_norm.beta"], cfg.layer_norm_eps)


def decoder(p, ids, enc, cfg, training=True):
    """W:394-466."""
    x = p["decoder.embed_tokens.embeddings"][ids.long()]  # W:405
    pe = torch.from_numpy(positional_encoding(cfg.max_target_positions, cfg.d_model)).to(x.dtype)
    x = x + pe[: x.shape[1]]  # W:408
    x = dropout(x, cfg.dropout, training, site="decoder.embed")
    S = ids.shape[1]
    mask = torch.from_numpy(decoder_mask(S))[None]  # W:416-418
    for i in range(cfg.decoder_layers):
        x = decoder_layer(p, f"decoder.layers.{i}", x, enc, mask, cfg, training)
    return layer_norm(x, p["decoder.layer_norm.gamma"], p["decoder.layer_norm.beta"], cfg.layer_norm_eps)


def decoder_input_ids(labels, start_id):
    """W:559-563: pad(labels[:, :-1], left 1, start token)."""
    B = labels.shape[0]
    start = torch.full((B, 1), start_id, dtype=labels.dtype)
    return torch.cat([start, labels[:, :-1]], dim=1)


def forward_loss(p, feats, labels, cfg, training=True):
    """W:547-616 with training=True, labels given, no decoder_attention_mask: returns
    (loss, logits).  Position t sees labels[t-1] and is scored against labels[t+1]
    (double shift, W:559-563 then W:585-586); pad tokens are included in the mean."""
    dtype = p["lm_head.kernel"].dtype
    feats = feats.to(dtype)
    enc = encoder(p, feats, cfg, training)
    dec_in = decoder_input_ids(labels, cfg.decoder_start_token_id)
    h = decoder(p, dec_in, enc, cfg, training)
    logits = h @ p["lm_head.kernel"]  # W:579
    shift_labels = labels[:, 1:].long()
    shift_logits = logits[:, :-1, :]
    loss = F.cross_entropy(shift_logits.reshape(-1, logits.shape[-1]), shift_labels.reshape(-1), reduction="mean")  # W:589-600
    return loss, logits


def loss_and_grads(p, feats, labels, cfg, training=True):
    """tape.gradient(loss, model.trainable_variables), W:826-833."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, _ = forward_loss(leaves, feats, labels, cfg, training)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads


# --------------------------------------------------------------------------------------
# Keras OptimizerV2 Adam (TF 2.10; not in /root/reference).  W:901: lr 1e-4, eps 1e-7.
#   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2
#   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) ; theta <- theta - lr_t * m / (sqrt(v) + eps)
# eps_mode "torch" gives torch.optim.Adam's placement (eps added to sqrt(v_hat)).
# --------------------------------------------------------------------------------------
@dataclass
class AdamState:
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)
    t: int = 0


def adam_step(params, grads, state: AdamState, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-7,
              eps_mode="tf", weight_decay=0.0):
    state.t += 1
    t = state.t
    for k, w in params.items():
        g = grads[k].to(w.dtype)
        if k not in state.m:
            state.m[k] = torch.zeros_like(w)
            state.v[k] = torch.zeros_like(w)
        m = state.m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        v = state.v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        if weight_decay:
            w.mul_(1.0 - lr * weight_decay)
        if eps_mode == "tf":
            lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
            w.sub_(lr_t * m / (v.sqrt() + eps))
        else:
            mhat = m / (1.0 - beta1 ** t)
            vhat = v / (1.0 - beta2 ** t)
            w.sub_(lr * mhat / (vhat.sqrt() + eps))
    return params


def train_steps(cfg, params, pool_feats, pool_labels, batch_size, num_steps, lr=1e-4,
                n_replicas=1, training_dropout=False):
    """W:894-958 loop + W:819-848 step for ``n_replicas`` replicas on one host: each
    replica takes ``batch_size`` consecutive samples of the global batch; gradients are
    SUMMED across replicas and the printed loss is the SUM of per-replica mean losses
    (no division by num_replicas, W:829-836, W:848)."""
    if not training_dropout:
        cfg = make_config_like(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    state = AdamState()
    it = batches(pool_feats, pool_labels, batch_size * n_replicas)
    losses: List[float] = []
    for _ in range(num_steps):
        f, l = next(it)
        tot_loss = 0.0
        tot = None
        for r in range(n_replicas):
            fr = torch.from_numpy(np.ascontiguousarray(f[r * batch_size:(r + 1) * batch_size]))
            lr_ = torch.from_numpy(np.ascontiguousarray(l[r * batch_size:(r + 1) * batch_size]))
            if fr.shape[0] == 0:
                continue
            loss, g = loss_and_grads(params, fr, lr_, cfg)
            tot_loss += float(loss)
            tot = g if tot is None else {k: tot[k] + g[k] for k in g}
        adam_step(params, tot, state, lr=lr)
        losses.append(tot_loss)
    return losses, state


def make_config_like(cfg: WhisperConfig, **kw) -> WhisperConfig:
    import copy
    c = copy.copy(cfg)
    for k, v in kw.items():
        setattr(c, k, v)
    return c
