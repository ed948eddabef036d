"""CPU oracle for the Wav2Vec2 pre-training step.  TEST INFRASTRUCTURE ONLY.

Op-for-op restatement of ``/root/reference/speech_jobs/wav2vec2_dist.py`` ("V:") for the
path ``main`` actually runs: ``Wav2Vec2ForPreTraining`` (V:826-937) on 2 s synthetic clips
(V:1123-1153), loss = contrastive + 0.1 * (-perplexity) (V:1199-1248), local
clip_by_global_norm(1.0), gradient all-reduce, Keras Adam(lr 3e-5, eps 1e-8, clipnorm 1.0)
(V:1271-1275).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

PARITY UNPINNED: the reference holds no tests / fixtures for this path and TensorFlow is
not installed; pinned by closed-form KATs (tests/test_oracle_kat.py) and fp64 finite
differences only.  TF semantics encoded by hand, beyond those listed in whisper_oracle.py:
grouped Keras Conv1D kernel layout [k, C_in/groups, C_out] with output channel o in group
o // (C_out/groups); tf.argmin first-index tie-break; tf.nn.top_k lowest-index-first
tie-break; tf.clip_by_global_norm (g * clip / max(norm, clip)); Keras ``clipnorm``
(tf.clip_by_norm per variable, after aggregation).

Reference behaviours reproduced, not fixed: GroupNorm on EVERY conv layer with
groups = num_conv_pos_embedding_groups (V:248); the quantizer sees the PROJECTED features
(V:784); hard VQ (argmin / one-hot, no Gumbel, no straight-through) so
``quantizer.projection`` gets no gradient and the diversity term has zero gradient
(V:631-660); no cosine normalisation and no masking in the contrastive loss; the same
negative indices for every time step of a batch row, possibly containing the positive
(V:908-937); attention scores divided by sqrt(head_dim) after q·kᵀ (V:349); no final
encoder LayerNorm.  The negative indices come from TF's unseeded RNG in the reference;
here they are an INPUT of the step, drawn by ``sample_negative_indices`` from a seeded
NumPy generator with the same recipe.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .whisper_oracle import AdamState, dense, gelu_erf, layer_norm, same_pad


# --------------------------------------------------------------------------------------
# V:24-128 Wav2Vec2Config (only the fields the pre-training path reads)
# --------------------------------------------------------------------------------------
@dataclass
class Wav2Vec2Config:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    num_codevectors_per_group: int = 320
    num_codevector_groups: int = 2
    codevector_dim: int = 256
    proj_codevector_dim: int = 256
    contrastive_logits_temperature: float = 0.1
    num_negatives: int = 100
    diversity_loss_weight: float = 0.1
    hidden_dropout: float = 0.1
    activation_dropout: float = 0.1
    attention_dropout: float = 0.1


def make_config(model_size: str = "small", **overrides) -> Wav2Vec2Config:
    """V:25-86: "small" and "tiny" shrink the base model; anything else is base."""
    if model_size == "small":
        kw = dict(hidden_size=512, num_hidden_layers=6, num_attention_heads=8, intermediate_size=2048,
                  conv_dim=(256,) * 5, conv_stride=(5, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 2),
                  num_conv_pos_embeddings=64, num_conv_pos_embedding_groups=8,
                  num_codevectors_per_group=160, codevector_dim=128, proj_codevector_dim=128)
    elif model_size == "tiny":
        kw = dict(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024,
                  conv_dim=(128,) * 4, conv_stride=(5, 2, 2, 2), conv_kernel=(10, 3, 3, 2),
                  num_conv_pos_embeddings=32, num_conv_pos_embedding_groups=4,
                  num_codevectors_per_group=80, codevector_dim=64, proj_codevector_dim=64)
    else:
        kw = {}
    kw.update(overrides)
    return Wav2Vec2Config(**kw)


def feature_lengths(cfg: Wav2Vec2Config, T_in: int) -> List[int]:
    out, T = [], T_in
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        T = same_pad(T, k, s)[0]
        out.append(T)
    return out


# --------------------------------------------------------------------------------------
# parameters (names follow the reference attribute paths)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: Wav2Vec2Config) -> Dict[str, Tuple[int, ...]]:
    H, I = cfg.hidden_size, cfg.intermediate_size
    C = cfg.conv_dim[-1]
    G = cfg.num_conv_pos_embedding_groups
    s: Dict[str, Tuple[int, ...]] = {}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):  # V:229-268, conv_bias False
        s[f"feature_extractor.conv_layers.{i}.conv.kernel"] = (k, cin, c)
        s[f"feature_extractor.conv_layers.{i}.norm.gamma"] = (c,)
        s[f"feature_extractor.conv_layers.{i}.norm.beta"] = (c,)
        cin = c
    s["feature_extractor.pos_conv_embed.kernel"] = (cfg.num_conv_pos_embeddings, C // G, C)  # V:271-277
    s["feature_extractor.pos_conv_embed.bias"] = (C,)
    s["feature_extractor.layer_norm.gamma"] = (C,)
    s["feature_extractor.layer_norm.beta"] = (C,)
    s["feature_projection.kernel"] = (C, H)  # V:754
    s["feature_projection.bias"] = (H,)
    s["feature_projection_layer_norm.gamma"] = (H,)
    s["feature_projection_layer_norm.beta"] = (H,)
    for i in range(cfg.num_hidden_layers):  # V:401-415
        p = f"encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{p}.attention.{n}.kernel"] = (H, H)
            s[f"{p}.attention.{n}.bias"] = (H,)
        s[f"{p}.attention_layer_norm.gamma"] = (H,)
        s[f"{p}.attention_layer_norm.beta"] = (H,)
        s[f"{p}.feed_forward.intermediate_dense.kernel"] = (H, I)
        s[f"{p}.feed_forward.intermediate_dense.bias"] = (I,)
        s[f"{p}.feed_forward.output_dense.kernel"] = (I, H)
        s[f"{p}.feed_forward.output_dense.bias"] = (H,)
        s[f"{p}.feed_forward_layer_norm.gamma"] = (H,)
        s[f"{p}.feed_forward_layer_norm.beta"] = (H,)
    gd = cfg.codevector_dim // cfg.num_codevector_groups
    s["quantizer.codevectors"] = (cfg.num_codevector_groups, cfg.num_codevectors_per_group, gd)  # V:570-576
    s["quantizer.projection.kernel"] = (H, cfg.codevector_dim)  # V:579
    s["quantizer.projection.bias"] = (cfg.codevector_dim,)
    for n, fan_in in (("project_hid", H), ("project_q", cfg.codevector_dim)):  # V:550-561, V:765-766
        s[f"{n}.dense.kernel"] = (fan_in, cfg.proj_codevector_dim)
        s[f"{n}.dense.bias"] = (cfg.proj_codevector_dim,)
        s[f"{n}.layer_norm.gamma"] = (cfg.proj_codevector_dim,)
        s[f"{n}.layer_norm.beta"] = (cfg.proj_codevector_dim,)
    return s


def param_count(cfg: Wav2Vec2Config) -> int:
    return sum(int(np.prod(v)) for v in param_shapes(cfg).values())


def init_params(cfg: Wav2Vec2Config, seed: int = 1234, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Keras defaults: glorot-uniform kernels (Conv1D fans = k*C_in/groups, k*C_out/groups... Keras
    uses the kernel tensor's own shape: fan_in = k * shape[1], fan_out = k * shape[2]), zero biases,
    norm gamma 1 / beta 0, codevectors ~ N(0, 1) (tf.random.normal, V:571)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".kernel"):
            if len(shape) == 3:
                fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, size=shape)
        elif name == "quantizer.codevectors":
            a = rng.standard_normal(shape)
        elif name.endswith(".gamma"):
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        out[name] = torch.from_numpy(a.astype(np.float32)).to(dtype)
    return out


# --------------------------------------------------------------------------------------
# data (V:1123-1153) and negative sampling (V:908-937)
# --------------------------------------------------------------------------------------
def create_dummy_pool(seed: int = 1234, num_samples: int = 50, length: int = 32000) -> np.ndarray:
    """50 x tf.random.normal([32000]) fp32; labels are a constant 0.0 and unused."""
    return np.random.default_rng(seed).standard_normal((num_samples, length)).astype(np.float32)


def batches(pool: np.ndarray, global_batch: int):
    """dataset.batch(global_batch, drop_remainder=True).repeat() (V:1147-1151)."""
    n = pool.shape[0] // global_batch * global_batch
    while True:
        for s in range(0, n, global_batch):
            yield pool[s:s + global_batch]


def sample_negative_indices(rng: np.random.Generator, batch_size: int, T: int, num_negatives: int = 100) -> np.ndarray:
    """V:908-937 -> [B, num_negatives] int32 (the reference tiles this over the time axis):
    K = max(min(num_negatives, T-1), 1) positions with the smallest uniform-int draws (top_k of the
    negated draws, ties to the lowest index), repeated / truncated to num_negatives."""
    K = max(min(num_negatives, T - 1), 1)
    r = rng.integers(0, T, size=(batch_size, T))
    order = np.argsort(r, axis=1, kind="stable")[:, :K]
    if K < num_negatives:
        reps = -(-num_negatives // K)
        order = np.tile(order, (1, reps))
    return order[:, :num_negatives].astype(np.int32)


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def conv1d_same(x, kernel, bias, stride, groups=1):
    """Keras Conv1D(padding="same", groups=groups) on channels-last x; kernel [k, Cin/groups, Cout]."""
    k = kernel.shape[0]
    _, left, right = same_pad(x.shape[1], k, stride)
    xt = F.pad(x.transpose(1, 2), (left, right))
    y = F.conv1d(xt, kernel.permute(2, 1, 0), bias=None, stride=stride, groups=groups).transpose(1, 2)
    return y if bias is None else y + bias


def group_norm(x, gamma, beta, groups, eps=1e-5):
    """V:140-196: statistics over (time, C/groups) per (batch, group), biased variance,
    contiguous channel groups, per-channel affine."""
    B, T, C = x.shape
    xg = x.reshape(B, T, groups, C // groups)
    mu = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mu) ** 2).mean(dim=(1, 3), keepdim=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(B, T, C)
    return gamma * xn + beta


# When set (oracle.dropout.HostDropout with the Wav2Vec2 site table), the reference's Dropout layers (V:296, V:359,
# V:393, V:396, V:431, V:560, V:779) are applied with the counter-based masks the HIP kernels draw; None = rates 0.
DROPOUT_PROVIDER = None


def _drop(x, rate, site):
    if DROPOUT_PROVIDER is None or rate <= 0.0:
        return x
    keep, scale = DROPOUT_PROVIDER.mask(site, tuple(x.shape), rate)
    return x * keep.to(x.dtype) * scale


def feature_extractor(p, audio, cfg):
    """V:283-298."""
    x = audio.unsqueeze(-1)
    G = cfg.num_conv_pos_embedding_groups
    for i, s in enumerate(cfg.conv_stride):
        pre = f"feature_extractor.conv_layers.{i}"
        x = conv1d_same(x, p[f"{pre}.conv.kernel"], None, s)
        x = gelu_erf(group_norm(x, p[f"{pre}.norm.gamma"], p[f"{pre}.norm.beta"], G))
    pos = conv1d_same(x, p["feature_extractor.pos_conv_embed.kernel"], p["feature_extractor.pos_conv_embed.bias"], 1, groups=G)
    x = x + pos
    x = layer_norm(x, p["feature_extractor.layer_norm.gamma"], p["feature_extractor.layer_norm.beta"], cfg.layer_norm_eps)
    return _drop(x, cfg.hidden_dropout, "feature_extractor")  # V:296


def attention(p, prefix, x, num_heads, attn_dropout=0.0):
    """V:333-376 (attention_mask None in training)."""
    B, T, H = x.shape
    hd = H // num_heads
    q = dense(x, p[f"{prefix}.q_proj.kernel"], p[f"{prefix}.q_proj.bias"])
    k = dense(x, p[f"{prefix}.k_proj.kernel"], p[f"{prefix}.k_proj.bias"])
    v = dense(x, p[f"{prefix}.v_proj.kernel"], p[f"{prefix}.v_proj.bias"])

    def split(t):
        return t.reshape(B, T, num_heads, hd).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)  # V:348-349
    ctx = _drop(torch.softmax(s, dim=-1), attn_dropout, prefix) @ v  # V:359
    ctx = ctx.permute(0, 2, 1, 3).reshape(B, T, H)
    return dense(ctx, p[f"{prefix}.out_proj.kernel"], p[f"{prefix}.out_proj.bias"])


def encoder_layer(p, prefix, x, cfg):
    """V:419-439 (do_stable_layer_norm=True branch)."""
    h = layer_norm(x, p[f"{prefix}.attention_layer_norm.gamma"], p[f"{prefix}.attention_layer_norm.beta"], cfg.layer_norm_eps)
    x = x + _drop(attention(p, f"{prefix}.attention", h, cfg.num_attention_heads, cfg.attention_dropout),
                  cfg.hidden_dropout, f"{prefix}.attention_output")  # V:431
    h = layer_norm(x, p[f"{prefix}.feed_forward_layer_norm.gamma"], p[f"{prefix}.feed_forward_layer_norm.beta"], cfg.layer_norm_eps)
    h = gelu_erf(dense(h, p[f"{prefix}.feed_forward.intermediate_dense.kernel"], p[f"{prefix}.feed_forward.intermediate_dense.bias"]))
    h = _drop(h, cfg.activation_dropout, f"{prefix}.intermediate")  # V:393
    h = dense(h, p[f"{prefix}.feed_forward.output_dense.kernel"], p[f"{prefix}.feed_forward.output_dense.bias"])
    return x + _drop(h, cfg.hidden_dropout, f"{prefix}.output")  # V:396


def quantizer(p, hidden, cfg, force_idx=None, clip_probs=True):
    """V:581-667 -> (quantized [B,T,cd], indices [B,T,G], perplexity, distances).
    ``clip_probs=False`` is the older form of speech_jobs/whisper_single.py:577-579 (no clip_by_value of the
    mean one-hots before the log).
    ``force_idx`` (tests of the bf16 path only) replaces the argmin by given code indices, so
    that a near-tie resolved differently under bf16 rounding does not mask everything else."""
    B, T, _ = hidden.shape
    G = cfg.num_codevector_groups
    gd = cfg.codevector_dim // G
    h = dense(hidden, p["quantizer.projection.kernel"], p["quantizer.projection.bias"]).reshape(B, T, G, gd)
    quant, idxs, probs, dists = [], [], [], []
    for g in range(G):
        cv = p["quantizer.codevectors"][g]
        dist = ((h[:, :, g, None, :] - cv[None, None]) ** 2).sum(-1)  # V:623-627
        dists.append(dist.detach())
        idx = torch.argmin(dist, dim=-1)  # first index on ties, as tf.argmin
        if force_idx is not None:
            idx = force_idx[..., g].long()
        enc = F.one_hot(idx, cfg.num_codevectors_per_group).to(h.dtype)
        quant.append(enc @ cv)  # V:638: gradient reaches the codebook only
        idxs.append(idx)
        probs.append(enc.mean(dim=(0, 1)))
    avg = torch.stack(probs)
    if clip_probs:
        avg = avg.clamp(1e-10, 1.0)  # V:653-657
    perplexity = torch.exp(-(avg * torch.log(avg + 1e-10)).sum(-1)).mean()
    return torch.cat(quant, dim=-1), torch.stack(idxs, dim=-1), perplexity, torch.stack(dists, dim=2)


def projection_head(p, name, x, cfg):
    """V:557-561."""
    h = dense(x, p[f"{name}.dense.kernel"], p[f"{name}.dense.bias"])
    h = layer_norm(h, p[f"{name}.layer_norm.gamma"], p[f"{name}.layer_norm.beta"], cfg.layer_norm_eps)
    return _drop(h, cfg.hidden_dropout, name)  # V:560


def forward(p, audio, cfg, force_idx=None, clip_probs=True):
    """V:768-825 + V:841-863 with training=True -> dict."""
    dtype = p["feature_projection.kernel"].dtype
    feats = feature_extractor(p, audio.to(dtype), cfg)
    h = dense(feats, p["feature_projection.kernel"], p["feature_projection.bias"])
    h = layer_norm(h, p["feature_projection_layer_norm.gamma"], p["feature_projection_layer_norm.beta"], cfg.layer_norm_eps)
    h = _drop(h, cfg.hidden_dropout, "feature_projection")  # V:779
    quantized, idx, perplexity, dists = quantizer(p, h, cfg, force_idx, clip_probs)  # on the projected features (V:784)
    x = h
    for i in range(cfg.num_hidden_layers):
        x = encoder_layer(p, f"encoder.layers.{i}", x, cfg)
    return {"projected_states": projection_head(p, "project_hid", x, cfg),
            "projected_quantized_features": projection_head(p, "project_q", quantized, cfg),
            "codevector_perplexity": perplexity, "code_indices": idx, "code_distances": dists,
            "extract_features": feats}


def contrastive_loss(h, q, neg_idx, temperature):
    """V:866-899.  h, q [B,T,D]; neg_idx [B, num_negatives] (V: the same indices for every t) or
    [B, T, num_negatives] (whisper_single.py:745-839: a row of indices per time step)."""
    pos = (h * q).sum(-1) / temperature
    if neg_idx.dim() == 3:
        B = q.shape[0]
        neg_q = q[torch.arange(B)[:, None, None], neg_idx.long()]  # [B, T, N, D]  (tf.gather batch_dims=1)
        neg = (h.unsqueeze(2) * neg_q).sum(-1) / temperature
    else:
        neg_q = q[torch.arange(q.shape[0])[:, None], neg_idx.long()]  # [B, N, D]
        neg = torch.einsum("btd,bnd->btn", h, neg_q) / temperature
    logits = torch.cat([pos.unsqueeze(-1), neg], dim=-1)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]),
                           torch.zeros(logits.shape[0] * logits.shape[1], dtype=torch.long), reduction="mean")
    return logits, loss


def step_loss(p, audio, neg_idx, cfg, num_replicas=1, force_idx=None):
    """V:1199-1231: contrastive + 0.1 * (-perplexity), NaN -> 0, / num_replicas."""
    out = forward(p, audio, cfg, force_idx)
    _, cl = contrastive_loss(out["projected_states"], out["projected_quantized_features"], neg_idx,
                             cfg.contrastive_logits_temperature)
    loss = cl + cfg.diversity_loss_weight * (-out["codevector_perplexity"])
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss / num_replicas, out


def loss_and_grads(p, audio, neg_idx, cfg, num_replicas=1, force_idx=None):
    """tape.gradient with None -> zeros (V:1234-1240)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, out = step_loss(leaves, audio, neg_idx, cfg, num_replicas, force_idx)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads, out


def clip_by_global_norm(grads, clip=1.0):
    """tf.clip_by_global_norm (V:1243)."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    scale = clip / max(float(norm), clip)
    return {k: g * scale for k, g in grads.items()}, float(norm)


def clip_by_norm_each(grads, clip=1.0):
    """Keras clipnorm (V:1274): tf.clip_by_norm per variable = g * clip / max(||g||, clip)."""
    out = {}
    for k, g in grads.items():
        n = float(torch.sqrt((g.double() ** 2).sum()))
        out[k] = g * (clip / max(n, clip))
    return out


def adam_step(params, grads, state: AdamState, lr=3e-5, beta1=0.9, beta2=0.999, eps=1e-8):
    from .whisper_oracle import adam_step as _adam
    return _adam(params, grads, state, lr=lr, beta1=beta1, beta2=beta2, eps=eps, eps_mode="tf")


def train_steps(cfg, params, pool, batch_size, num_steps, seed=1234, n_replicas=1, lr=3e-5, code_trace=None):
    """V:1263-1376 loop for ``n_replicas`` replicas on one host.  Returns the printed losses
    (sum over replicas of loss / n_replicas) and the Adam state.  ``code_trace``: a list that receives every step's
    quantiser choices [B, T, G] (teacher-forcing fixture of the bf16 golden test)."""
    rng = np.random.default_rng(seed)
    T = feature_lengths(cfg, pool.shape[1])[-1]
    it = batches(pool, batch_size * n_replicas)
    state = AdamState()
    losses = []
    for _ in range(num_steps):
        a = next(it)
        neg = sample_negative_indices(rng, batch_size * n_replicas, T, cfg.num_negatives)
        tot, agg = 0.0, None
        for r in range(n_replicas):
            sl = slice(r * batch_size, (r + 1) * batch_size)
            loss, g, out = loss_and_grads(params, torch.from_numpy(a[sl]), torch.from_numpy(neg[sl]), cfg, n_replicas)
            if code_trace is not None:
                code_trace.append(out["code_indices"].to(torch.int32).numpy().copy())
            g, _ = clip_by_global_norm(g, 1.0)  # local, before the all-reduce (V:1243)
            tot += float(loss)
            agg = g if agg is None else {k: agg[k] + g[k] for k in g}
        agg = clip_by_norm_each(agg, 1.0)  # Keras clipnorm after aggregation
        adam_step(params, agg, state, lr=lr)
        losses.append(tot)
    return losses, state


# --------------------------------------------------------------------------------------
# BASELINE config #1 as the file is named: speech_jobs/whisper_single.py ("S:") is a single-device
# Wav2Vec2-base pre-training job.  Same model as V: base; what differs on the step:
#   S:1094-1111  5 s clips (80000 samples -> T = 250), dataset.batch(B).repeat() WITHOUT drop_remainder
#   S:789-839    negatives: a shuffle of range(T) (tf.random.shuffle(seed=42)), rolled by t + 1 for time step
#                t, first num_negatives columns -> a different row of indices for every t
#   S:577-579    perplexity without the clip_by_value of V:653
#   S:1143-1180  loss = contrastive + 0.1 * (-perplexity); no NaN guard, no / replicas, no clipping
#   S:1189       Adam(3e-5), default epsilon 1e-7, no clipnorm
# --------------------------------------------------------------------------------------
def sample_negative_indices_roll(rng: np.random.Generator, T: int, num_negatives: int = 100) -> np.ndarray:
    """S:789-839 -> [T, num_negatives] int32: row t = roll(perm, t + 1)[:num_negatives], perm a shuffle of
    range(T), the same for every batch row.  TF's shuffle stream (op seed 42, global seed unset) cannot be
    reproduced here; the permutation comes from ``rng`` and is an input of the step, like V:'s indices."""
    perm = rng.permutation(T).astype(np.int32)
    n = min(num_negatives, T)  # shift[:, :num_negatives] of a T-wide tensor
    return np.stack([np.roll(perm, t + 1)[:n] for t in range(T)])


def step_loss_single(p, audio, neg_idx_t, cfg, force_idx=None):
    """S:1143-1175.  ``neg_idx_t`` [T, N] (tiled over the batch, S:803)."""
    out = forward(p, audio, cfg, force_idx, clip_probs=False)
    B = audio.shape[0]
    neg = neg_idx_t[None].expand(B, -1, -1)
    _, cl = contrastive_loss(out["projected_states"], out["projected_quantized_features"], neg,
                             cfg.contrastive_logits_temperature)
    return cl + cfg.diversity_loss_weight * (-out["codevector_perplexity"]), out


def loss_and_grads_single(p, audio, neg_idx_t, cfg, force_idx=None):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, out = step_loss_single(leaves, audio, neg_idx_t, cfg, force_idx)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads, out


def batches_keep_remainder(pool: np.ndarray, batch: int):
    """S:1111: dataset.batch(batch_size).repeat() - the last batch of a pass over the 50 clips is short."""
    n = pool.shape[0]
    while True:
        for s_ in range(0, n, batch):
            yield pool[s_:s_ + batch]


def train_steps_single(cfg, params, pool, batch_size, num_steps, seed=42, lr=3e-5):
    """S:1183-1263 loop with S:1143-1180's step: one permutation draw per step, Adam eps 1e-7, nothing clipped."""
    rng = np.random.default_rng(seed)
    T = feature_lengths(cfg, pool.shape[1])[-1]
    it = batches_keep_remainder(pool, batch_size)
    state = AdamState()
    losses = []
    for _ in range(num_steps):
        a = next(it)
        neg = torch.from_numpy(sample_negative_indices_roll(rng, T, cfg.num_negatives))
        loss, g, _ = loss_and_grads_single(params, torch.from_numpy(a), neg, cfg)
        adam_step(params, g, state, lr=lr, eps=1e-7)
        losses.append(float(loss))
    return losses, state


def train_steps_stable(cfg, params, pool, batch_per_replica, world, num_steps, seed=42, lr=3e-5):
    """stable_jobs/wav2vec2_dist.py ("T:") - the S: model and step under MultiWorkerMirroredStrategy (T:1143-1190, T:1193-1268):
    ``create_dummy_dataset(GLOBAL_BATCH).batch().repeat()`` keeps the short last batch (T:1094-1111); replica r takes rows
    [r*B, (r+1)*B) of each global batch (possibly fewer, possibly none); each replica's loss is the mean over ITS rows, the
    gradients are SUMMED over replicas (no 1/N, T:1183), Adam eps 1e-7 with nothing clipped (T:1200); the printed loss is
    the SUM of the replica losses (T:1190).  One permutation draw per replica per step, in rank order.  A replica with no
    rows contributes zero gradient and zero loss (the reference's arithmetic has no defined value there: a mean over an
    empty batch)."""
    rng = np.random.default_rng(seed)
    T = feature_lengths(cfg, pool.shape[1])[-1]
    it = batches_keep_remainder(pool, batch_per_replica * world)
    state = AdamState()
    losses = []
    for _ in range(num_steps):
        a = next(it)
        total, lsum = None, 0.0
        for r in range(world):
            neg = torch.from_numpy(sample_negative_indices_roll(rng, T, cfg.num_negatives))
            rows = a[r * batch_per_replica:(r + 1) * batch_per_replica]
            if rows.shape[0] == 0:
                continue
            loss, g, _ = loss_and_grads_single(params, torch.from_numpy(rows), neg, cfg)
            lsum += float(loss)
            total = g if total is None else {k: total[k] + g[k] for k in g}
        adam_step(params, total, state, lr=lr, eps=1e-7)
        losses.append(lsum)
    return losses, state

