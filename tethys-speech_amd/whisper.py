"""Whisper encoder-decoder training step on the HIP kernels (host orchestration).

Mirrors the reference's operator API for this path (speech_jobs/whisper_dist.py, "W:"):
``WhisperConfig`` (W:10-45), ``create_whisper_model(model_type)`` (W:852-890) and a model
object whose ``forward_backward(features, labels)`` does what ``model(features,
labels=..., training=True)`` + ``tape.gradient`` do at W:826-833.  The forward/backward is
written out by hand (no autograd): every tensor op is a C-ABI call from ``ops``.

Two precisions:
  "fp32"  parity mode — fp32 activations, exact-fp32 MFMA GEMMs reading the fp32 master
          weights, attention in the reference's own materialised-score shape (W:147-167).
  "bf16"  perf mode — bf16 activations and bf16 weight shadows, fused flash attention,
          fp32 accumulation, fp32 gradients/optimizer state.
Dropout (W:29-30) is off by default (parity mode: TF's RNG stream cannot be reproduced, so a step with
dropout has no parity definition against the reference).  ``enable_dropout`` turns on the
reference's training-mode sites (W:160, W:205, W:342, W:411) on the bf16 path with counter-based masks
(tmi_dropout / the fused attention kernels); the oracle fed the same masks is the checker.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops
from .blocks import Arena, KernelBlocks, _round_up


# ----------------------------------------------------------------------------- config
# dropout site ids (seed = f(base, step, site), KernelBlocks._site_seed; restated in oracle/dropout.py)
SITE_ENC_STEM, SITE_DEC_EMBED = 1, 2
SITE_ENC_ATTN, SITE_ENC_FFN, SITE_DEC_SELF, SITE_DEC_CROSS, SITE_DEC_FFN = 100, 200, 300, 400, 500


_XENT_EXACT = os.environ.get("TMI_XENT_EXACT_TARGET", "1") != "0"


@dataclass
class WhisperConfig:  # W:10-45
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


_SIZES = {  # W:859-886
    "tiny": dict(d_model=384, encoder_layers=4, encoder_attention_heads=6, decoder_layers=4,
                 decoder_attention_heads=6, d_ff=1536),
    "base": dict(d_model=512, encoder_layers=6, encoder_attention_heads=8, decoder_layers=6,
                 decoder_attention_heads=8, d_ff=2048),
    "small": dict(),
    "medium": dict(d_model=1024, encoder_layers=24, encoder_attention_heads=16, decoder_layers=24,
                   decoder_attention_heads=16, d_ff=4096),
    "large": dict(d_model=1280, encoder_layers=32, encoder_attention_heads=20, decoder_layers=32,
                  decoder_attention_heads=20, d_ff=5120),
}


def make_config(model_type: str = "small", **overrides) -> WhisperConfig:
    cfg = WhisperConfig()
    for k, v in _SIZES.get(model_type, {}).items():
        setattr(cfg, k, v)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def same_pad(T: int, k: int, s: int) -> Tuple[int, int, int]:
    """TensorFlow "SAME": -> (T_out, pad_left, pad_right)."""
    out = -(-T // s)
    total = max((out - 1) * s + k - T, 0)
    return out, total // 2, total - total // 2


def positional_encoding(max_len: int, d_model: int) -> np.ndarray:
    """W:49-69."""
    pe = np.zeros((max_len, d_model))
    position = np.arange(0, max_len)[:, np.newaxis]
    div_term = np.exp(np.arange(0, d_model, 2) * -(np.log(10000.0) / d_model))
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe.astype(np.float32)


# ----------------------------------------------------------------------------- parameters
class ParamArena(Arena):
    """All trainable parameters in ONE flat fp32 buffer (plus same-shaped grad / Adam m / v
    buffers), in forward order so that backward fills the gradient arena from its end
    towards its start: all-reduce buckets are contiguous slices, ready in reverse order.

    Storage tensors fuse the reference's separate q/k/v Dense kernels column-wise
    ([d, 3d] for self-attention, [d, 2d] k|v for cross-attention); ``ref_views`` exposes
    them under the reference's own variable paths as strided views.
    """

    def __init__(self, cfg: WhisperConfig, device):
        self.cfg = cfg
        d, ff, V = cfg.d_model, cfg.d_ff, cfg.vocab_size
        spec: List[Tuple[str, Tuple[int, ...]]] = []

        def ln(p):
            spec.append((f"{p}.gamma", (d,)))
            spec.append((f"{p}.beta", (d,)))

        def ffn(p):
            spec.extend([(f"{p}.fc1.kernel", (d, ff)), (f"{p}.fc1.bias", (ff,)),
                         (f"{p}.fc2.kernel", (ff, d)), (f"{p}.fc2.bias", (d,))])

        spec.append(("encoder.conv1.kernel", (3, cfg.n_mels, d)))
        spec.append(("encoder.conv1.bias", (d,)))
        spec.append(("encoder.conv2.kernel", (3, d, d)))
        spec.append(("encoder.conv2.bias", (d,)))
        for i in range(cfg.encoder_layers):
            p = f"encoder.layers.{i}"
            ln(f"{p}.self_attn_layer_norm")
            spec.extend([(f"{p}.self_attn.qkv.kernel", (d, 3 * d)), (f"{p}.self_attn.qkv.bias", (3 * d,)),
                         (f"{p}.self_attn.out_proj.kernel", (d, d)), (f"{p}.self_attn.out_proj.bias", (d,))])
            ln(f"{p}.final_layer_norm")
            ffn(f"{p}.feed_forward")
        ln("encoder.layer_norm")
        spec.append(("decoder.embed_tokens.embeddings", (V, d)))
        # W:122-123 projects the encoder output to k/v in every decoder layer: the L kernels are stored
        # side by side as one [d, L*2d] matrix, so those projections (forward, dgrad, wgrad) are one GEMM
        # each per step instead of L (reference variables = column slices, see Arena.ref_views)
        if cfg.decoder_layers:
            spec.extend([("decoder.cross_kv.kernel", (d, cfg.decoder_layers * 2 * d)),
                         ("decoder.cross_kv.bias", (cfg.decoder_layers * 2 * d,))])
        for i in range(cfg.decoder_layers):
            p = f"decoder.layers.{i}"
            ln(f"{p}.self_attn_layer_norm")
            spec.extend([(f"{p}.self_attn.qkv.kernel", (d, 3 * d)), (f"{p}.self_attn.qkv.bias", (3 * d,)),
                         (f"{p}.self_attn.out_proj.kernel", (d, d)), (f"{p}.self_attn.out_proj.bias", (d,))])
            ln(f"{p}.encoder_attn_layer_norm")
            spec.extend([(f"{p}.encoder_attn.q_proj.kernel", (d, d)), (f"{p}.encoder_attn.q_proj.bias", (d,)),
                         (f"{p}.encoder_attn.out_proj.kernel", (d, d)), (f"{p}.encoder_attn.out_proj.bias", (d,))])
            ln(f"{p}.final_layer_norm")
            ffn(f"{p}.feed_forward")
        ln("decoder.layer_norm")
        # the LM head is STORED with its vocab axis padded to a multiple of 64 (zero columns that
        # stay zero under Adam: g = 0 => m = v = 0 => no update), so logits / dlogits rows are
        # 16-byte aligned and the vocab-long reductions are whole K tiles; logical shape (d, V)
        self.v_pad = _round_up(V, 64)
        spec.append(("lm_head.kernel", (d, self.v_pad)))
        self.logical: Dict[str, Tuple[int, ...]] = {"lm_head.kernel": (d, V)}

        super().__init__(spec, device, logical=self.logical, fuse_width=d)


# ----------------------------------------------------------------------------- model
class WhisperForConditionalGeneration(KernelBlocks):
    """W:536-616 (training path only).  Holds parameters, bf16 shadows and all activation
    workspaces; sized lazily for a batch size on first use."""

    def __init__(self, config: WhisperConfig, device="cuda:0", precision: str = "bf16", seed: int = 1234):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        ops.lib()  # fail loudly now if the HIP library is missing
        self.config = config
        self.device = torch.device(device)
        self.precision = precision
        self.dtype = torch.float32 if precision == "fp32" else torch.bfloat16
        self.hidden = config.d_model
        self.layer_norm_eps = config.layer_norm_eps
        self.arena = ParamArena(config, self.device)
        self.arena.init_keras_defaults(seed)
        d = config.d_model
        if d % config.encoder_attention_heads or d % config.decoder_attention_heads:
            raise ValueError("d_model must divide by the head counts")
        if precision == "bf16" and (d // config.encoder_attention_heads != 64 or d // config.decoder_attention_heads != 64):
            raise ValueError("the fused attention kernel is built for head_dim 64")
        self.pe_enc = torch.from_numpy(positional_encoding(config.n_ctx, d)).to(self.device)
        self.pe_dec = torch.from_numpy(positional_encoding(config.max_target_positions, d)).to(self.device)
        self.pe_enc_t = self.pe_enc.to(self.dtype)
        self._ws_key = None
        self.ws: Dict[str, torch.Tensor] = {}
        self._ws_sets: Dict[tuple, Dict[str, torch.Tensor]] = {}
        # bf16 mirror of the whole parameter arena (perf mode), same flat indexing: every
        # kernel in its natural Keras [in, out] layout.  Forward reads it k-strided (hardware
        # transposed LDS reads), dgrad reads it k-contiguous.  Written by the Adam kernel itself.
        self.mirror = None
        if precision == "bf16":
            self.mirror = torch.zeros(self.arena.numel, dtype=torch.bfloat16, device=self.device)
            self.refresh_shadows()
        # weight / bias gradients on a second stream beside the dgrad chain (blocks.KernelBlocks)
        self.enable_wgrad_stream(os.environ.get("TMI_WGRAD_STREAM", "1") != "0")

    # -- weights ---------------------------------------------------------------------



    # -- workspaces --------------------------------------------------------------------

    def _prepare(self, B: int, T_in: int, S: int):
        key = (B, T_in, S)
        if self._ws_key == key:
            return
        # one workspace set per batch shape, kept alive across shape changes: a captured HIP graph has the
        # addresses of the set it was captured with baked in (a short final batch must not free them), and the
        # zero pad rows of the conv buffers are an invariant of each set
        self.ws = self._ws_sets.setdefault((key, self._drop_p > 0.0), {})
        while len(self._ws_sets) > 4:  # (a holder of an evicted set, e.g. GraphedTrainStep, keeps it alive itself)
            self._ws_sets.pop(next(k for k in self._ws_sets if self._ws_sets[k] is not self.ws))
        self._ws_key = key
        cfg = self.config
        if S > cfg.max_target_positions:
            raise ValueError("target length exceeds max_target_positions")
        self.T1, self.pl1, self.pr1 = same_pad(T_in, 3, 1)
        self.T, self.pl2, self.pr2 = same_pad(self.T1, 3, 2)
        if self.T > cfg.n_ctx:
            raise ValueError("encoder length exceeds n_ctx")
        d, ff = cfg.d_model, cfg.d_ff
        self.Tp0 = T_in + self.pl1 + self.pr1
        self.Tp1 = self.T1 + self.pl2 + self.pr2
        slack = 2  # rows of zero slack so the padded-K tail reads of the last window stay in bounds
        f32 = torch.float32
        z = dict(zero=True)
        self._buf("xp0", (B, self.Tp0 + slack, cfg.n_mels), **z)
        self._buf("h1pad", (B, self.Tp1 + slack, d), **z)
        self._buf("u1pad", (B, self.Tp1 + slack, d), **z)
        self._buf("dh1pad", (B, self.Tp1 + slack, d), **z)
        self._buf("u2", (B, self.T, d))
        self._buf("du2pad", (B, self.T + 1, d), **z)
        # conv1's reduction length 3*n_mels (240) is not a multiple of the GEMM's 64-deep K tile: the bf16
        # path multiplies against a copy of the kernel padded with zero rows, so the tile-aligned kernels
        # apply; the extra A columns are the next frames of xp0 (finite, inside the slack) times zero
        self.K1p = -(-3 * cfg.n_mels // 64) * 64
        if self.precision == "bf16" and self.K1p != 3 * cfg.n_mels and self.K1p - 3 * cfg.n_mels <= slack * cfg.n_mels:
            self._buf("w1pad", (self.K1p, d), **z)
        else:
            self.ws.pop("w1pad", None)
        R, Rd = B * self.T, B * S
        # decoder: the operands of the six weight gradients per layer live at a constant layer stride (one allocation per
        # kind) - the saved activations xn1 / ctx / xn2 / ctxc / xn3 / g and the gradients dqkv / dyos / dqc / dyoc / dU /
        # dyf each Dense layer receives - so that ONE batched GEMM per kind computes it for all layers after the decoder's
        # backward loop (KernelBlocks._wgrad_batched; the encoder's layers are chip-sized GEMMs and keep their own launches)
        for n, shp in (("xn1", (Rd, d)), ("ctx", (Rd, d)), ("xn2", (Rd, d)), ("ctxc", (Rd, d)), ("xn3", (Rd, d)), ("g", (Rd, ff)),
                       ("dqkv", (Rd, 3 * d)), ("dyos", (Rd, d)), ("dqc", (Rd, d)), ("dyoc", (Rd, d)), ("dU", (Rd, ff)),
                       ("dyf", (Rd, d))):
            self._buf_layers("dec", cfg.decoder_layers, n, shp)
        for side, L, rows in (("enc", cfg.encoder_layers, R), ("dec", cfg.decoder_layers, Rd)):
            for i in range(L):
                p = f"{side}{i}."
                self._buf(p + "x_in", (rows, d))
                if side == "enc":
                    self._buf(p + "xn1", (rows, d))
                    self._buf(p + "ctx", (rows, d))
                    self._buf(p + "xn2", (rows, d))
                    self._buf(p + "g", (rows, ff))
                self._buf(p + "qkv", (rows, 3 * d))
                self._buf(p + "x_mid", (rows, d))
                self._buf(p + "u", (rows, ff))
                for s_ in ("ln1", "ln2"):
                    self._buf(p + s_ + ".mean", (rows,), f32)
                    self._buf(p + s_ + ".rstd", (rows,), f32)
                if side == "dec":
                    self._buf(p + "x_mid2", (rows, d))
                    self._buf(p + "qc", (rows, d))
                    self._buf(p + "ln3.mean", (rows,), f32)
                    self._buf(p + "ln3.rstd", (rows,), f32)
        self._buf("enc_x", (R, d))
        self._buf("enc_out", (R, d))
        self._buf("enc_ln.mean", (R,), f32)
        self._buf("enc_ln.rstd", (R,), f32)
        self._buf("dec_x", (Rd, d))
        self._buf("dec_out", (Rd, d))
        self._buf("dec_ln.mean", (Rd,), f32)
        self._buf("dec_ln.rstd", (Rd,), f32)
        self.ldl = self.arena.v_pad
        self._buf("logits", (Rd, self.ldl), **z)
        self._buf("row_loss", (Rd,), f32)
        self._buf("loss", (1,), f32)
        # backward scratch, shared by every layer
        Rm = max(R, Rd)
        self._buf("dres_enc", (R, d))
        self._buf("dres_dec", (Rd, d))
        self._buf("d_enc_out", (R, d))
        self._buf("dtmp", (Rm, d))
        if self.precision == "bf16":
            self._buf("lmh_dx32", (Rd, d), f32)  # split-K accumulator of the LM-head dgrad
        self._buf("dctx", (Rm, d))
        self._buf("dctxc0", (B * S, d)); self._buf("dctxc1", (B * S, d))  # dO of the cross-attention backward (alternating by layer)
        if self._drop_p > 0.0:
            # masked copies of the residual-stream gradient (dropout mode): two buffers used by alternate layers, so a
            # layer's mask pass does not have to wait for the previous layer's weight gradient, which still reads its copy
            self._buf("dyd0", (Rm, d))
            self._buf("dyd1", (Rm, d))
        self._buf("dqkv", (Rm, 3 * d))
        Lkv = max(1, cfg.decoder_layers) * 2 * d
        self._buf("kvc_all", (R, Lkv))   # cross-attention k/v of every decoder layer, side by side
        self._buf("dkv_all", (R, Lkv))
        self._buf("dU", (Rm, ff))
        He, Hd = cfg.encoder_attention_heads, cfg.decoder_attention_heads
        if self.precision == "bf16":
            for i in range(cfg.encoder_layers):
                self._buf(f"enc{i}.stats", (B, He, self.T, 2), f32)
            for i in range(cfg.decoder_layers):
                self._buf(f"dec{i}.stats", (B, Hd, S, 2), f32)
                self._buf(f"dec{i}.statsc", (B, Hd, S, 2), f32)
            self._buf("delta", (B, max(He, Hd), max(self.T, S)), f32)
        else:
            for i in range(cfg.encoder_layers):
                self._buf(f"enc{i}.P", (B, He, self.T, self.T), f32)
            for i in range(cfg.decoder_layers):
                self._buf(f"dec{i}.P", (B, Hd, S, S), f32)
                self._buf(f"dec{i}.Pc", (B, Hd, S, self.T), f32)
            self._buf("dP", (B, max(He, Hd), max(self.T, S), self.T), f32)

    # -- building blocks -----------------------------------------------------------------




    # attention: q/k/v given as (tensor2d, column offset); rows are (b, t) with Tq / Tk per batch


    def embedding_tables(self):
        """(arena offset, rows, row length) of the tf.keras.layers.Embedding tables (W:382): rows that never see a
        gradient are skipped by the optimizer (optim.Adam._update)."""
        name = "decoder.embed_tokens.embeddings"
        V, d = self.arena.shapes[name]
        return [(self.arena.offsets[name], V, d)]

    def grad_ready_names(self) -> List[str]:
        """The parameters at which backward reports "everything stored at or after this one is final", in the
        order it reports them (arena order reversed).  The data-parallel strategy launches its buckets from
        these reports, so a replica with nothing to compute (an empty slice of a short final batch) walks the
        same list: every rank then issues the same collectives in the same order."""
        cfg = self.config
        names = ["decoder.layer_norm.gamma"]
        names += [f"decoder.layers.{i}.self_attn_layer_norm.gamma" for i in reversed(range(cfg.decoder_layers))]
        if cfg.decoder_layers:
            names.append("decoder.cross_kv.kernel")
        names += ["decoder.embed_tokens.embeddings", "encoder.layer_norm.gamma"]
        names += [f"encoder.layers.{i}.self_attn_layer_norm.gamma" for i in reversed(range(cfg.encoder_layers))]
        names.append("encoder.conv1.kernel")
        return names

    def report_zero_gradients(self, grad_ready=None):
        """The gradient arena of a replica whose slice of the batch is empty: zeros, reported through
        ``grad_ready`` range by range exactly as a real backward would."""
        a = self.arena
        a.g.zero_()
        a.g_clean = False
        hi = a.numel
        for name in self.grad_ready_names():
            lo = a.offsets[name]
            if grad_ready is not None and lo < hi:
                grad_ready(lo, hi)
                hi = lo

    # -- forward -------------------------------------------------------------------------
    def late_adam_range(self):
        """Arena range whose first reader in a step is the decoder, a whole encoder forward after the step began: the
        cross-attention k/v projection, the decoder layers and the decoder's final LayerNorm (the embedding table before
        it and the LM head after it are the EARLY slices).  None when there is no such range (no decoder layers)."""
        a = self.arena
        if not self.config.decoder_layers or "decoder.cross_kv.kernel" not in a.offsets:
            return None
        # (an "enc" slice for the upper encoder layers, ahead of this one on the stream and waited for at its first layer, was
        # measured too: from layer 1 the chain waits for it, +0.3 ms; from layer 2 it is level, 8.30 vs 8.32 ms - not kept)
        return [(a.offsets["decoder.cross_kv.kernel"], a.offsets["lm_head.kernel"], "dec")]

    def early_adam_ranges(self):
        """The two variables whose gradients are final while the decoder's backward still runs (the EARLY slices):
        the LM head (last in the arena) and the decoder's embedding table."""
        a = self.arena
        e_lo = a.offsets["decoder.embed_tokens.embeddings"]
        return [(a.offsets["lm_head.kernel"], a.numel), (e_lo, e_lo + a.grad("decoder.embed_tokens.embeddings").numel())]

    def forward_backward(self, features: torch.Tensor, labels: torch.Tensor, loss_scale: float = 1.0,
                         grad_ready=None, early_update=None):
        """Pins the launch stream for the duration of the step (KernelBlocks.begin_step), then runs
        ``_forward_backward``."""
        self.begin_step()
        try:
            return self._forward_backward(features, labels, loss_scale, grad_ready, early_update)
        finally:
            self.end_step()
            self._drop_step += 1  # next step draws fresh masks

    def _forward_backward(self, features: torch.Tensor, labels: torch.Tensor, loss_scale: float = 1.0,
                         grad_ready=None, early_update=None):
        """One replica's forward + backward (W:826-833).  features [B, n_mels, T_in] fp32,
        labels [B, S] int32, both on the device.  Gradients land in ``arena.g`` (which is
        zeroed first); returns the device scalar loss (mean over B*(S-1), W:600).
        ``grad_ready(lo, hi)`` is called during backward each time the gradients of the arena
        range [lo, hi) are final, last range first (the data-parallel strategy all-reduces them
        under the rest of backward).  ``early_update(lo, hi)`` (optim.Adam.begin_early, one replica): the optimizer
        update of an arena range, called on the second stream as soon as that range's gradients are final and its
        weights have been read for the last time in this step - the LM head and the embedding table, under the
        decoder's backward chain."""
        cfg = self.config
        B, Cn, T_in = features.shape
        S = labels.shape[1]
        if Cn != cfg.n_mels or labels.shape[0] != B:
            raise ValueError("bad batch shapes")
        if features.dtype != torch.float32 or labels.dtype != torch.int32:
            raise TypeError("features must be float32 and labels int32")
        self._prepare(B, T_in, S)
        ws, a, d, ff = self.ws, self.arena, cfg.d_model, cfg.d_ff
        T, He, Hd = self.T, cfg.encoder_attention_heads, cfg.decoder_attention_heads
        scal_e, scal_d = (d // He) ** -0.5, (d // Hd) ** -0.5
        if not getattr(a, "g_clean", False):
            self._wait_late()  # (a pending late Adam slice reads the gradients this fill would overwrite)
            ops.fill_zero(a.g)  # (an optimizer step with zero_grad leaves the arena clean: no fill pass)
        a.g_clean = False
        done = [a.numel]
        expected = iter(self.grad_ready_names())

        def ready(name):
            """Everything stored at or after parameter ``name`` now has its final gradient."""
            if name != next(expected, None):
                raise RuntimeError(f"backward reported {name} out of the order grad_ready_names() promises")
            self._flush_deferred()
            lo = a.offsets[name]
            if grad_ready is not None and lo < done[0]:
                # (a consumer that acts on the range at once must first order itself after the
                # weight-gradient stream: DataParallelStrategy does so through its pre_launch hook)
                grad_ready(lo, done[0])
                done[0] = lo

        drop = self._drop_p > 0.0
        # The decoder's embedding and layer 0 up to its cross-attention query depend on the labels only: a chain of
        # ~10 decoder-sized kernels that would otherwise sit, alone on the chip, between the encoder and the decoder.
        # They run on the second stream beside the encoder's forward.
        # decoder pieces (W:394-466); ids = [start, labels[:, :-1]] (W:559-563) inside the embedding kernel
        kvc = ws["kvc_all"]
        Ld = cfg.decoder_layers

        def dec_embed():
            y = ws["dec0.x_in"] if Ld else ws["dec_x"]
            ops.embed_fwd(labels, a.param("decoder.embed_tokens.embeddings"), self.pe_dec, y, B, S, d,
                          cfg.decoder_start_token_id)
            if drop:
                self._dropout(y, y, SITE_DEC_EMBED)  # W:411

        def dec_self_block(i):
            """Layer i up to the cross-attention query: needs nothing from the encoder."""
            p, k = f"decoder.layers.{i}", f"dec{i}."
            x_in = ws[k + "x_in"]
            self._ln_fwd(x_in, p + ".self_attn_layer_norm", ws[k + "xn1"], k + "ln1")
            self._dense_fwd(ws[k + "xn1"], p + ".self_attn.qkv.kernel", ws[k + "qkv"], scale_cols=d, scale=scal_d)
            qkv = ws[k + "qkv"]
            self._attn_fwd(k + ("stats" if self.precision == "bf16" else "P"), (qkv, 0), (qkv, d), (qkv, 2 * d),
                           ws[k + "ctx"], B, Hd, S, S, 1, site=SITE_DEC_SELF + i)
            self._dense_fwd(ws[k + "ctx"], p + ".self_attn.out_proj.kernel", ws[k + "x_mid"], resid=x_in, r_ld=d)
            # cross attention (W:278-290): k/v projections of the encoder output in every layer
            self._ln_fwd(ws[k + "x_mid"], p + ".encoder_attn_layer_norm", ws[k + "xn2"], k + "ln2")
            self._dense_fwd(ws[k + "xn2"], p + ".encoder_attn.q_proj.kernel", ws[k + "qc"], scale_cols=d, scale=scal_d)

        def dec_cross_ffn(i):
            p, k = f"decoder.layers.{i}", f"dec{i}."
            self._attn_fwd(k + ("statsc" if self.precision == "bf16" else "Pc"), (ws[k + "qc"], 0),
                           (kvc, 2 * i * d), (kvc, (2 * i + 1) * d), ws[k + "ctxc"], B, Hd, S, T, 0, site=SITE_DEC_CROSS + i)
            self._dense_fwd(ws[k + "ctxc"], p + ".encoder_attn.out_proj.kernel", ws[k + "x_mid2"],
                            resid=ws[k + "x_mid"], r_ld=d)
            self._ln_fwd(ws[k + "x_mid2"], p + ".final_layer_norm", ws[k + "xn3"], k + "ln3")
            self._dense_fwd(ws[k + "xn3"], p + ".feed_forward.fc1.kernel", ws[k + "g"], act=1, aux_out=ws[k + "u"])
            nxt = ws[f"dec{i + 1}.x_in"] if i + 1 < Ld else ws["dec_x"]
            self._dense_fwd(ws[k + "g"], p + ".feed_forward.fc2.kernel", nxt, resid=ws[k + "x_mid2"], r_ld=d,
                            **self._drop_epi(SITE_DEC_FFN + i))

        early_dec = self._side is not None and Ld > 0 and self._main is not None and os.environ.get("TMI_DEC_EARLY", "1") != "0"
        early_ev = None
        if early_dec:
            self._run_on_side(lambda: (dec_embed(), dec_self_block(0)), labels)
            early_ev = self._pop_side_reads(labels)
        # ---- encoder stem (W:329-339)
        xp0, h1pad, u1pad = ws["xp0"], ws["h1pad"], ws["u1pad"]
        ops.feat_to_channels_last(features, xp0, B, Cn, T_in, self.pl1, self.pr1 + (xp0.shape[1] - self.Tp0))
        w1pad = ws.get("w1pad")
        if w1pad is not None:
            ops.copy(w1pad[:3 * Cn], self.W("encoder.conv1.kernel")[0])
            ops.gemm(xp0, w1pad, h1pad, self.T1, d, self.K1p, Cn, 1, d, 1, ldc=d, nbatch=B,
                     a_sb=xp0.stride(0), c_sb=h1pad.stride(0), c_off=self.pl2 * d,
                     bias=a.param("encoder.conv1.bias"), act=1, aux_out=u1pad)
        else:
            self._gemm_xw(xp0, "encoder.conv1.kernel", h1pad, self.T1, d, 3 * Cn, Cn, ldc=d, nbatch=B,
                          a_sb=xp0.stride(0), c_sb=h1pad.stride(0), c_off=self.pl2 * d,
                          bias=a.param("encoder.conv1.bias"), act=1, aux_out=u1pad)
        w2, ld2 = self.W("encoder.conv2.kernel")
        x = ws["enc0.x_in"] if cfg.encoder_layers else ws["enc_x"]
        self._gemm_xw(h1pad, "encoder.conv2.kernel", x, T, d, 3 * d, 2 * d, ldc=d, nbatch=B, a_sb=h1pad.stride(0),
                      c_sb=T * d, bias=a.param("encoder.conv2.bias"), act=1, aux_out=ws["u2"], resid=self.pe_enc_t,
                      r_ld=d, r_sb=0)

        if drop:
            self._dropout(x, x, SITE_ENC_STEM)  # W:342
        # ---- encoder layers (W:218-236)
        for i in range(cfg.encoder_layers):
            p, k = f"encoder.layers.{i}", f"enc{i}."
            x_in = ws[k + "x_in"]
            self._ln_fwd(x_in, p + ".self_attn_layer_norm", ws[k + "xn1"], k + "ln1")
            self._dense_fwd(ws[k + "xn1"], p + ".self_attn.qkv.kernel", ws[k + "qkv"], scale_cols=d, scale=scal_e)
            qkv = ws[k + "qkv"]
            self._attn_fwd(k + ("stats" if self.precision == "bf16" else "P"), (qkv, 0), (qkv, d), (qkv, 2 * d),
                           ws[k + "ctx"], B, He, T, T, 0, site=SITE_ENC_ATTN + i)
            self._dense_fwd(ws[k + "ctx"], p + ".self_attn.out_proj.kernel", ws[k + "x_mid"], resid=x_in, r_ld=d)
            self._ln_fwd(ws[k + "x_mid"], p + ".final_layer_norm", ws[k + "xn2"], k + "ln2")
            self._dense_fwd(ws[k + "xn2"], p + ".feed_forward.fc1.kernel", ws[k + "g"], act=1, aux_out=ws[k + "u"])
            nxt = ws[f"enc{i + 1}.x_in"] if i + 1 < cfg.encoder_layers else ws["enc_x"]
            # W:205: x_mid + Dropout(fc2(g)): the mask is a term of the GEMM epilogue (before the residual add)
            self._dense_fwd(ws[k + "g"], p + ".feed_forward.fc2.kernel", nxt, resid=ws[k + "x_mid"], r_ld=d,
                            **self._drop_epi(SITE_ENC_FFN + i))
        self._ln_fwd(ws["enc_x"], "encoder.layer_norm", ws["enc_out"], "enc_ln")
        enc_out = ws["enc_out"]

        # ---- decoder (W:394-466)
        self._wait_late()  # the previous step's late Adam slices, if they were left running (train.ADAM_LATE)
        if not early_dec:
            dec_embed()
        if Ld:
            if self._side is not None and Ld > 1:
                # W:122-123 for all layers: layer 0's k/v now (its cross-attention is next), the other layers' as ONE
                # GEMM on the second stream under decoder layer 0's chain of small kernels
                self._dense_fwd(enc_out, "decoder.cross_kv.kernel", kvc, n_off=0, n_cols=2 * d)
                self._run_on_side(lambda: self._dense_fwd(enc_out, "decoder.cross_kv.kernel", kvc[:, 2 * d:], n_off=2 * d,
                                                          n_cols=(Ld - 1) * 2 * d), enc_out)
                kv_rest = self._pop_side_reads(enc_out)
            else:
                self._dense_fwd(enc_out, "decoder.cross_kv.kernel", kvc)
                kv_rest = None
        for i in range(Ld):
            if not (early_dec and i == 0):
                dec_self_block(i)
            elif early_ev is not None:
                self._wait_events(early_ev)  # layer 0's self-attention block ran beside the encoder
            if i == 1 and kv_rest is not None:
                self._wait_events(kv_rest)
            dec_cross_ffn(i)
        self._ln_fwd(ws["dec_x"], "decoder.layer_norm", ws["dec_out"], "dec_ln")

        # ---- LM head + shifted cross-entropy (W:579-600); logits become dlogits in place
        V = cfg.vocab_size
        logits = ws["logits"]
        wl, ldw = self.W("lm_head.kernel")
        Vp = self.ldl  # pad columns of the stored kernel are zero: their logits are 0 and ignored by xent
        self._gemm_xw(ws["dec_out"], "lm_head.kernel", logits, B * S, Vp, d, d, ldc=Vp)
        gs = loss_scale / (B * (S - 1))
        # (tmi_linear_xent: the loss's target logit in fp32 from the LM head's own operands - bf16 logits have lost its low bits;
        # TMI_XENT_EXACT_TARGET=0: the plain cross-entropy of the stored logits)
        lm = (ws["dec_out"], d, wl, ldw, 1, d) if _XENT_EXACT else None
        ops.xent_fwd_bwd(logits, self.ldl, labels, ws["row_loss"], B, S, V, gs, lm=lm)
        ops.sum_scale(ws["row_loss"], ws["loss"], B * S, 1.0 / (B * (S - 1)))

        # ================= backward =================
        dres = ws["dres_dec"]
        dW = a.grad("lm_head.kernel")
        # the LM head's weight gradient (K = B*S rows against a [d, 51904] output: 120 us) feeds nothing on the chain: second
        # stream, under the decoder's backward (a chain of decoder-sized kernels on a mostly idle chip)
        self._run_on_side(lambda: ops.gemm(ws["dec_out"], logits, dW, d, Vp, B * S, 1, d, Vp, 1, Vp, splitk=0), logits)
        dtmp = ws["dtmp"][:B * S]
        # dgrad over the padded vocab (pad columns of dlogits are zero): a whole number of K tiles
        # (K = 51904 against only B*S x d outputs: in bf16 mode reduce it split-K into fp32 and round once)
        if self.precision == "bf16":
            acc = ws["lmh_dx32"]
            ops.fill_zero(acc)
            ops.gemm(logits, wl, acc, B * S, d, Vp, Vp, 1, 1, ldw, d, splitk=0)
            ops.cast_bf16(acc, d, dtmp, d, B * S, d)
        else:
            # (fp32 mode: the same split of the 51904-deep reduction, straight into the fp32 result)
            self._guard_write(dtmp)
            ops.fill_zero(dtmp)
            ops.gemm(logits, wl, dtmp, B * S, d, Vp, Vp, 1, 1, ldw, d, splitk=0)
        if early_update is not None:
            # lm_head: gradient final (weight gradient above, same stream), weights read for the last time by the dgrad just
            # enqueued on the main stream (the event _run_on_side records here orders the update behind it)
            lm_lo = a.offsets["lm_head.kernel"]
            # (keyed by a buffer nothing in backward rewrites: a _guard_write on the key would make the chain wait for Adam)
            self._run_on_side(lambda: early_update(lm_lo, a.numel), ws["row_loss"])
        # Column sums (bias gradients) and Dropout-masked copies of the residual-stream gradient come out of the LayerNorm
        # backward that produces it (tmi_layernorm_bwd_emit): ffn_emit(side, i) = what layer i's fc2 needs of the dres
        # handed down to it - its bias gradient and, with dropout, dres under the mask of W:205 in one of two
        # alternating buffers (a layer's weight gradient on the second stream may still be reading the other one)
        emit_on = os.environ.get("TMI_LN_EMIT", "1") != "0"

        Ld = cfg.decoder_layers
        # Decoder weight gradients deferred and batched over the layers (TMI_WGRAD_BATCH=0: one launch per layer, the round-2
        # form): 24 launches of 13-26 us on B*S = 800 rows become 6 with L times the tiles (KernelBlocks._wgrad_batched).
        batchd = os.environ.get("TMI_WGRAD_BATCH", "1") != "0" and Ld > 1

        def ffn_emit(side, i, rows):
            if not emit_on or i < 0:
                return None
            pre = f"{'encoder' if side == 'enc' else 'decoder'}.layers.{i}.feed_forward.fc2.bias"
            site = (SITE_ENC_FFN if side == "enc" else SITE_DEC_FFN) + i
            if side == "dec" and batchd:  # the per-layer buffer: masked copy (dropout) or snapshot (rate 0)
                return (a.grad(pre), ws[f"dec{i}.dyf"], site)
            return (a.grad(pre), ws[f"dyd{i & 1}"][:rows] if drop else None, site)

        def bias_emit(name, snapshot=None):
            """``snapshot``: per-layer buffer that receives a copy of the emitted dres (a deferred weight gradient's dy)."""
            return (a.grad(name), snapshot, None) if emit_on else None

        def snap(buf, have):
            """dres for a deferred reader: the LayerNorm backward wrote ``buf`` (``have``) or it is copied now."""
            if not have:
                self._guard_write(buf)
                ops.copy(buf, dres)
            return buf

        self._ln_bwd(dtmp, ws["dec_x"], "decoder.layer_norm", dres, "dec_ln", False, emit=ffn_emit("dec", Ld - 1, B * S))
        ready("decoder.layer_norm.gamma")

        d_enc, dkv = ws["d_enc_out"], ws["dkv_all"]
        # (opt-in: measured slower on MI355X, 9.41 -> 9.73 ms/step - four K = 12000 weight gradients with their own
        # split-K reductions cost more than the one fused GEMM saves by leaving the critical path)
        kv_per_layer = (self._side is not None and cfg.decoder_layers > 1 and self._main is not None and
                        os.environ.get("TMI_KV_PER_LAYER", "0") != "0")
        if cfg.decoder_layers:
            Lkv = cfg.decoder_layers * 2 * d
            gkv = a.grad("decoder.cross_kv.kernel").view(d, Lkv)
            gkvb = a.grad("decoder.cross_kv.bias")
            wkv, ldkv = self.W("decoder.cross_kv.kernel")
        for i in reversed(range(cfg.decoder_layers)):
            p, k = f"decoder.layers.{i}", f"dec{i}."
            Rd = B * S
            dt_, dctx = ws["dtmp"][:Rd], ws["dctx"][:Rd]
            dU = ws[k + "dU"] if batchd else ws["dU"][:Rd]
            dqkv = ws[k + "dqkv"] if batchd else ws["dqkv"][:Rd]
            wg = not batchd
            # FFN (with dropout the branch sees the masked gradient: the same mask, regenerated)
            dy = dres
            if batchd:
                dy = ws[k + "dyf"]
                if not emit_on:
                    if drop:
                        self._dropout(dres, dy, SITE_DEC_FFN + i)
                    else:
                        snap(dy, False)
            elif drop:
                dy = ws[f"dyd{i & 1}"][:Rd]
                if not emit_on:
                    self._dropout(dres, dy, SITE_DEC_FFN + i)
            self._dense_bwd(ws[k + "g"], dy, p + ".feed_forward.fc2.kernel", dU, aux_in=ws[k + "u"], bias_done=emit_on, wgrad=wg)
            self._dense_bwd(ws[k + "xn3"], dU, p + ".feed_forward.fc1.kernel", dt_, wgrad=wg)
            self._ln_bwd(dt_, ws[k + "x_mid2"], p + ".final_layer_norm", dres, k + "ln3", True,
                         emit=bias_emit(p + ".encoder_attn.out_proj.bias", ws[k + "dyoc"] if batchd else None))
            # cross attention: dK / dV feed only the shared k/v projections' backward after the loop, so their pass runs on the
            # second stream (its dO lives in a buffer of its own: the chain rewrites ws["dctx"] two kernels later)
            dctxc = ws[f"dctxc{i & 1}"][:Rd]
            dyoc = snap(ws[k + "dyoc"], emit_on) if batchd else dres
            self._dense_bwd(ws[k + "ctxc"], dyoc, p + ".encoder_attn.out_proj.kernel", dctxc, bias_done=emit_on, wgrad=wg)
            dqc = ws[k + "dqc"] if batchd else ws["dtmp"][:Rd]
            self._attn_bwd(k + ("statsc" if self.precision == "bf16" else "Pc"), (ws[k + "qc"], 0),
                           (kvc, 2 * i * d), (kvc, (2 * i + 1) * d), ws[k + "ctxc"], dctxc, (dqc, 0),
                           (dkv, 2 * i * d), (dkv, (2 * i + 1) * d), B, Hd, S, T, 0, site=SITE_DEC_CROSS + i,
                           dkv_on_side=not kv_per_layer)
            if kv_per_layer:
                # this layer's dk / dv are final: its share of the cross-attention k/v projections' backward (weight
                # and bias gradient, and d enc_out += dkv_i . Wkv_i^T) goes to the second stream now, under the chain of
                # decoder-sized kernels that follows, instead of one K = L*2d GEMM alone on the chip after the loop
                def kv_backward(i=i):
                    lo = 2 * i * d
                    ops.gemm(enc_out, dkv, gkv, d, 2 * d, B * T, 1, enc_out.stride(0), dkv.stride(0), 1, Lkv, splitk=0,
                             b_off=lo, c_off=lo)
                    ops.bias_grad(dkv[:, lo:lo + 2 * d], gkvb[lo:lo + 2 * d])
                    ops.gemm(dkv, wkv, d_enc, B * T, d, 2 * d, dkv.stride(0), 1, 1, ldkv, d_enc.stride(0),
                             accumulate=(i != cfg.decoder_layers - 1), a_off=lo, b_off=lo)
                self._run_on_side(kv_backward, dkv[:, 2 * i * d:])
            dxn2 = ws["dctx"][:Rd]
            self._dense_bwd(ws[k + "xn2"], dqc, p + ".encoder_attn.q_proj.kernel", dxn2, wgrad=wg)
            self._ln_bwd(dxn2, ws[k + "x_mid"], p + ".encoder_attn_layer_norm", dres, k + "ln2", True,
                         emit=bias_emit(p + ".self_attn.out_proj.bias", ws[k + "dyos"] if batchd else None))
            # self attention
            dyos = snap(ws[k + "dyos"], emit_on) if batchd else dres
            self._dense_bwd(ws[k + "ctx"], dyos, p + ".self_attn.out_proj.kernel", dctx, bias_done=emit_on, wgrad=wg)
            qkv = ws[k + "qkv"]
            self._attn_bwd(k + ("stats" if self.precision == "bf16" else "P"), (qkv, 0), (qkv, d), (qkv, 2 * d),
                           ws[k + "ctx"], dctx, (dqkv, 0), (dqkv, d), (dqkv, 2 * d), B, Hd, S, S, 1, site=SITE_DEC_SELF + i)
            self._dense_bwd(ws[k + "xn1"], dqkv, p + ".self_attn.qkv.kernel", dt_, wgrad=wg)
            self._ln_bwd(dt_, ws[k + "x_in"], p + ".self_attn_layer_norm", dres, k + "ln1", True, emit=ffn_emit("dec", i - 1, Rd))
            if not batchd:
                ready(p + ".self_attn_layer_norm.gamma")
        if cfg.decoder_layers:
            if kv_per_layer:
                self._wait_events(self._pop_side_reads(dkv))  # d_enc is complete after layer 0's share
            else:
                # every layer's dk / dv is in place (their passes ran on the second stream: join it): one weight gradient, one
                # bias gradient and one dgrad (K = L*2d) for the cross-attention k/v projections of all layers
                self._join_side()
                self._dense_bwd(enc_out, dkv, "decoder.cross_kv.kernel", d_enc)
            if batchd:
                st = {n: ws[f"dec*.{n}"] for n in ("xn1", "ctx", "xn2", "ctxc", "xn3", "g", "dqkv", "dyos", "dqc", "dyoc", "dU", "dyf")}
                lay = "decoder.layers.{}"

                def decoder_weight_grads():
                    nb = not emit_on  # the biases of the residual-stream layers come out of the LayerNorm backward otherwise
                    self._wgrad_batched(st["g"], st["dyf"], lay + ".feed_forward.fc2.kernel", Ld, bias=nb)
                    self._wgrad_batched(st["xn3"], st["dU"], lay + ".feed_forward.fc1.kernel", Ld)
                    self._wgrad_batched(st["ctxc"], st["dyoc"], lay + ".encoder_attn.out_proj.kernel", Ld, bias=nb)
                    self._wgrad_batched(st["xn2"], st["dqc"], lay + ".encoder_attn.q_proj.kernel", Ld)
                    self._wgrad_batched(st["ctx"], st["dyos"], lay + ".self_attn.out_proj.kernel", Ld, bias=nb)
                    self._wgrad_batched(st["xn1"], st["dqkv"], lay + ".self_attn.qkv.kernel", Ld)
                # on the second stream, under the start of the encoder's backward
                self._run_on_side(decoder_weight_grads, st["dqkv"])
                for i in reversed(range(Ld)):
                    ready(f"decoder.layers.{i}.self_attn_layer_norm.gamma")
            ready("decoder.cross_kv.kernel")
        # the embedding's backward (mask of W:411, scatter of the rows) feeds nothing on the chain: second stream
        gemb, dres_dec = a.grad("decoder.embed_tokens.embeddings"), dres

        def embed_backward():
            if drop:
                ops.dropout(dres_dec, dres_dec, dres_dec.shape[0], dres_dec.shape[1], self._drop_p, self._site_seed(SITE_DEC_EMBED))
            ops.embed_bwd(labels, dres_dec, gemb, B, S, d, cfg.decoder_start_token_id)
            if early_update is not None:  # the table's gradient is final; its forward read happened long ago
                e_lo = a.offsets["decoder.embed_tokens.embeddings"]
                early_update(e_lo, e_lo + gemb.numel())
        self._run_on_side(embed_backward, dres_dec)
        ready("decoder.embed_tokens.embeddings")

        # ---- encoder backward
        dres = ws["dres_enc"]
        if cfg.decoder_layers == 0:
            ops.fill_zero(d_enc)
        R = B * T
        self._ln_bwd(d_enc, ws["enc_x"], "encoder.layer_norm", dres, "enc_ln", False,
                     emit=ffn_emit("enc", cfg.encoder_layers - 1, R))
        ready("encoder.layer_norm.gamma")
        for i in reversed(range(cfg.encoder_layers)):
            p, k = f"encoder.layers.{i}", f"enc{i}."
            dU, dt_, dctx, dqkv = ws["dU"][:R], ws["dtmp"][:R], ws["dctx"][:R], ws["dqkv"][:R]
            dy = dres
            if drop:
                dy = ws[f"dyd{i & 1}"][:R]
                if not emit_on:
                    self._dropout(dres, dy, SITE_ENC_FFN + i)
            # (TMI_DEFER_WGRAD=1: the two FFN weight gradients are enqueued when the attention backward starts - MFMA-bound work
            # beside the VALU-bound attention kernels instead of beside the FFN dgrads)
            # (the four weight gradients of a layer alternate between the two weight-gradient streams: KernelBlocks.N_LANES)
            self._dense_bwd(ws[k + "g"], dy, p + ".feed_forward.fc2.kernel", dU, aux_in=ws[k + "u"], bias_done=emit_on,
                            defer=dy is not dres, lane=0)
            self._dense_bwd(ws[k + "xn2"], dU, p + ".feed_forward.fc1.kernel", dt_, defer=True, lane=1)
            self._ln_bwd(dt_, ws[k + "x_mid"], p + ".final_layer_norm", dres, k + "ln2", True,
                         emit=bias_emit(p + ".self_attn.out_proj.bias"))
            self._dense_bwd(ws[k + "ctx"], dres, p + ".self_attn.out_proj.kernel", dctx, bias_done=emit_on, lane=0)
            qkv = ws[k + "qkv"]
            self._flush_deferred()
            self._attn_bwd(k + ("stats" if self.precision == "bf16" else "P"), (qkv, 0), (qkv, d), (qkv, 2 * d),
                           ws[k + "ctx"], dctx, (dqkv, 0), (dqkv, d), (dqkv, 2 * d), B, He, T, T, 0, site=SITE_ENC_ATTN + i)
            self._dense_bwd(ws[k + "xn1"], dqkv, p + ".self_attn.qkv.kernel", dt_, lane=1)
            self._ln_bwd(dt_, ws[k + "x_in"], p + ".self_attn_layer_norm", dres, k + "ln1", True, emit=ffn_emit("enc", i - 1, R))
            ready(p + ".self_attn_layer_norm.gamma")

        # ---- stem backward: x0 = gelu(u2) + PE ; u2 = conv2(h1) ; h1 = gelu(u1) ; u1 = conv1(x)
        if drop:
            self._dropout(dres, dres, SITE_ENC_STEM)
        du2pad, dh1pad = ws["du2pad"], ws["dh1pad"]
        du2 = du2pad[:, 1:]  # row 0 of every batch stays zero (the "t-1" term of the first output)
        # one launch over the B per-sample spans (du2 skips the zero row 0 of every sample)
        ops.gelu_bwd_batched(dres, ws["u2"], du2, T * d, B, T * d, T * d, du2pad.stride(0))
        # the pad rows of du2pad are zero, so the bias gradient is one column sum over the whole buffer
        gw2 = a.grad("encoder.conv2.kernel").view(3 * d, d)

        def conv2_weight_grads():
            ops.bias_grad(du2pad.view(-1, d), a.grad("encoder.conv2.bias"))
            ops.gemm(h1pad, du2pad, gw2, 3 * d, d, T, 1, 2 * d, d, 1, d, kbatch=B, a_skb=h1pad.stride(0),
                     b_skb=du2pad.stride(0), b_off=d, splitk=0)
        # conv2's weight gradient (88 us + its split-K reduce) feeds nothing on the chain: beside the two dgrad launches
        # below (du2pad and h1pad are not rewritten before the join)
        if os.environ.get("TMI_CONV_WGRAD_SIDE", "1") != "0":
            self._run_on_side(conv2_weight_grads, du2pad)
        else:
            conv2_weight_grads()
        sd = du2pad.stride(0)
        # even padded rows u = 2j: dY[j]·W0ᵀ + dY[j-1]·W2ᵀ  (kbatch walks the two kernel taps)
        ops.gemm(du2pad, w2, dh1pad, T, d, d, d, 1, 1, ld2, 2 * d, nbatch=B, a_sb=sd, c_sb=dh1pad.stride(0),
                 kbatch=2, a_skb=-d, b_skb=2 * d * ld2, a_off=d, aux_in=u1pad)
        # odd padded rows u = 2j + 1: dY[j]·W1ᵀ
        ops.gemm(du2pad, w2, dh1pad, T, d, d, d, 1, 1, ld2, 2 * d, nbatch=B, a_sb=sd, c_sb=dh1pad.stride(0),
                 a_off=d, b_off=d * ld2, c_off=d, aux_in=u1pad)
        if self.pl2:
            ops.fill_zero(dh1pad[:, :self.pl2])
        ops.fill_zero(dh1pad[:, self.pl2 + self.T1:])
        ops.bias_grad(dh1pad.view(-1, d), a.grad("encoder.conv1.bias"))  # (pad rows were just zeroed)
        gw1 = a.grad("encoder.conv1.kernel").view(3 * Cn, d)
        ops.gemm(xp0, dh1pad, gw1, 3 * Cn, d, self.T1, 1, Cn, d, 1, d, kbatch=B, a_skb=xp0.stride(0),
                 b_skb=dh1pad.stride(0), b_off=self.pl2 * d, splitk=0)
        ready("encoder.conv1.kernel")
        self._join_side()
        return ws["loss"]

    def __call__(self, features, labels=None, training=True):
        """Reference call surface (W:829): returns {"loss": ...}.  Gradients are a side effect."""
        if not training or labels is None:
            raise NotImplementedError("only the training path (labels given, training=True) is on the hot path")
        return {"loss": self.forward_backward(features, labels)}


def create_whisper_model(model_type: str = "small", device="cuda:0", precision: str = "bf16", seed: int = 1234,
                         **overrides) -> WhisperForConditionalGeneration:
    """W:852-890."""
    return WhisperForConditionalGeneration(make_config(model_type, **overrides), device=device,
                                           precision=precision, seed=seed)
