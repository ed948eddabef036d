"""TEST INFRASTRUCTURE (checker only): numpy restatement of the counter-based dropout generator of
tethys-speech_amd/csrc/tmi_common.h (tmi_mix32 / tmi_row_key / tmi_pair_hash / tmi_stream_key / tmi_keep, and the attention
kernels' tmi_quad_hash / tmi_keep_attn).

The reference applies tf.keras.layers.Dropout in training (speech_jobs/whisper_dist.py:160, 205, 342, 411);
TensorFlow's RNG stream cannot be reproduced, so the masks below are this build's own and parity with
dropout enabled is defined against THIS generator: the same integer arithmetic on the host."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & M32


def mix32(x):
    x = _u32(x)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


M24 = np.uint64(0xFFFFFF)


def _mul24(a, b):
    return ((_u32(a) & M24) * (_u32(b) & M24)) & M32


def row_key(key, row):
    """tmi_row_key: two full avalanches per mask row (seed, stream and row reach every bit of both words)."""
    key, row = _u32(key), _u32(row)
    ra = mix32(key ^ ((row * np.uint64(0x9E3779B1)) & M32))
    rb = mix32(((key + np.uint64(0x632BE5AB)) & M32) ^ ((row * np.uint64(0x85EBCA6B)) & M32))
    return ra, rb


def pair_hash(rk, cp):
    """tmi_pair_hash: 32 bits for column pair ``cp`` (= column >> 1) of the row with key ``rk`` = (ra, rb)."""
    ra, rb = rk
    h = _mul24(_u32(ra) ^ _u32(cp), 0x9E3779)
    h = h ^ (h >> np.uint64(15)) ^ _u32(rb)
    return _mul24(h, 0x85EBCB)


def stream_key(seed, stream_id):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    return mix32(np.uint64(lo) ^ mix32((np.uint64(hi) + _u32(stream_id)) & M32))


def drop_thr(p):
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def keep_scale(p):
    thr = drop_thr(p)
    return float(np.float32(65536.0) / np.float32(65536 - thr))


def keep_rows(key, rows, cols, thr):
    """tmi_keep over a [rows, cols] grid of one stream: bool array."""
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    h = pair_hash(row_key(key, r), c >> np.uint64(1))
    d = np.where(c & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF))
    return d >= np.uint64(thr)


def keep_flat(seed, rows, cols, p):
    """[rows, cols] bool mask of tmi_dropout / the GEMM-epilogue and LayerNorm-backward dropout terms: stream 0."""
    return keep_rows(stream_key(seed, 0), rows, cols, drop_thr(p))


def quad_hash(rk, cq):
    """tmi_quad_hash: two 32-bit words (ha, hb) for column quad ``cq`` (= column >> 2) of the row with key ``rk``: the
    attention generator - one evaluation serves the four consecutive keys a forward lane holds in one accumulator quad."""
    ra, rb = rk
    h = _mul24(_u32(ra) ^ _u32(cq), 0x9E3779)
    h = h ^ (h >> np.uint64(15)) ^ _u32(rb)
    return _mul24(h, 0x85EBCB), _mul24(h, 0xC2B2AE)


def keep_rows_attention(key, rows, cols, thr):
    """tmi_keep_attn over a [rows, cols] grid of one stream: column 4cq + {0, 1, 2, 3} draws the {low, high} half of
    {ha, hb} as a SIGNED 16-bit number and is kept when draw >= thr - 32768."""
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    ha, hb = quad_hash(row_key(key, r), c >> np.uint64(2))
    w = np.where(c & np.uint64(2), hb, ha)
    d = np.where(c & np.uint64(1), w >> np.uint64(16), w & np.uint64(0xFFFF)).astype(np.uint16).view(np.int16)
    return d.astype(np.int32) >= (int(thr) - 32768)


def keep_attention(seed, B, H, Tq, Tk, p):
    """[B, H, Tq, Tk] bool mask of the attention kernels (tmi_attn_fwd draws it and stores it, tmi_attn_bwd reads the
    stored bits): stream b*H + head, row = query, column = key."""
    out = np.empty((B, H, Tq, Tk), dtype=bool)
    thr = drop_thr(p)
    for b in range(B):
        for h in range(H):
            out[b, h] = keep_rows_attention(stream_key(seed, b * H + h), Tq, Tk, thr)
    return out


# ---- site seeds of the Whisper step (tethys-speech_amd/whisper.py SITE_*, KernelBlocks._site_seed restated)
SITE_BASE = {"encoder.stem": 1, "decoder.embed": 2}
_LAYER_SITES = (("encoder.layers.", ".self_attn", 100), ("encoder.layers.", ".feed_forward", 200),
                ("decoder.layers.", ".self_attn", 300), ("decoder.layers.", ".encoder_attn", 400),
                ("decoder.layers.", ".feed_forward", 500))


def site_id(name: str) -> int:
    if name in SITE_BASE:
        return SITE_BASE[name]
    for head, tail, base in _LAYER_SITES:
        if name.startswith(head) and name.endswith(tail):
            return base + int(name[len(head):-len(tail)])
    raise KeyError(name)


def site_seed(base_seed: int, step: int, site: int) -> int:
    return (base_seed + step * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


# ---- Wav2Vec2 sites (tethys-speech_amd/wav2vec2.py SITE_*)
W2V_SITE_BASE = {"feature_extractor": 1, "feature_projection": 2, "project_hid": 3, "project_q": 4}
_W2V_LAYER_SITES = ((".attention", 100), (".attention_output", 200), (".intermediate", 300), (".output", 400))


def w2v_site_id(name: str) -> int:
    if name in W2V_SITE_BASE:
        return W2V_SITE_BASE[name]
    head = "encoder.layers."
    for tail, base in _W2V_LAYER_SITES:
        if name.startswith(head) and name.endswith(tail) and name[len(head):-len(tail)].isdigit():
            return base + int(name[len(head):-len(tail)])
    raise KeyError(name)


class HostDropout:
    """Mask provider for the oracles' DROPOUT_PROVIDER hooks: the masks the HIP step draws in step ``step``."""

    def __init__(self, base_seed: int, step: int = 0, site_id=site_id):
        self.base_seed, self.step, self.site_id = base_seed, step, site_id

    def mask(self, site: str, shape, rate):
        import torch
        seed = site_seed(self.base_seed, self.step, self.site_id(site))
        if len(shape) == 4:  # attention probabilities [B, H, Tq, Tk]
            keep = keep_attention(seed, *shape, rate)
        else:                # hidden states [B, T, d] as rows x d
            rows = int(np.prod(shape[:-1]))
            keep = keep_flat(seed, rows, shape[-1], rate).reshape(shape)
        return torch.from_numpy(keep), keep_scale(rate)
