"""TEST INFRASTRUCTURE (checker only): numpy restatement of the counter-based dropout generator of
tethys-speech_amd/csrc/tmi_common.h (tmi_mix32 / tmi_pair_hash / tmi_stream_key / tmi_keep).

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


def pair_hash(pid, key):
    a = (_u32(pid) ^ _u32(key)) & M32
    a ^= a >> np.uint64(17)  # fold bits 17..31 into the 24 the multiplier sees (counters >= 2^24, full-width keys)
    h = _mul24(a, 0x9E3779)
    h ^= h >> np.uint64(15)
    return (_mul24(h, 0x85EBCB) + (a >> np.uint64(8))) & M32


def stream_key(seed, stream_id):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    return mix32(np.uint64(lo) ^ mix32((np.uint64(hi) + _u32(stream_id)) & M32))


def drop_thr(p):
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def keep_scale(p):
    thr = drop_thr(p)
    return float(np.float32(65536.0) / np.float32(65536 - thr))


def keep_counter(key, idx, thr):
    """keep decision of flat counters ``idx`` (uint64 array) in the stream with key ``key``."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = pair_hash((idx >> np.uint64(1)) & M32, key)
    r = np.where(idx & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF))
    return r >= np.uint64(thr)


def keep_flat(seed, rows, cols, p):
    """[rows, cols] bool mask of tmi_dropout (cols even): counter = r * cols + c, stream 0."""
    idx = np.arange(rows * cols, dtype=np.uint64).reshape(rows, cols)
    return keep_counter(stream_key(seed, 0), idx, drop_thr(p))


def keep_attention(seed, B, H, Tq, Tk, p):
    """[B, H, Tq, Tk] bool mask of the attention kernels: stream b*H + head, counter q * 2*ceil(Tk/2) + k."""
    kp = (Tk + 1) // 2
    q = np.arange(Tq, dtype=np.uint64)[:, None]
    k = np.arange(Tk, dtype=np.uint64)[None, :]
    pid = (q * np.uint64(kp) + (k >> np.uint64(1))) & M32
    out = np.empty((B, H, Tq, Tk), dtype=bool)
    thr = np.uint64(drop_thr(p))
    for b in range(B):
        for h in range(H):
            hh = pair_hash(pid, stream_key(seed, b * H + h))
            r = np.where(k & np.uint64(1), hh >> np.uint64(16), hh & np.uint64(0xFFFF))
            out[b, h] = r >= thr
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
