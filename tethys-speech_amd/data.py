"""Synthetic dataset of the reference (W:784-815), resident in HBM.

50 samples: features randn(50, n_mels, seq_len) fp32, labels [50, max_target_length] int32
with [0]=BOS(1), [1:len-1]=randint(3,100), [len-1]=EOS(2), rest 0, len=randint(50,90);
``dataset.batch(global_batch).repeat()`` without drop_remainder.  The reference draws from
the unseeded legacy NumPy RNG; here the pool comes from ``numpy.random.default_rng(seed)``
(seed 1234) and is uploaded once, so steps read their batch from device
memory with no host copy.
"""
from __future__ import annotations

import numpy as np
import torch


def make_pool(seed=1234, n_mels=80, seq_len=3000, max_target_length=100, num_samples=50):
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


class DummyDataset:
    """Iterator over per-replica batches.  The global batch of step i is samples
    [i*GB, (i+1)*GB) of the repeating pool pass (short last batch of each pass kept, as the
    reference); replica r takes rows [r*B, (r+1)*B) of it (possibly fewer, possibly none)."""

    def __init__(self, batch_size, n_mels=80, seq_len=3000, max_target_length=100, device="cuda:0",
                 rank=0, world=1, seed=1234, drop_remainder=False, num_samples=50):
        f, l = make_pool(seed, n_mels, seq_len, max_target_length, num_samples)
        self.features = torch.from_numpy(f).to(device)
        self.labels = torch.from_numpy(l).to(device)
        self.batch_size, self.rank, self.world = batch_size, rank, world
        self.global_batch = batch_size * world
        self.n = num_samples
        self.drop_remainder = drop_remainder
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._pos >= self.n or (self.drop_remainder and self._pos + self.global_batch > self.n):
            self._pos = 0
        s = self._pos
        e = min(self.n, s + self.global_batch)
        self._pos = e
        lo = min(e, s + self.rank * self.batch_size)
        hi = min(e, lo + self.batch_size)
        return self.features[lo:hi], self.labels[lo:hi]


def create_dummy_dataset(batch_size, n_mels=80, seq_len=3000, max_target_length=100, **kw):
    """W:784 signature; ``batch_size`` is the PER-REPLICA batch (the reference passes the
    global batch and lets tf.distribute split it; rank/world are keyword arguments here)."""
    return DummyDataset(batch_size, n_mels, seq_len, max_target_length, **kw)


class W2VDummyDataset:
    """V:1123-1153: 50 x normal([32000]) fp32 clips (2 s at 16 kHz), labels unused,
    ``batch(global_batch, drop_remainder=True).repeat()``; replica r takes rows
    [r*B, (r+1)*B) of every global batch.  The pool is uploaded once."""

    def __init__(self, batch_size, length=32000, device="cuda:0", rank=0, world=1, seed=1234, num_samples=50,
                 drop_remainder=True):
        """``drop_remainder=False`` is speech_jobs/whisper_single.py:1111's / stable_jobs/wav2vec2_dist.py:1111's
        ``dataset.batch(global_batch).repeat()``: the last global batch of a pass is short, and a replica's slice
        [r*B, (r+1)*B) of it may be short or empty."""
        pool = np.random.default_rng(seed).standard_normal((num_samples, length)).astype(np.float32)
        self.audio = torch.from_numpy(pool).to(device)
        self.batch_size, self.rank, self.world = batch_size, rank, world
        self.global_batch = batch_size * world
        self.n = num_samples // self.global_batch * self.global_batch if drop_remainder else num_samples
        if self.n == 0:
            raise ValueError("global batch larger than the 50-clip pool")
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._pos >= self.n:
            self._pos = 0
        s = min(self._pos + self.rank * self.batch_size, self.n)
        self._pos += self.global_batch
        return self.audio[s:min(s + self.batch_size, self.n)]
