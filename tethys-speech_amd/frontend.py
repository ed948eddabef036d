"""Log-mel front end on the GPU (speech_jobs/whisper_dist.py:739-766, ``extract_fbank_features``):
16 kHz waveforms -> log-mel features the Whisper encoder consumes.

Dead code in the reference's training path (its batches are pre-made random "mel" tensors, W:792;
SURVEY 8f row 4); built so that real audio can feed the step.  Two launches per batch, both through
the C ABI:

  1. windowed DFT = one exact-fp32 ``tmi_gemm``: the frames are OVERLAPPING ROWS of the waveform
     (row stride = hop 160, K = n_fft 400), B = [400, 402] = Hann[n] * (cos | -sin)(2 pi k n / 400);
  2. ``tmi_logmel_from_spectrum``: power, 80-bin HTK mel matrix (0-8 kHz), log(x + 1e-6).

Output is channels-first ``[B, n_mels, frames]`` (what the encoder reads) by default; the reference
returns ``[frames, n_mels]`` and feeds it un-transposed (W:972-977) — ``reference_layout=True``
reproduces that layout.  The mel / DFT matrices are built once on the host in fp64 (constants)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def hann_periodic(n: int) -> np.ndarray:
    """tf.signal.hann_window(periodic=True), the default window of tf.signal.stft (W:744)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def mel_weight_matrix(n_mels=80, n_bins=201, sample_rate=16000, lower=0.0, upper=8000.0) -> np.ndarray:
    """tf.signal.linear_to_mel_weight_matrix (W:755-758): HTK mel scale, DC bin zeroed, triangles
    evaluated in the mel domain."""
    mel = lambda f: 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    lin = np.linspace(0.0, sample_rate / 2.0, n_bins)[1:]
    bins_mel = mel(lin)[:, None]
    edges = np.linspace(mel(lower), mel(upper), n_mels + 2)
    lo, ce, up = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    w = np.maximum(0.0, np.minimum((bins_mel - lo) / (ce - lo), (up - bins_mel) / (up - ce)))
    return np.concatenate([np.zeros((1, n_mels)), w], axis=0)


class LogMelFrontend:
    def __init__(self, device="cuda:0", sample_rate=16000, n_mels=80, n_fft=400, hop_length=160, eps=1e-6):
        lib()  # fail loudly if the HIP library is missing
        self.device = torch.device(device)
        self.n_fft, self.hop, self.n_mels, self.eps = int(n_fft), int(hop_length), int(n_mels), float(eps)
        self.n_bins = self.n_fft // 2 + 1
        n = np.arange(self.n_fft)[:, None]
        k = np.arange(self.n_bins)[None, :]
        ang = 2.0 * np.pi * n * k / self.n_fft
        win = hann_periodic(self.n_fft)[:, None]
        dft = np.concatenate([win * np.cos(ang), -win * np.sin(ang)], axis=1)  # [n_fft, 2 * n_bins]
        self.dft = torch.from_numpy(dft.astype(np.float32)).to(self.device).contiguous()
        self.mel = torch.from_numpy(mel_weight_matrix(n_mels, self.n_bins, sample_rate, 0.0, sample_rate // 2)
                                    .astype(np.float32)).to(self.device).contiguous()
        self._spec = None

    def num_frames(self, n_samples: int) -> int:
        return 1 + (n_samples - self.n_fft) // self.hop if n_samples >= self.n_fft else 0

    def __call__(self, waveform: torch.Tensor, reference_layout: bool = False) -> torch.Tensor:
        """waveform [B, N] float32 on the device -> [B, n_mels, frames] (or [B, frames, n_mels])."""
        if waveform.dim() == 1:
            waveform = waveform[None]
        if waveform.dtype != torch.float32 or waveform.device != self.device or not waveform.is_contiguous():
            raise TypeError("waveform must be a contiguous float32 tensor on the front end's device")
        B, N = waveform.shape
        F = self.num_frames(N)
        if F <= 0:
            raise ValueError("waveform shorter than one frame")
        nb2 = 2 * self.n_bins
        if self._spec is None or self._spec.shape != (B, F, nb2):
            self._spec = torch.empty(B, F, nb2, dtype=torch.float32, device=self.device)
        spec = self._spec
        # frames are overlapping rows: A[b, f, n] = waveform[b, f * hop + n]
        ops.gemm(waveform, self.dft, spec, F, nb2, self.n_fft, self.hop, 1, nb2, 1, nb2, nbatch=B, a_sb=N, b_sb=0,
                 c_sb=F * nb2)
        if reference_layout:
            out = torch.empty(B, F, self.n_mels, dtype=torch.float32, device=self.device)
            ld = self.n_mels
        else:
            out = torch.empty(B, self.n_mels, F, dtype=torch.float32, device=self.device)
            ld = F
        for b in range(B):
            check(lib().tmi_logmel_from_spectrum(spec[b].data_ptr(), nb2, self.mel.data_ptr(), out[b].data_ptr(), F,
                                                 self.n_bins, self.n_mels, self.eps, 0 if reference_layout else 1, ld,
                                                 ops.stream()), "tmi_logmel_from_spectrum")
        return out
