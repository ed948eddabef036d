"""CPU restatement of the reference's log-mel front end (speech_jobs/whisper_dist.py:739-766,
``extract_fbank_features``).  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing on the
product path).  PARITY UNPINNED: the arithmetic lives in ``tf.signal`` (TensorFlow 2.10, NGC 22.12 image,
Dockerfile:1), which is absent here and not installable; the reference has no test or fixture for it.
What is restated is tf.signal's published behaviour at the reference's call site:

  tf.signal.stft(x, frame_length=400, frame_step=160, fft_length=400)          (W:744-749)
      frames = 1 + (N - 400) // 160 (pad_end=False), periodic Hann window, rfft -> 201 bins
  power = |stft|^2                                                             (W:752)
  tf.signal.linear_to_mel_weight_matrix(80, 201, 16000, 0, 8000)               (W:755-758)
      HTK mel scale mel(f) = 1127 ln(1 + f / 700); the DC bin is excluded (zero row); band edges
      are 82 points equally spaced in mel between mel(0) and mel(8000); triangle weights
      max(0, min(lower slope, upper slope)) evaluated in the mel domain
  log(power . mel + 1e-6)                                                      (W:761-764)

The result is [frames, n_mels]; the reference passes it to the encoder un-transposed (W:972-977),
a latent bug noted in SURVEY.md 8(f): the model wants [n_mels, frames].
"""
import numpy as np


def hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def linear_to_mel_weight_matrix(n_mels=80, n_bins=201, sample_rate=16000, lower=0.0, upper=8000.0) -> np.ndarray:
    def mel(f):
        return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, n_bins)[1:]  # DC bin excluded
    bins_mel = mel(lin)[:, None]
    edges = np.linspace(mel(lower), mel(upper), n_mels + 2)
    lo, ce, up = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (bins_mel - lo) / (ce - lo)
    upper_slopes = (up - bins_mel) / (up - ce)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    return np.concatenate([np.zeros((1, n_mels)), w], axis=0)  # [n_bins, n_mels]


def extract_fbank_features(waveform, sample_rate=16000, n_mels=80, n_fft=400, hop_length=160, dtype=np.float64):
    x = np.asarray(waveform, dtype=dtype)
    n = x.shape[-1]
    frames = 1 + (n - n_fft) // hop_length if n >= n_fft else 0
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(frames)[:, None]
    fr = x[..., idx] * hann_periodic(n_fft).astype(dtype)
    spec = np.fft.rfft(fr, n=n_fft, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ linear_to_mel_weight_matrix(n_mels, n_fft // 2 + 1, sample_rate, 0.0, sample_rate // 2).astype(dtype)
    return np.log(mel + 1e-6)
