"""Float64 oracle of the mel front end (TEST INFRASTRUCTURE ONLY).

Reference path: ``meldataset.py:34-40`` (DEFAULT_MEL_PARAMS), ``:58-77``
(``torchaudio.transforms.MelSpectrogram(**mel_params)``), ``:644`` (the call)
and ``:650`` (``(log(1e-5 + mel) - mean) / std`` with mean, std = -4, 4).

The transform itself lives in torchaudio, which is neither vendored in the
reference nor installed in this image, so this is a restatement of
torchaudio's *published* MelSpectrogram semantics with its defaults:
``f_min=0, f_max=sr//2, power=2, normalized=False, center=True,
pad_mode="reflect", window=hann(periodic), onesided=True, norm=None,
mel_scale="htk"``.  PARITY UNPINNED by the reference (it holds no vectors for
this stage); pinned by agreement of two independent restatements:

* :func:`mel_spectrogram` -- float64, direct O(N^2) DFT or ``numpy.fft.rfft``;
* :func:`mel_spectrogram_torch_stft` -- ``torch.stft`` float32 on CPU.
"""
from __future__ import annotations

import numpy as np

DEFAULT_MEL_PARAMS = dict(sample_rate=24000, n_mels=80, n_fft=1024,
                          win_length=1024, hop_length=300)  # meldataset.py:34-40
LOG_EPS = 1e-5      # meldataset.py:650
MEL_MEAN = -4.0     # meldataset.py:111
MEL_STD = 4.0


def hann_window(n: int) -> np.ndarray:
    """Periodic Hann window (torch.hann_window default), float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_freqs: int = 513, f_min: float = 0.0, f_max: float = 12000.0,
                   n_mels: int = 80, sample_rate: int = 24000) -> np.ndarray:
    """(n_freqs, n_mels) triangular HTK filterbank, norm=None (float64)."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]          # (n_freqs, n_mels+2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def reflect_index(j: np.ndarray, n: int) -> np.ndarray:
    """Index into the un-padded signal for reflect padding (no edge repeat)."""
    j = np.where(j < 0, -j, j)
    return np.where(j >= n, 2 * (n - 1) - j, j)


def num_frames(n_samples: int, hop: int = 300) -> int:
    """center=True STFT frame count: 1 + N // hop."""
    return 1 + n_samples // hop


def frame_signal(x: np.ndarray, n_fft: int = 1024, hop: int = 300) -> np.ndarray:
    """(L, n_fft) centre-padded (reflect, n_fft//2 each side) frames."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    L = num_frames(n, hop)
    idx = (np.arange(L)[:, None] * hop + np.arange(n_fft)[None, :]) - n_fft // 2
    return x[reflect_index(idx, n)]


def stft_power(x: np.ndarray, n_fft: int = 1024, hop: int = 300,
               direct_dft: bool = False) -> np.ndarray:
    """(n_fft//2+1, L) power spectrogram |STFT|^2, float64."""
    frames = frame_signal(x, n_fft, hop) * hann_window(n_fft)[None, :]
    if direct_dft:
        k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
        n = np.arange(n_fft, dtype=np.float64)[None, :]
        ang = -2.0 * np.pi * ((k * n) % n_fft) / n_fft
        spec = frames @ (np.cos(ang) + 1j * np.sin(ang)).T
    else:
        spec = np.fft.rfft(frames, axis=1)
    return (spec.real ** 2 + spec.imag ** 2).T


def mel_spectrogram(x: np.ndarray, sample_rate: int = 24000, n_fft: int = 1024,
                    hop_length: int = 300, n_mels: int = 80,
                    direct_dft: bool = False, **_unused) -> np.ndarray:
    """(n_mels, 1 + N//hop) mel power spectrogram, float64."""
    fb = mel_filterbank(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    return fb.T @ stft_power(x, n_fft, hop_length, direct_dft)


def log_normalise(mel: np.ndarray) -> np.ndarray:
    """meldataset.py:650 -- (log(1e-5 + mel) - (-4)) / 4."""
    return (np.log(LOG_EPS + mel) - MEL_MEAN) / MEL_STD


def log_mel(x: np.ndarray, **kw) -> np.ndarray:
    return log_normalise(mel_spectrogram(x, **kw))


def mel_spectrogram_torch_stft(x, sample_rate: int = 24000, n_fft: int = 1024,
                               hop_length: int = 300, n_mels: int = 80):
    """Second, independent restatement in float32 through ``torch.stft`` (CPU)."""
    import torch
    xt = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    spec = torch.stft(xt, n_fft, hop_length, n_fft, torch.hann_window(n_fft),
                      center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    power = spec.abs().pow(2)
    fb = torch.as_tensor(mel_filterbank(n_fft // 2 + 1, 0.0, float(sample_rate // 2),
                                        n_mels, sample_rate), dtype=torch.float32)
    return (fb.T @ power).numpy()
