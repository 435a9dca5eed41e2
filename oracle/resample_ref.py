"""Float64 oracle of the resampler (TEST INFRASTRUCTURE ONLY).

Restates the published algorithm of ``torchaudio.functional.resample`` (defaults
``resampling_method="sinc_interp_hann"``, ``lowpass_filter_width=6``, ``rolloff=0.99``), which the
reference calls at meldataset.py:621-627.  torchaudio is not installed and the reference holds no
vectors for it: PARITY UNPINNED by the reference; pinned here by construction properties (DC gain,
band-limited sine reconstruction, identity when the rates agree).
"""
import math

import numpy as np


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = np.arange(-width, width + orig, dtype=np.float64) / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx[None, :]) * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * (base_freq / orig)
    return kernels, width, orig, new                      # (new, 2*width + orig)


def resample(x, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    x = np.asarray(x, dtype=np.float64)
    if orig_freq == new_freq:
        return x
    k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    n = x.shape[-1]
    xp = np.pad(x, (width, width + orig))
    n_blocks = (xp.shape[0] - k.shape[1]) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(xp, k.shape[1])[::orig][:n_blocks]    # (blocks, taps)
    out = (frames @ k.T).reshape(-1)                                                        # block-major, phase-minor
    return out[:math.ceil(new * n / orig)]
