"""Synthetic 24 kHz utterances for benchmarks and parity tests.

Signal model follows the reference's sine-glide test-signal tool
(Utils/dynamic_pitch_tools.py:11-62: linear F0 glide, phase = cumsum(2 pi f0 / sr),
amplitude 0.8, 20 ms raised-cosine fades, peak-normalise above 0.99) and its
frame-rate label sampler (``sample_reference_f0``, :65-76).  SURVEY.md 8(d)
fixes the distribution: utterance i uses ``default_rng(1234 + i)``, start
frequency U(60, 250) Hz, end frequency U(120, 500) Hz, and a deterministic
unvoiced gap (f0 = 0) of 10-30 frames so the voicing target is not constant.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 1234


def raised_cosine_fade(n_samples: int, sr: int, fade_time: float = 0.02) -> np.ndarray:
    """float64 gain envelope with a half-cosine ramp at both ends."""
    env = np.ones(n_samples, dtype=np.float64)
    k = int(max(fade_time * sr, 0))
    if k > 0:
        ramp = 0.5 - 0.5 * np.cos(np.linspace(0.0, np.pi, k, dtype=np.float64))
        env[:k] = ramp
        env[-k:] = ramp[::-1]
    return env


def sine_from_f0(f0_curve: np.ndarray, sr: int, amplitude: float = 0.8) -> np.ndarray:
    """Sinusoid following ``f0_curve`` (Hz per sample); float32 out."""
    phase = np.cumsum(2.0 * np.pi * np.asarray(f0_curve, dtype=np.float64) / float(sr))
    audio = (amplitude * np.sin(phase)).astype(np.float32)
    audio = (audio * raised_cosine_fade(audio.shape[0], sr)).astype(np.float32)
    peak = float(np.max(np.abs(audio))) if audio.size else 0.0
    if peak > 0.99:
        audio = audio / (peak + 1e-6)
    return audio.astype(np.float32)


def glide(duration: float, start_hz: float, end_hz: float, sr: int):
    """(audio f32, time axis f32, f0 curve f32) of a linear glide."""
    n = int(duration * sr)
    t = np.linspace(0.0, duration, n, endpoint=False, dtype=np.float64)
    f0 = np.linspace(start_hz, end_hz, n, dtype=np.float64)
    return sine_from_f0(f0, sr), t.astype(np.float32), f0.astype(np.float32)


def frame_rate_f0(time_axis: np.ndarray, f0_curve: np.ndarray, num_frames: int) -> np.ndarray:
    """Analytic F0 sampled at ``num_frames`` evenly spaced frame times."""
    if num_frames <= 0:
        return np.zeros((0,), dtype=np.float32)
    if time_axis.size == 0:
        return np.zeros((num_frames,), dtype=np.float32)
    total = time_axis[-1]
    if time_axis.size > 1:
        total += time_axis[1] - time_axis[0]
    at = np.linspace(0.0, total, num=num_frames, endpoint=False, dtype=np.float64)
    return np.interp(at, time_axis, f0_curve).astype(np.float32)


def utterance(index: int, duration: float = 2.0, sr: int = 24000, hop: int = 300):
    """Utterance ``index`` of the benchmark set: (audio (N,), f0 (L,), is_silence (L,))."""
    rng = np.random.default_rng(SEED_BASE + int(index))
    f_a = rng.uniform(60.0, 250.0)
    f_b = rng.uniform(120.0, 500.0)
    audio, t, curve = glide(duration, f_a, f_b, sr)
    n_frames = 1 + audio.shape[0] // hop
    f0 = frame_rate_f0(t, curve, n_frames)
    gap_len = int(rng.integers(10, 31))
    gap_at = int(rng.integers(0, max(n_frames - gap_len, 1)))
    f0[gap_at:gap_at + gap_len] = 0.0
    # silence the audio under the gap so label and signal agree
    lo, hi = gap_at * hop, min((gap_at + gap_len) * hop, audio.shape[0])
    audio = audio.copy()
    audio[lo:hi] = 0.0
    sil = (f0 == 0).astype(np.float32)
    return audio, f0, sil


def batch(start: int, count: int, duration: float = 2.0, sr: int = 24000, hop: int = 300,
          max_frames: int = 192):
    """``count`` utterances -> (waves (count, N) f32, f0 (count, max_frames), sil (count, max_frames)).

    Labels are zero-padded to ``max_frames`` exactly as the reference's Collater pads
    (meldataset.py:806-816): padded frames carry f0 = 0 and is_silence = 0.
    """
    n = int(duration * sr)
    waves = np.zeros((count, n), dtype=np.float32)
    f0s = np.zeros((count, max_frames), dtype=np.float32)
    sils = np.zeros((count, max_frames), dtype=np.float32)
    for i in range(count):
        a, f0, sil = utterance(start + i, duration, sr, hop)
        L = min(f0.shape[0], max_frames)
        waves[i] = a
        f0s[i, :L] = f0[:L]
        sils[i, :L] = sil[:L]
    return waves, f0s, sils
