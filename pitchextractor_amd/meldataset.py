"""Data layer drop-in for the reference's ``meldataset.py`` with the mel transform on the GPU.

Public surface kept: ``DEFAULT_MEL_PARAMS``, ``MelDataset`` (same constructor keywords,
``to_melspec``, ``mean/std = -4/4``, ``max_mel_length = 192``, ``path|label`` list lines),
``Collater`` and ``build_dataloader(path_list, validation, batch_size, num_workers, device,
collate_config, dataset_config)`` yielding ``(mels (B,1,80,192), f0s (B,192), is_silences (B,192))``.

What moved: in the reference every item runs the mel transform on the CPU inside a DataLoader
worker (meldataset.py:644).  Here workers only read audio and labels; the batch's raw audio goes
to HBM once and ONE launch of the fused mel kernel does framing + FFT + mel + log + the random
192-frame crop (meldataset.py:668-672, applied as a per-item frame offset) + zero padding
(meldataset.py:806-816).  All index arithmetic -- segment pre-crop (meldataset.py:178-201), F0
alignment (f0_backends.py:788-806), crop offsets -- is reproduced exactly and runs on the host.

Out of scope here: the F0 tracker backends and the WORLD / pitch-shift
augmentation need packages that are not installable offline.  F0 labels therefore come from the
reference's cache files under the reference's own contract (meldataset.py:519-604):
``<wav>_f0<cache_identifier>.npy`` validated by its sibling ``.json`` (cache_identifier, sample_rate,
hop_length), then the legacy ``<wav>_f0.npy``; or from an ``f0_provider`` callable.  A cache whose
metadata does not match is skipped with a warning and left on disk (the reference deletes and
recomputes it; this build cannot recompute).  An item with no usable label source fails loudly, as the
reference does when no backend is usable (meldataset.py:80-88).  ``<wav>_mel.npy`` caches
(meldataset.py:679-741) are honoured read-only under the reference's rule: whole-file items without
augmentation whose ``<wav>_mel_meta.json`` equals the expected metadata (audio/dataset sample rate, sample
count, channel count, mel_params) take their spectrogram from the cache -- normalised on the host with the
reference's own float32 expression, cropped, and written over that batch row after the mel launch -- so a
dataset trained with cached spectrograms keeps seeing exactly those values.  Caches are never written (the
device recomputes the mel in one launch per batch) and a mismatching cache is skipped with a warning, not
deleted (the reference clears every cache of the dataset on the first mismatch, :743-767).  Files at another sample rate are resampled on the GPU
(meldataset.py:621-627 -> ``pitchextractor_amd.resample.Resampler``) right before the mel launch; one
batch must come from one source rate.
"""
from __future__ import annotations

import json
import logging
import math
import os
import random
import re
import struct

import numpy as np
import torch
from torch.utils.data import DataLoader

from .mel import DEFAULT_MEL_PARAMS, MAX_MEL_LENGTH, MEL_MEAN, MEL_STD, LOG_EPS, MelSpectrogram
from .resample import Resampler

logger = logging.getLogger(__name__)
logger.setLevel(logging.DEBUG)

np.random.seed(1)      # meldataset.py:31-32
random.seed(1)


# --------------------------------------------------------------------------- host-side arithmetic
def align_length(values, target_frames: int) -> np.ndarray:
    """``F0Extractor.align_length`` (f0_backends.py:788-806): float64 linear interpolation to
    ``target_frames`` points, then zero each frame whose nearest (round-half-even) source frame is 0."""
    values = np.asarray(values, dtype=np.float64)
    if target_frames <= 0:
        return np.zeros((0,), dtype=np.float32)
    if values.size == target_frames:
        return values.astype(np.float32)
    if values.size == 0:
        return np.zeros((target_frames,), dtype=np.float32)
    last = values.size - 1
    grid = np.linspace(0.0, last, num=target_frames)
    out = np.interp(grid, np.linspace(0.0, last, num=values.size), values)
    zeros = values == 0.0
    if np.any(zeros):
        out[zeros[np.clip(np.round(grid).astype(int), 0, last)]] = 0.0
    return out.astype(np.float32)


def segment_plan(total_frames: int, source_sr: int, target_sr: int, hop: int, win: int, target_frames: int,
                 rng=random):
    """Pre-crop of meldataset.py:178-201 -> (start_frame, num_frames or None, use_full_file)."""
    if target_frames > 0 and source_sr and total_frames > 0:
        requested = (target_frames * hop) / float(target_sr) + max(win, hop) / float(target_sr)
        seg = int(np.ceil(requested * float(source_sr)))
        if seg > 0 and seg < total_frames:
            max_start = max(0, total_frames - seg)
            start = rng.randint(0, max_start) if max_start > 0 else 0
            return start, seg, False
        if seg > 0:
            return 0, seg, True
    return 0, None, True


def read_wav(path, start: int = 0, frames: int | None = None):
    """Minimal RIFF/WAVE reader (PCM 8/16/24/32-bit and IEEE float32/64) -> (float32 [n, ch] or [n], sr).
    Stands in for ``soundfile`` (absent from this image); uses it when importable."""
    try:
        import soundfile as sf  # pragma: no cover - optional
        with sf.SoundFile(path, mode="r") as f:
            if start:
                f.seek(int(start))
            data = f.read(frames=-1 if frames is None else int(frames), dtype="float32", always_2d=False)
            return np.asarray(data, dtype=np.float32), f.samplerate
    except ImportError:
        pass
    with open(path, "rb") as fh:
        riff, _, wave_id = struct.unpack("<4sI4s", fh.read(12))
        if riff != b"RIFF" or wave_id != b"WAVE":
            raise RuntimeError(f"Failed to load audio file '{path}': not a RIFF/WAVE file")
        fmt = None
        while True:
            head = fh.read(8)
            if len(head) < 8:
                raise RuntimeError(f"Failed to load audio file '{path}': no data chunk")
            cid, size = struct.unpack("<4sI", head)
            if cid == b"fmt ":
                raw = fh.read(size + (size & 1))
                tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", raw[:16])
                if tag == 0xFFFE and size >= 26:
                    tag = struct.unpack("<H", raw[24:26])[0]
                fmt = (tag, ch, sr, align, bits)
            elif cid == b"data":
                if fmt is None:
                    raise RuntimeError(f"Failed to load audio file '{path}': data before fmt")
                tag, ch, sr, align, bits = fmt
                total = size // align
                start = min(int(start or 0), total)
                n = total - start if frames is None else min(int(frames), total - start)
                fh.seek(start * align, 1)
                buf = fh.read(n * align)
                if tag == 3:
                    data = np.frombuffer(buf, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
                elif tag == 1 and bits == 16:
                    data = np.frombuffer(buf, dtype="<i2").astype(np.float32) / 32768.0
                elif tag == 1 and bits == 32:
                    data = np.frombuffer(buf, dtype="<i4").astype(np.float32) / 2147483648.0
                elif tag == 1 and bits == 8:
                    data = (np.frombuffer(buf, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
                elif tag == 1 and bits == 24:
                    b = np.frombuffer(buf, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
                    v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
                    data = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / 8388608.0
                else:
                    raise RuntimeError(f"Failed to load audio file '{path}': unsupported format {tag}/{bits}")
                return (data.reshape(-1, ch) if ch > 1 else data), sr
            else:
                fh.seek(size + (size & 1), 1)


def wav_info(path):
    """(frames, sample_rate, channels) without reading the samples."""
    with open(path, "rb") as fh:
        riff, _, wave_id = struct.unpack("<4sI4s", fh.read(12))
        if riff != b"RIFF" or wave_id != b"WAVE":
            raise RuntimeError("not a RIFF/WAVE file")
        fmt = None
        while True:
            head = fh.read(8)
            if len(head) < 8:
                raise RuntimeError("no data chunk")
            cid, size = struct.unpack("<4sI", head)
            if cid == b"fmt ":
                raw = fh.read(size + (size & 1))
                _, ch, sr, _, align, _ = struct.unpack("<HHIIHH", raw[:16])
                fmt = (ch, sr, align)
            elif cid == b"data":
                ch, sr, align = fmt
                return size // align, sr, ch
            else:
                fh.seek(size + (size & 1), 1)


# --------------------------------------------------------------------------- F0 cache naming
_BACKEND_TYPES = ("pyworld", "crepe", "swiftf0", "praat", "parselmouth")            # f0_backends.py:587-593
_DEFAULT_BACKENDS = [{"name": "pyworld_harvest", "type": "pyworld"}, {"name": "pyworld_dio", "type": "pyworld"}]


def _norm_backend(name) -> str:
    return re.sub(r"[^a-z0-9]+", "_", str(name).lower()).strip("_")               # f0_backends.py:596-597


def _enabled(value) -> bool:
    if isinstance(value, str):                                                      # f0_backends.py:619-631
        v = value.strip().lower()
        if not v or v in {"0", "false", "no", "off"}:
            return False
        if v in {"1", "true", "yes", "on"}:
            return True
    return bool(value)


def f0_cache_identifier(f0_params: dict | None) -> str:
    """The reference's ``F0Extractor.cache_identifier`` (f0_backends.py:661-757) for an ``f0_params``
    block: "-" + the cache keys of the enabled backends in chain order, joined by "_" (the shipped
    config.yml -> "-swiftf0").  The reference drops backends whose package is missing on the machine that
    computed the cache; set ``f0_params['cache_identifier']`` to name such a cache explicitly."""
    cfg = f0_params or {}
    if cfg.get("cache_identifier") is not None:
        return str(cfg["cache_identifier"])
    backends = cfg.get("backends") or {}
    if cfg.get("backend_order"):
        sequence = list(cfg["backend_order"])
    elif backends:
        sequence = list(backends.keys())
    else:
        sequence = [e["name"] for e in _DEFAULT_BACKENDS]
    defaults = {e["name"]: e for e in _DEFAULT_BACKENDS}
    keys = []
    for raw in sequence:
        if isinstance(raw, dict):
            entry = dict(raw)
        else:
            name = str(raw)
            bcfg, bkey = None, name
            if backends:
                if name in backends:
                    bcfg = backends[name]
                else:
                    for k, v in backends.items():
                        if _norm_backend(k) == _norm_backend(name):
                            bcfg, bkey = v, k
                            break
                if bcfg is None:
                    continue                                   # declared order names an unconfigured backend
            entry = {**defaults.get(bkey, defaults.get(name, {"name": name, "type": name})), **(bcfg or {})}
            entry.setdefault("name", bkey or name)
            entry.setdefault("type", entry.get("backend", entry.get("type", name)))
        if not _enabled(entry.get("enabled", True)):
            continue
        if str(entry.get("type") or entry.get("backend") or "pyworld").lower() not in _BACKEND_TYPES:
            continue
        name = _norm_backend(entry.get("name") or entry.get("type") or "backend")
        bconf = entry.get("config") or {k: v for k, v in entry.items()
                                        if k not in {"name", "type", "backend", "enabled"}}
        suffix = bconf.get("cache_key_suffix") if isinstance(bconf, dict) else None
        keys.append(_norm_backend(f"{name}-{suffix}" if suffix else name))
    return ("-" + "_".join(keys)) if keys else ""


# --------------------------------------------------------------------------- dataset
class MelDataset(torch.utils.data.Dataset):
    def __init__(self, data_list, sr=DEFAULT_MEL_PARAMS["sample_rate"], mel_params=None, f0_params=None,
                 data_augmentation=False, validation=False, verbose=True, synthetic_data=None,
                 f0_provider=None):
        self.verbose = verbose
        self.data_list = [line[:-1].split("|")[0] for line in data_list]        # meldataset.py:55-56

        mel_params = dict(mel_params or {})
        if "win_len" in mel_params and "win_length" not in mel_params:
            mel_params["win_length"] = mel_params.pop("win_len")                # meldataset.py:59-60
        self.mel_params = DEFAULT_MEL_PARAMS.copy()
        self.mel_params.update(mel_params)
        self.sr = sr if sr is not None else self.mel_params.get("sample_rate", DEFAULT_MEL_PARAMS["sample_rate"])
        self.mel_params["sample_rate"] = self.sr
        if self.verbose:
            print(f"[MelDataset] Using mel-spectrogram parameters: {self.mel_params}")
        self.to_melspec = MelSpectrogram(**self.mel_params)                     # HIP transform, (N,) -> (80, L)

        self.f0_params = f0_params or {}
        self.f0_provider = f0_provider
        self.f0_cache_identifier = f0_cache_identifier(self.f0_params)
        self.f0_cache_suffix = f"_f0{self.f0_cache_identifier}.npy"               # meldataset.py:91-92
        self.f0_meta_suffix = self.f0_cache_suffix.replace(".npy", ".json")
        self.mean, self.std = MEL_MEAN, MEL_STD
        self.data_augmentation = data_augmentation and (not validation)
        self.max_mel_length = MAX_MEL_LENGTH
        self.zero_value = float(self.f0_params.get("zero_fill_value", 0.0))
        self.bad_F0 = int(self.f0_params.get("bad_f0_threshold", 5))
        self.requires_cuda_backend = False
        self._audio_metadata_cache = {}
        self._invalid_paths = set()
        self._mel_cache_suffix, self._mel_meta_suffix = "_mel.npy", "_mel_meta.json"      # meldataset.py:102-103
        self._cache_enabled = True
        self._mel_cache_warned = False
        if synthetic_data and synthetic_data.get("enabled", False) and not validation:
            logger.warning("synthetic_data augmentation (WORLD / pitch-shift) needs pyworld/librosa, which are "
                           "not available: disabled")
        self.synthetic_enabled = False

    def __len__(self):
        return len(self.data_list)

    # ---- labels ---------------------------------------------------------------------------
    def _f0_cache_paths(self, path):
        return path + self.f0_cache_suffix, path + self.f0_meta_suffix, path + "_f0.npy"

    def _load_cached_f0(self, path):
        """meldataset.py:566-604: the identifier-named cache if its .json agrees on (cache_identifier,
        sample_rate, hop_length); else the legacy ``_f0.npy``; else None.  Nothing is deleted."""
        data_path, meta_path, legacy_path = self._f0_cache_paths(path)
        if os.path.isfile(data_path) and data_path != legacy_path:
            metadata = None
            if os.path.isfile(meta_path):
                try:
                    with open(meta_path, "r", encoding="utf-8") as fh:
                        metadata = json.load(fh)
                except (OSError, json.JSONDecodeError):
                    metadata = None
            expected = {"cache_identifier": self.f0_cache_identifier, "sample_rate": int(self.sr),
                        "hop_length": int(self.mel_params["hop_length"])}
            if metadata and all(metadata.get(k) == v for k, v in expected.items()):
                try:
                    return np.load(data_path).astype(np.float32)
                except (OSError, ValueError):
                    logger.warning("[MelDataset] unreadable F0 cache %s: skipped", data_path)
            else:
                logger.warning("[MelDataset] F0 cache %s %s: skipped (the reference would recompute it)", data_path,
                               "has no readable metadata" if not metadata else
                               f"was computed for {({k: metadata.get(k) for k in expected})}, expected {expected}")
        if os.path.isfile(legacy_path):
            try:
                return np.load(legacy_path).astype(np.float32)
            except (OSError, ValueError):
                logger.warning("[MelDataset] unreadable legacy F0 cache %s: skipped", legacy_path)
        return None

    def _f0_for(self, path, waveform, start_sample, expected_frames):
        cached = self._load_cached_f0(path)
        if cached is not None:
            if expected_frames is None:
                return cached
            hop = max(int(self.mel_params["hop_length"]), 1)
            lo = max(0, int(math.floor(start_sample / float(hop))))              # meldataset.py:532-537
            if lo >= cached.shape[0]:
                return np.zeros((0,), dtype=np.float32)
            return cached[lo:min(cached.shape[0], lo + int(expected_frames) + 4)]
        if self.f0_provider is not None:
            return np.asarray(self.f0_provider(path, waveform, self.sr), dtype=np.float32)
        raise RuntimeError(f"no F0 labels for {path}: no valid '{os.path.basename(path)}{self.f0_cache_suffix}' "
                           "(+ .json) cache, no legacy '_f0.npy' and no f0_provider (the reference's tracker "
                           "backends are outside this build)")

    # ---- cached spectrograms (meldataset.py:679-741), read-only ------------------------------
    def _build_mel_metadata(self, num_samples: int, wave_sr: int) -> dict:
        """meldataset.py:679-701 for the mono float waveform handed to ``_build_training_example``."""
        def plain(v):
            if isinstance(v, np.ndarray):
                return v.tolist()
            if isinstance(v, np.generic):
                return v.item()
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu()
                return v.item() if v.numel() == 1 else v.tolist()
            return v
        return {"audio_sample_rate": int(wave_sr), "audio_num_samples": int(num_samples), "audio_num_channels": 1,
                "dataset_sample_rate": int(self.sr), "mel_params": {k: plain(v) for k, v in self.mel_params.items()}}

    def _mel_cache_paths(self, path):
        return path + self._mel_cache_suffix, path + self._mel_meta_suffix

    def _load_cached_mel(self, path, expected_metadata):
        """meldataset.py:706-741: the cached power-mel (n_mels, L) if its metadata file equals
        ``expected_metadata``; None otherwise.  Nothing is deleted."""
        if not self._cache_enabled or self.data_augmentation:
            return None
        mel_path, meta_path = self._mel_cache_paths(path)
        if not os.path.isfile(mel_path):
            return None
        why = None
        if not os.path.isfile(meta_path):
            why = "has no metadata file"
        else:
            try:
                with open(meta_path, "r", encoding="utf-8") as fh:
                    cached = json.load(fh)
            except (OSError, json.JSONDecodeError):
                cached, why = None, "has unreadable metadata"
            if why is None and cached != expected_metadata:
                why = "was computed for other audio / mel parameters"
        if why is None:
            try:
                mel = np.load(mel_path)
                if mel.ndim == 2 and mel.shape[0] == int(self.mel_params["n_mels"]):
                    return np.ascontiguousarray(mel, dtype=np.float32)
                why = f"has shape {mel.shape}"
            except (OSError, ValueError):
                why = "is unreadable"
        if not self._mel_cache_warned:          # once per dataset, like the reference's one-shot invalidation
            self._mel_cache_warned = True
            logger.warning("[MelDataset] mel cache %s %s: skipped, the device recomputes it (the reference would clear "
                           "the dataset's caches here)", mel_path, why)
        return None

    # ---- one item -------------------------------------------------------------------------
    def _metadata(self, path):
        md = self._audio_metadata_cache.get(path)
        if md is None:
            frames, sr, ch = wav_info(path)
            md = {"frames": frames, "sample_rate": sr, "channels": ch}
            self._audio_metadata_cache[path] = md
        return md

    def path_to_wave_and_label(self, path):
        """Everything of meldataset.py:178-245 + :629-677 except the mel transform itself.
        Returns (waveform f32 (N,), f0 (L,), is_silence (L,), crop_start) with L = min(mel_len, 192)."""
        md = self._metadata(path)
        hop = int(self.mel_params["hop_length"])
        win = int(self.mel_params.get("win_length") or self.mel_params.get("n_fft", hop))
        start, seg, full = segment_plan(int(md["frames"]), md["sample_rate"], self.sr, hop, win,
                                        int(self.max_mel_length))
        wave, wave_sr = read_wav(path, start, seg)
        if wave.ndim > 1:
            wave = np.mean(wave, axis=-1)
        wave = wave.astype(np.float32)
        # the device resamples; everything below only needs the resampled LENGTH: ceil(new * L / orig)
        n_target = Resampler(wave_sr, self.sr).out_len(len(wave)) if wave_sr != self.sr else len(wave)
        start_sample = 0 if full else int(round(start / float(md["sample_rate"]) * self.sr))
        expected = None if full else int(np.ceil(n_target / max(hop, 1))) + 2
        f0 = self._f0_for(path, wave, start_sample, expected)
        if self.data_augmentation:
            wave = (0.5 + 0.5 * np.random.random()) * wave                        # meldataset.py:232-234
            wave = wave.astype(np.float32)
        mel_len = 1 + n_target // hop
        # meldataset.py:236-237,640-642: whole-file items without augmentation may come from the spectrogram cache
        self._last_cached_mel = None
        if full and not self.data_augmentation:
            cached = self._load_cached_mel(path, self._build_mel_metadata(n_target, self.sr))
            if cached is not None:
                mel_len = cached.shape[1]                                         # meldataset.py:651
                self._last_cached_mel = cached
        f0 = align_length(f0, mel_len)
        sil = (f0 == 0).astype(np.float32)
        crop = 0
        if mel_len > self.max_mel_length:
            crop = int(np.random.randint(0, mel_len - self.max_mel_length))       # meldataset.py:668-672
            f0 = f0[crop:crop + self.max_mel_length]
            sil = sil[crop:crop + self.max_mel_length]
        f0 = np.where(np.isnan(f0), np.float32(self.zero_value), f0).astype(np.float32)
        self._last_sr = wave_sr
        if self._last_cached_mel is not None:
            # meldataset.py:650,668-670 in the reference's own float32 torch arithmetic, on the host
            m = (torch.log(LOG_EPS + torch.from_numpy(self._last_cached_mel)) - self.mean) / self.std
            self._last_cached_mel = m[:, crop:crop + self.max_mel_length].contiguous()
        return wave, f0, sil, crop

    def __getitem__(self, idx):
        total = len(self.data_list)
        if total == 0:
            raise IndexError("MelDataset is empty")
        for attempt in range(total):
            path = self.data_list[(idx + attempt) % total]
            if path in self._invalid_paths:
                continue
            try:
                wave, f0, sil, crop = self.path_to_wave_and_label(path)
            except (FileNotFoundError, RuntimeError, OSError, ValueError, struct.error) as exc:
                if isinstance(exc, NotImplementedError):
                    raise
                self._invalid_paths.add(path)
                logger.warning("[MelDataset] Skipping unreadable audio file: %s (%s)", path, exc)
                continue
            item = (torch.from_numpy(wave), torch.from_numpy(f0), torch.from_numpy(sil), crop, int(self._last_sr))
            return item if self._last_cached_mel is None else item + (self._last_cached_mel,)
        raise RuntimeError("No valid audio files could be loaded from the dataset")

    def path_to_mel_and_label(self, path, device="cuda"):
        """Reference-shaped single item: (mel (80, L<=192) normalised log-mel on the device, f0, is_silence)."""
        wave, f0, sil, crop = self.path_to_wave_and_label(path)
        if self._last_cached_mel is not None:
            return self._last_cached_mel.to(device), torch.from_numpy(f0), torch.from_numpy(sil)
        wave_dev = torch.from_numpy(wave).to(device)
        if self._last_sr != self.sr:
            wave_dev = Resampler(self._last_sr, self.sr)(wave_dev)
        mel = self.to_melspec(wave_dev)
        mel = (torch.log(LOG_EPS + mel) - self.mean) / self.std
        return mel[:, crop:crop + self.max_mel_length], torch.from_numpy(f0), torch.from_numpy(sil)


class Collater(object):
    """Zero-pads items to 192 frames (meldataset.py:790-826).

    Accepts the reference's ``(mel (80,L), f0, is_silence)`` items, or this build's raw-audio items
    ``(wave (N,), f0, is_silence, crop_start[, source_sr[, cached_mel (80,L<=192)]])``; for the latter it returns
    host tensors ``(waves (B,Nmax), lengths, crop_starts, f0s, is_silences, source_sr)`` for the device mel stage,
    followed by ``(cached_rows (K,), cached_mels (K,80,192))`` when any item carries a cached spectrogram."""

    def __init__(self, return_wave=False):
        self.return_wave = return_wave
        self.min_mel_length = MAX_MEL_LENGTH
        self.max_mel_length = MAX_MEL_LENGTH

    def __call__(self, batch):
        B = len(batch)
        L = self.max_mel_length
        f0s = torch.zeros((B, L)).float()
        sils = torch.zeros((B, L)).float()
        if len(batch[0]) == 3:
            n_mels = batch[0][0].size(0)
            mels = torch.zeros((B, n_mels, L), dtype=torch.float32, device=batch[0][0].device)
            for i, (mel, f0, sil) in enumerate(batch):
                n = mel.size(1)
                mels[i, :, :n] = mel
                f0s[i, :n] = f0
                sils[i, :n] = sil
            return mels.unsqueeze(1), f0s, sils
        n_max = max(int(item[0].shape[0]) for item in batch)
        waves = torch.zeros((B, n_max), dtype=torch.float32)
        lengths = torch.zeros((B,), dtype=torch.int32)
        crops = torch.zeros((B,), dtype=torch.int32)
        rates = {int(item[4]) for item in batch if len(item) > 4}
        if len(rates) > 1:
            raise RuntimeError(f"one batch mixes source sample rates {sorted(rates)}: group files by rate")
        for i, item in enumerate(batch):
            wave, f0, sil, crop = item[:4]
            n = wave.shape[0]
            waves[i, :n] = wave
            lengths[i] = n
            crops[i] = int(crop)
            f0s[i, :f0.shape[0]] = f0
            sils[i, :sil.shape[0]] = sil
        out = (waves, lengths, crops, f0s, sils, (rates.pop() if rates else 0))
        rows = [i for i, item in enumerate(batch) if len(item) > 5]
        if rows:                                       # normalised, cropped cache rows, zero-padded like :812-816
            cached = torch.zeros((len(rows), batch[rows[0]][5].shape[0], L), dtype=torch.float32)
            for k, i in enumerate(rows):
                m = batch[i][5]
                cached[k, :, :m.shape[1]] = m
            out += (torch.tensor(rows, dtype=torch.int64), cached)
        return out


class H2DPrefetcher:
    """Host->device copies on a side HIP stream so the next batch's raw audio (49 MB at B = 256) crosses PCIe
    underneath the current step.  ``submit`` starts the copies of a tuple of (pinned) host tensors and returns a
    ticket; ``acquire`` makes the compute stream wait for that ticket only.  ``stream``: issue the copies from an
    existing stream instead of a private one -- data-parallel runs pass the model's weight-gradient side stream, which
    is idle during the forward pass when the next batch is submitted, because those runs cap the HIP hardware queues
    at three (compute, side, RCCL) and a fourth stream would share one of them (+1.7 ms per step, measured)."""

    def __init__(self, device, stream=None):
        self.device = torch.device(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)

    def submit(self, host_items):
        with torch.cuda.stream(self.stream):
            dev = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in host_items)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return dev, ready

    def acquire(self, ticket):
        dev, ready = ticket
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for t in dev:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)              # allocated on the side stream, consumed on the compute stream
        return dev


class DeviceMelLoader:
    """Iterates a host DataLoader of raw-audio batches and yields the reference's batch tuple
    ``(mels (B,1,80,192), f0s, is_silences)`` with the mel computed on the GPU in one launch."""

    def __init__(self, loader: DataLoader, mel: MelSpectrogram, device):
        self.loader, self.mel, self.device = loader, mel, torch.device(device)
        self.dataset = loader.dataset
        self._resamplers = {}
        self._h2d = None

    def __len__(self):
        return len(self.loader)

    def set_epoch(self, epoch: int):
        """Data-parallel runs: re-seed the cross-rank permutation (distributed.EpochShardSampler)."""
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)

    def _finish(self, ticket):
        waves, lengths, crops, f0s, sils, src_sr, *cached = self._h2d.acquire(ticket)
        if src_sr and src_sr != self.mel.sample_rate:
            rs = self._resamplers.setdefault(src_sr, Resampler(src_sr, self.mel.sample_rate))
            waves = rs(waves)                                 # zero-padded rows resample exactly like each item alone
            lengths = torch.tensor([rs.out_len(int(n)) for n in self._host_lengths.pop(0)], dtype=torch.int32,
                                   device=self.device)
        else:
            self._host_lengths.pop(0)
        mels = self.mel.log_mel_ragged(waves, lengths, crops, max_frames=MAX_MEL_LENGTH)
        if cached:                                            # rows whose spectrogram came from <wav>_mel.npy
            rows, cached_mels = cached
            mels[:, 0].index_copy_(0, rows, cached_mels)
        return mels, f0s, sils

    def __iter__(self):
        """Double-buffered: batch k+1's H2D is in flight on the side stream while batch k is being consumed."""
        if self._h2d is None:
            from . import distributed as pdist
            shared = None
            if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                    (torch.distributed.get_world_size() > 1 or pdist.rehearse_single_rank()):
                from .model import _side_stream
                shared = _side_stream(self.device)
            self._h2d = H2DPrefetcher(self.device, stream=shared)
        self._host_lengths = []
        pending = None
        for host in self.loader:
            self._host_lengths.append(host[1].tolist())
            ticket = self._h2d.submit(host)
            if pending is not None:
                yield self._finish(pending)
            pending = ticket
        if pending is not None:
            yield self._finish(pending)


def build_dataloader(path_list, validation=False, batch_size=4, num_workers=1, device="cpu", collate_config=None,
                     dataset_config=None, shard=None):
    """Reference signature (meldataset.py:829-875) plus ``shard=(rank, world, seed)`` for data-parallel runs:
    the loader then draws its indices from ``distributed.EpochShardSampler`` (equal contiguous per-rank shards
    of a per-epoch permutation) instead of the single-process shuffle."""
    dataset_config = dict(dataset_config or {})
    dataloader_options = dataset_config.pop("dataloader", {}) or {}
    if torch.device(device).type != "cuda":
        raise RuntimeError("build_dataloader (HIP path): device must be a HIP ('cuda') device; the mel stage has "
                           "no CPU fallback")
    dataset = MelDataset(path_list, validation=validation, **dataset_config)
    collate_fn = Collater(**(collate_config or {}))
    kwargs = dict(batch_size=batch_size, shuffle=(not validation), num_workers=num_workers,
                  drop_last=(not validation), collate_fn=collate_fn, pin_memory=True)
    if shard is not None:
        from .distributed import EpochShardSampler
        rank, world, seed = shard
        kwargs["sampler"] = EpochShardSampler(len(dataset), batch_size, rank, world, seed=seed,
                                              shuffle=(not validation), drop_last=(not validation))
        kwargs["shuffle"] = False
    start_method = dataloader_options.get("start_method")
    if start_method and num_workers > 0:
        kwargs["multiprocessing_context"] = torch.multiprocessing.get_context(start_method)
    if dataloader_options.get("persistent_workers") is not None and num_workers > 0:
        kwargs["persistent_workers"] = bool(dataloader_options["persistent_workers"])
    if dataloader_options.get("prefetch_factor") is not None and num_workers > 0:
        kwargs["prefetch_factor"] = int(dataloader_options["prefetch_factor"])
    return DeviceMelLoader(DataLoader(dataset, **kwargs), dataset.to_melspec, device)
