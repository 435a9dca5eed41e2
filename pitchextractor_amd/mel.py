"""HIP mel front end: drop-in for ``torchaudio.transforms.MelSpectrogram`` as the
reference uses it (meldataset.py:34-40,58-77,644) plus the fused log/normalise of
meldataset.py:650, batched on the GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

DEFAULT_MEL_PARAMS = {
    "sample_rate": 24000,
    "n_mels": 80,
    "n_fft": 1024,
    "win_length": 1024,
    "hop_length": 300,
}

LOG_EPS = 1e-5
MEL_MEAN, MEL_STD = -4.0, 4.0
MAX_MEL_LENGTH = 192


class MelSpectrogram:
    """``transform(wave)``: (N,) -> (n_mels, 1 + N // hop) mel power, or (B, N) -> (B, n_mels, L).

    Same constructor keywords as the torchaudio transform the reference builds
    from ``mel_params``; everything not listed keeps torchaudio's defaults
    (centre reflect padding, periodic Hann, power 2, HTK scale, norm None).
    Inputs must live on a HIP device: there is no CPU path.
    """

    def __init__(self, sample_rate=24000, n_fft=1024, win_length=None, hop_length=None,
                 f_min=0.0, f_max=None, n_mels=80, **unsupported):
        for key, val in unsupported.items():
            raise NotImplementedError(f"MelSpectrogram option {key}={val!r} is not implemented on the HIP path")
        self.sample_rate = int(sample_rate)
        self.n_fft = int(n_fft)
        self.win_length = int(win_length) if win_length is not None else self.n_fft
        self.hop_length = int(hop_length) if hop_length is not None else self.win_length // 2
        self.n_mels = int(n_mels)
        self.f_min = float(f_min)
        self.f_max = float(f_max) if f_max is not None else float(self.sample_rate // 2)
        self._plan = None
        self._plan_device = None

    # the plan owns device tables; it is rebuilt lazily per process / device (picklable object)
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plan"] = None
        state["_plan_device"] = None
        return state

    def _get_plan(self, device: torch.device):
        if self._plan is None or self._plan_device != device:
            self.close()
            lib = _lib.load()
            handle = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.pe_mel_plan_create(C.byref(handle), self.sample_rate, self.n_fft,
                                                  self.win_length, self.hop_length, self.n_mels,
                                                  self.f_min, self.f_max), "pe_mel_plan_create")
            self._plan, self._plan_device = handle, device
        return self._plan

    def close(self):
        if getattr(self, "_plan", None) is not None:
            _lib.load().pe_mel_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_frames(self, n_samples: int) -> int:
        return 1 + int(n_samples) // self.hop_length

    def _run(self, waves: torch.Tensor, out: torch.Tensor, strides, out_frames, log_mode, pad_value):
        if not waves.is_cuda:
            raise RuntimeError("MelSpectrogram (HIP) needs a device tensor; no CPU fallback exists")
        if waves.dtype != torch.float32:
            raise TypeError("wave must be float32")
        if waves.stride(-1) != 1:
            waves = waves.contiguous()
        from . import ops
        plan = self._get_plan(waves.device)
        n_valid = min(self.num_frames(waves.shape[1]), out_frames)
        with torch.cuda.device(waves.device):
            # algorithmic bytes: 300 new samples read + n_mels floats written per frame
            ops._call("pe_mel_forward", plan, waves.data_ptr(), waves.shape[0], waves.shape[1],
                      waves.stride(0), out.data_ptr(), *strides, out_frames, log_mode, LOG_EPS, MEL_MEAN,
                      MEL_STD, pad_value, _lib.stream_ptr(),
                      work=float(waves.shape[0] * n_valid * 4 * (self.hop_length + self.n_mels)))
        return out

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        single = wave.dim() == 1
        waves = wave.unsqueeze(0) if single else wave
        if waves.dim() != 2:
            raise ValueError("expected (N,) or (B, N)")
        L = self.num_frames(waves.shape[1])
        out = torch.empty((waves.shape[0], self.n_mels, L), dtype=torch.float32, device=waves.device)
        self._run(waves, out, (out.stride(0), out.stride(1), out.stride(2)), L, 0, 0.0)
        return out[0] if single else out

    forward = __call__

    def log_mel_batch(self, waves: torch.Tensor, max_frames: int = MAX_MEL_LENGTH,
                      out: torch.Tensor | None = None) -> torch.Tensor:
        """(B, N) raw audio -> (B, 1, n_mels, max_frames) normalised log-mel, zero padded.

        One launch does what the reference spreads over ``to_melspec`` (meldataset.py:644),
        ``(log(1e-5 + mel) + 4) / 4`` (:650) and Collater's zero padding (:806-816).
        Utterances longer than ``max_frames`` frames are truncated on the right (the random
        crop of meldataset.py:668-672 is applied to the audio before this call).
        """
        B = waves.shape[0]
        if out is None:
            out = torch.empty((B, 1, self.n_mels, max_frames), dtype=torch.float32, device=waves.device)
        self._run(waves, out, (out.stride(0), out.stride(2), out.stride(3)), max_frames, 1, 0.0)
        return out

    def log_mel_ragged(self, waves: torch.Tensor, lengths: torch.Tensor, frame_start: torch.Tensor | None = None,
                       max_frames: int = MAX_MEL_LENGTH, out: torch.Tensor | None = None) -> torch.Tensor:
        """Ragged batch: ``waves`` (B, Nmax) zero-padded rows, ``lengths`` (B,) int32 sample counts,
        ``frame_start`` (B,) int32 first frame kept per item (meldataset.py:668-672 random crop).
        Returns (B, 1, n_mels, max_frames) normalised log-mel; frames past an item's end are 0."""
        from . import ops
        if not (waves.is_cuda and lengths.is_cuda and waves.dtype == torch.float32 and waves.dim() == 2):
            raise RuntimeError("log_mel_ragged needs float32 (B, N) device audio and device lengths")
        if lengths.dtype != torch.int32 or lengths.numel() != waves.shape[0] or not lengths.is_contiguous():
            raise ValueError("lengths must be a contiguous int32 tensor of size B")
        if frame_start is not None and (frame_start.dtype != torch.int32 or frame_start.numel() != waves.shape[0]
                                        or not frame_start.is_cuda or not frame_start.is_contiguous()):
            raise ValueError("frame_start must be a contiguous int32 device tensor of size B")
        if waves.stride(1) != 1:
            waves = waves.contiguous()
        B = waves.shape[0]
        if out is None:
            out = torch.empty((B, 1, self.n_mels, max_frames), dtype=torch.float32, device=waves.device)
        plan = self._get_plan(waves.device)
        with torch.cuda.device(waves.device):
            ops._call("pe_mel_forward_ragged", plan, waves.data_ptr(), B, waves.shape[1], waves.stride(0),
                      lengths.data_ptr(), _lib.ptr(frame_start), out.data_ptr(), out.stride(0), out.stride(2),
                      out.stride(3), max_frames, 1, LOG_EPS, MEL_MEAN, MEL_STD, 0.0, _lib.stream_ptr(),
                      work=float(B * max_frames * 4 * (self.hop_length + self.n_mels)))
        return out
