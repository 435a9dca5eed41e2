"""On-device resampler: ``torchaudio.functional.resample(x, orig, new)`` with its defaults
(sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), as the reference uses it at
meldataset.py:621-627.  (B, N) or (N,) float32 device audio in, ceil(new*N/orig) samples out."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, ops


class Resampler:
    def __init__(self, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.lowpass_filter_width, self.rolloff = int(lowpass_filter_width), float(rolloff)
        g = math.gcd(self.orig_freq, self.new_freq)
        self._orig, self._new = self.orig_freq // g, self.new_freq // g
        self._plan, self._device = None, None

    def out_len(self, n_in: int) -> int:
        return -(-int(n_in) * self._new // self._orig)

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_plan"], st["_device"] = None, None
        return st

    def _get_plan(self, device):
        if self._plan is None or self._device != device:
            handle = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(_lib.load().pe_resample_plan_create(C.byref(handle), self.orig_freq, self.new_freq,
                                                               self.lowpass_filter_width, self.rolloff),
                           "pe_resample_plan_create")
            self._plan, self._device = handle, device
        return self._plan

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        if not wave.is_cuda or wave.dtype != torch.float32:
            raise RuntimeError("Resampler (HIP) needs float32 device audio; no CPU fallback exists")
        if self.orig_freq == self.new_freq:
            return wave
        single = wave.dim() == 1
        x = wave.unsqueeze(0) if single else wave
        if x.stride(-1) != 1:
            x = x.contiguous()
        n_out = self.out_len(x.shape[1])
        y = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ops._call("pe_resample_forward", self._get_plan(x.device), x.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                      y.data_ptr(), y.stride(0), n_out, _lib.stream_ptr())
        return y[0] if single else y
