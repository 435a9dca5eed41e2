"""ctypes binding of libpitchextractor_hip.so (the C ABI in include/pitchextractor_hip.h).

There is deliberately no fallback: if the library is missing or a call fails the
caller gets an exception, never a silent eager/CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PITCHEXTRACTOR_HIP_LIB", _PKG / "libpitchextractor_hip.so"))

_lib = None


class HipLibraryError(RuntimeError):
    pass


_p = C.c_void_p
_i = C.c_int
_l = C.c_long
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes).  Mirrors include/pitchextractor_hip.h one to one;
# tests/test_abi.py checks the two against each other and against the .so.
PROTOTYPES = {
    "pe_abi_version": (_i, []),
    "pe_device_count": (_i, []),
    "pe_mel_plan_create": (_i, [C.POINTER(_p), _i, _i, _i, _i, _i, _f, _f]),
    "pe_mel_plan_destroy": (_i, [_p]),
    "pe_mel_num_frames": (_i, [_p, _i]),
    "pe_mel_forward": (_i, [_p, _p, _i, _i, _l, _p, _l, _l, _l, _i, _i, _f, _f, _f, _f, _p]),
}


def load():
    """Load the shared library once and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m pitchextractor_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status < 0:
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "workspace too small"}.get(status, "error")
        raise HipLibraryError(f"{what}: {kind} ({status})")
    raise HipLibraryError(f"{what}: hipError_t {status}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
