"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU needed)."""
import re
from pathlib import Path

import pytest

from pitchextractor_amd import _lib, build

ROOT = Path(__file__).resolve().parent.parent


def header_functions():
    text = (ROOT / "include" / "pitchextractor_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    build.build_library(verbose=False)
    return _lib.load()


def test_header_and_bindings_agree():
    assert header_functions() == sorted(_lib.PROTOTYPES)


def test_every_declared_symbol_is_exported(lib):
    for name in header_functions():
        assert hasattr(lib, name), name


def test_abi_version(lib):
    assert lib.pe_abi_version() >= 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_argument_checks_run_before_any_device_call(lib):
    """The entry points validate their arguments on the host (no GPU in this test): a leading dimension the 32-bit
    buffer offsets of the operand loaders cannot address, a 3x3 convolution whose tensor passes 2 GiB and a missing
    operand-scale word are refused with the documented codes instead of being launched."""
    import ctypes
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    word = ctypes.cast((ctypes.c_uint * 1)(), ctypes.c_void_p)
    unsupported, bad_arg = -2, -1
    codes = {c: getattr(_lib, c) for c in ("PE_E_UNSUPPORTED", "PE_E_ARG") if hasattr(_lib, c)}
    unsupported, bad_arg = codes.get("PE_E_UNSUPPORTED", unsupported), codes.get("PE_E_ARG", bad_arg)
    # NT: lda = 2^21 floats; TN: ldb = 2^24
    assert lib.pe_gemm_nt_h2(p, 1 << 21, p, 64, p, 64, 128, 64, 64, None, None, 0, word, word, None) == unsupported
    assert lib.pe_gemm_tn_h2(p, 64, p, 1 << 24, p, 64, 64, 64, 64, 0, None, 0, word, word, None) == unsupported
    # h2 without the scale words
    assert lib.pe_gemm_nt_h2(p, 64, p, 64, p, 64, 128, 64, 64, None, None, 0, None, None, None) == bad_arg
    # staged-window convolution: 2^31 bytes of input
    assert lib.pe_conv3x3_fwd_wf_h2(p, p, p, 4096, 192, 80, 64, 64, 0, None, None, word, word) == unsupported
