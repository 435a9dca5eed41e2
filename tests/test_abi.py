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
