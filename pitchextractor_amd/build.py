"""In-tree build of the HIP/C-ABI library (gfx950 only).

``python -m pitchextractor_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU.  Objects are cached under ``csrc/_obj`` by source
mtime; the shared object lands next to the package so it travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = CSRC / "_obj"
LIB_PATH = PKG_DIR / "libpitchextractor_hip.so"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-gpu-rdc"]


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _headers():
    return sorted(CSRC.glob("*.h")) + sorted((PKG_DIR.parent / "include").glob("*.h"))


def _needs(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


# sources whose single-term ("mixed precision") MFMA pipelines are compiled a second time for fp16 operands
# (-DPE_F16_BUILD: same kernels, v_cvt_f16_f32 / v_mfma_f32_32x32x16_f16, exports pe_*_f16 only)
F16_SOURCES = ("gemm", "conv", "lstm", "lstm_persistent")


# per-source flags.  lstm_persistent: the cell / gate updates are issued in the shadow of MFMAs, where hipcc's SLP
# packing of adjacent scalar f32 operations into v_pk_* costs more issue time than it saves (MI355X guide, constants
# table; measured: forward item 11.4 -> 10.7 k cycles)
EXTRA_FLAGS = {"lstm_persistent": ["-fno-slp-vectorize"]}      # (gemm.hip / lstm.hip: no difference, measured)


def _compile(job, verbose: bool) -> Path:
    src, f16 = job
    obj = OBJ_DIR / (src.stem + ("_f16" if f16 else "") + ".o")
    if _needs(obj, [src] + _headers() + [Path(__file__)]):
        cmd = [HIPCC, *CXXFLAGS, *EXTRA_FLAGS.get(src.stem, []), *(["-DPE_F16_BUILD"] if f16 else []), "-c", str(src),
               "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return obj


def build_library(force: bool = False, verbose: bool = True, jobs: int = 4) -> Path:
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    if force:
        for o in OBJ_DIR.glob("*.o"):
            o.unlink()
    srcs = _sources()
    if not srcs:
        raise RuntimeError(f"no .hip sources under {CSRC}")
    work = [(s, False) for s in srcs] + [(s, True) for s in srcs if s.stem in F16_SOURCES]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda j: _compile(j, verbose), work))
    if force or _needs(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc",
               "-o", str(LIB_PATH), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
