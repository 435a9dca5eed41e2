"""Tensor-level wrappers over the C ABI (include/pitchextractor_hip.h).

Every function checks operand shapes/devices on the host before a kernel is launched (an
out-of-bounds launch can reset the GPU), passes raw device pointers plus the current HIP
stream, and never falls back to a torch op.  Activations are channels-last:
``[B, T, F, C]`` float32 contiguous.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib

_WS: dict = {}


def _chk(cond, msg):
    if not cond:
        raise ValueError(msg)


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    _chk(t.is_cuda, f"{name}: expected a HIP device tensor (no CPU fallback)")
    _chk(t.dtype == torch.float32, f"{name}: expected float32")
    return t


def _dense(t: torch.Tensor, name: str) -> torch.Tensor:
    _f32c(t, name)
    _chk(t.is_contiguous(), f"{name}: expected a contiguous tensor")
    return t


# Mixed precision with bf16 ACTIVATION STORAGE (reference trainer.py:226-235 / README.md:36: autocast keeps conv and
# linear outputs in 16 bits): when True (Trainer's mixed-precision scope with the bf16 dtype turns it on) the first
# convolution writes a bf16 tensor and every pass of the conv stack follows the dtype of its inputs -- the `*_a16`
# entry points of the C ABI.  The temporal heads, the losses and all parameters stay fp32.
ACT_BF16 = False


def act_dtype():
    return torch.bfloat16 if (ACT_BF16 and MATMUL_BF16 and HALF_DTYPE == "bf16") else torch.float32


def _actc(t: torch.Tensor, name: str) -> torch.Tensor:
    """An activation tensor of the conv stack: float32, or bfloat16 under ``ACT_BF16``."""
    _chk(t.is_cuda, f"{name}: expected a HIP device tensor (no CPU fallback)")
    _chk(t.dtype in (torch.float32, torch.bfloat16), f"{name}: expected float32 or bfloat16")
    return t


def _actd(t: torch.Tensor, name: str) -> torch.Tensor:
    _actc(t, name)
    _chk(t.is_contiguous(), f"{name}: expected a contiguous tensor")
    return t


def _a16(*tensors) -> str:
    """"_a16" if the given activation tensors are bfloat16, "" if float32; mixing is an error."""
    kinds = {t.dtype for t in tensors if t is not None}
    _chk(len(kinds) == 1, "activation tensors of one call must share a dtype")
    return "_a16" if kinds.pop() == torch.bfloat16 else ""


def _rows2d(t: torch.Tensor, name: str, act=False):
    """(rows, cols, ld) of a 2-D row-major view whose rows may be strided (``act``: bf16 activations allowed)."""
    (_actc if act else _f32c)(t, name)
    _chk(t.dim() == 2 and t.stride(1) == 1, f"{name}: expected 2-D with unit column stride")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t.shape[0], t.shape[1], ld


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer (bytes) per device AND current stream.  Consumers on one stream run in order, so
    a single buffer serves all ops launched there; work on a side stream gets its own."""
    key = (torch.device(device), _lib.stream_ptr())
    buf = _WS.get(key)
    nbytes = max(int(nbytes), 1 << 20)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=key[0])
        _WS[key] = buf
    return buf


class KernelTimer:
    """Optional HIP-event timing of every C-ABI call (bench.py): events are recorded on the stream
    the kernels are launched on and read back after the timed region, so nothing synchronises."""

    def __init__(self, only=None):
        self.records = {}
        self.only = None if only is None else frozenset(only)     # time just these entry points (None = all)

    def add(self, name, start, end, work):
        self.records.setdefault(name, []).append((start, end, work))

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = dict(calls=len(recs), total_ms=float(sum(ms)), avg_ms=float(sum(ms) / len(ms)),
                             work=float(sum(w for _, _, w in recs)))
        return out


TIMER: KernelTimer | None = None


TIMER_TAG = ""            # appended as "#tag" to the timer key of the calls made inside `timer_tag(tag)`


class timer_tag:
    """Label the timed C-ABI calls of a region (bench.py separates the data-gradient launches of the conv kernel,
    which share the GPU with side-stream weight-gradient kernels, from its forward launches, which run alone)."""

    def __init__(self, tag):
        self.tag, self.prev = tag, ""

    def __enter__(self):
        global TIMER_TAG
        self.prev, TIMER_TAG = TIMER_TAG, self.tag

    def __exit__(self, *exc):
        global TIMER_TAG
        TIMER_TAG = self.prev


def _call(name, *args, work=0.0):
    lib = _lib.load()
    if TIMER is None or (TIMER.only is not None and name not in TIMER.only):
        _lib.check(getattr(lib, name)(*args), name)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(getattr(lib, name)(*args), name)
    b.record()
    TIMER.add(name + ("#" + TIMER_TAG if TIMER_TAG else ""), a, b, work)


def _s():
    return _lib.stream_ptr()


# Opt-in mixed precision (reference trainer.py:103 autocast; config training.mixed_precision): when True the
# conv / linear products and their weight gradients round their operands to bf16 on the way into LDS and
# accumulate in fp32; the persistent LSTM recurrences round W_hh and h to bf16 as well (cell state fp32);
# attention and everything in HBM stay fp32.  False is the parity mode.
MATMUL_BF16 = False
# 16-bit operand type of that mode: "bf16" (default: fp32 exponent range, no loss scaling needed) or "f16" (the
# reference's literal autocast dtype; Trainer(amp_dtype="fp16") drives it together with a GradScaler)
HALF_DTYPE = "bf16"
# How fp32 products run (always, for the weight-gradient products; when MATMUL_BF16 is off for the rest):
#   "x3"     every fp32 operand is split exactly into three bf16 terms and six cross products are
#            accumulated in fp32 on the bf16 MFMA pipe (16x the fp32 MFMA rate on gfx950); measured
#            error against fp64 is the same as the native path's (tests/test_ops_gpu.py).  Default.
#   "h2"     every fp32 operand becomes TWO fp16 terms of the tensor scaled by a power of two (absmax below) and
#            three cross products are accumulated in fp32: half the matrix work of "x3", per-product error
#            <= 2^-21 (csrc/gemm_engine.h).
#   "native" v_mfma_f32_32x32x2_f32.
FP32_MATMUL = os.environ.get("PE_FP32_MATMUL", "h2")
_FP32_MODES = ("native", "x3", "h2")


_UNIT_AMAX: dict = {}


def _unit_amax(device):
    """absmax word of a tensor known to lie in [-1, 1] (LSTM outputs): the bits of 1.0f, no pass over the data."""
    key = torch.device(device)
    if key not in _UNIT_AMAX:
        _UNIT_AMAX[key] = torch.full((1,), 0x3F800000, dtype=torch.int32, device=key)
    return _UNIT_AMAX[key]


def absmax_segments(flat, seg_off, seg_len, out=None):
    """out[s] = absmax word of flat[seg_off[s] : seg_off[s] + seg_len[s]] (int64 device tensors), one launch."""
    _dense(flat, "flat")
    n = seg_off.numel()
    _chk(seg_off.is_cuda and seg_len.is_cuda and seg_off.dtype == seg_len.dtype == torch.int64 and seg_len.numel() == n,
         "absmax_segments: int64 device offset / length arrays")
    if out is None:
        out = torch.empty((n,), dtype=torch.int32, device=flat.device)
    _chk(out.is_cuda and out.dtype == torch.int32 and out.numel() == n, "absmax_segments: out")
    _call("pe_absmax_segments", flat.data_ptr(), seg_off.data_ptr(), seg_len.data_ptr(), n, out.data_ptr(), _s())
    return out


def h2_active() -> bool:
    """True when fp32 products run as two scaled fp16 terms, i.e. when GEMM / conv operands need an absmax word."""
    return (not MATMUL_BF16) and FP32_MATMUL == "h2"


def amax_for(t):
    """absmax(t) in "h2" mode, else None: what model code hands on to the products that read ``t``."""
    return absmax(t) if h2_active() else None


_AMAX_POOL: dict = {}


def amax_word():
    """A zeroed int32 device word for a producing pass to max-merge its output's absmax into ("h2" mode; None
    otherwise).  Words are cut from a 256-word block zeroed by one fill; a block lives as long as any of its words."""
    if not h2_active():
        return None
    key = (torch.cuda.current_device(), _lib.stream_ptr())
    blk = _AMAX_POOL.get(key)
    if blk is None or blk[1] >= blk[0].numel():
        blk = [torch.zeros(256, dtype=torch.int32, device="cuda"), 0]
        _AMAX_POOL[key] = blk
    blk[1] += 1
    return blk[0][blk[1] - 1:blk[1]]


_BOUND_AMAX: dict = {}


def amax_bound(bound: float, device):
    """absmax word for a tensor known to satisfy |x| <= bound (no pass over the data); None outside "h2" mode."""
    if not h2_active():
        return None
    key = (torch.device(device), float(bound))
    if key not in _BOUND_AMAX:
        import struct
        bits = struct.unpack("<i", struct.pack("<f", float(bound)))[0]
        _BOUND_AMAX[key] = torch.full((1,), bits, dtype=torch.int32, device=key[0])
    return _BOUND_AMAX[key]


def absmax(t, out=None):
    """uint32 device word = IEEE bits of max |t| (t: float32, 1-D / 2-D row-strided / dense N-D): the scale source of
    an "h2" operand."""
    _f32c(t, "absmax operand")
    if t.dim() == 2 and t.stride(1) == 1:
        rows, cols, ld = _rows2d(t, "absmax operand")
    else:
        _chk(t.is_contiguous(), "absmax: expected a dense tensor or a row-strided matrix")
        cols = t.shape[-1] if t.dim() > 1 else t.numel()
        rows, ld = t.numel() // max(cols, 1), cols
    if out is None:
        out = torch.empty((1,), dtype=torch.int32, device=t.device)
    _chk(out.is_cuda and out.dtype == torch.int32 and out.numel() == 1, "absmax: out is an int32 device word")
    _call("pe_absmax", t.data_ptr(), rows, cols, ld, out.data_ptr(), _s())
    return out


def _tn_suffix():
    """Weight-gradient / k-major products: bf16 operands under mixed precision (what autocast's backward does),
    otherwise fp32-accurate (native or three-term split)."""
    _chk(FP32_MATMUL in _FP32_MODES, "ops.FP32_MATMUL must be 'native', 'x3' or 'h2'")
    if MATMUL_BF16:
        return "_" + HALF_DTYPE
    return "_" + FP32_MATMUL if FP32_MATMUL != "native" else ""


# which persistent LSTM recurrences use the three-term split when FP32_MATMUL == "x3" (tools/bench_lstm.py)
LSTM_X3 = {"fwd": os.environ.get("PE_LSTM_X3_FWD", "1") == "1", "bwd": os.environ.get("PE_LSTM_X3_BWD", "1") == "1"}


def _lstm_suffix(which):
    if MATMUL_BF16:
        return "_" + HALF_DTYPE        # mixed precision: 16-bit recurrent products (as autocast runs nn.LSTM)
    # "h2" products need a tensor-wide scale before the first element is produced; the recurrences make their
    # operands step by step, so they keep the three-term bf16 split in that mode
    return "_x3" if (FP32_MATMUL in ("x3", "h2") and LSTM_X3[which]) else ""


def _nt_suffix():
    if MATMUL_BF16:
        return "_" + HALF_DTYPE
    _chk(FP32_MATMUL in _FP32_MODES, "ops.FP32_MATMUL must be 'native', 'x3' or 'h2'")
    return "_" + FP32_MATMUL if FP32_MATMUL != "native" else ""


class matmul_bf16:
    """``with ops.matmul_bf16(True): ...`` -- the autocast-like scope Trainer.run opens for a mixed-precision step.
    ``act16=True`` (bf16 only) additionally stores the conv stack's activations and their gradients as bf16 tensors
    (``ACT_BF16``)."""

    def __init__(self, enabled=True, dtype="bf16", act16=False):
        _chk(dtype in ("bf16", "f16"), "matmul_bf16: dtype must be 'bf16' or 'f16'")
        self.enabled, self.dtype = bool(enabled), dtype
        self.act16 = bool(act16) and self.enabled and dtype == "bf16"

    def __enter__(self):
        global MATMUL_BF16, HALF_DTYPE, ACT_BF16
        self._prev = (MATMUL_BF16, HALF_DTYPE, ACT_BF16)
        MATMUL_BF16, HALF_DTYPE, ACT_BF16 = self.enabled, self.dtype, self.act16
        return self

    def __exit__(self, *exc):
        global MATMUL_BF16, HALF_DTYPE, ACT_BF16
        MATMUL_BF16, HALF_DTYPE, ACT_BF16 = self._prev
        return False


# ------------------------------------------------------------------ GEMM
def gemm_nt(A, B, bias0=None, bias1=None, out=None, accumulate=False, amax_a=None, amax_b=None):
    """out[M,N] = A[M,K] @ B[N,K]^T + bias0 + bias1 (+ out).  ``amax_*``: the operands' absmax words when the caller
    already has them ("h2" mode; computed here otherwise)."""
    M, K, lda = _rows2d(A, "A", act=True)
    N, K2, ldb = _rows2d(B, "B")
    _chk(K == K2, "gemm_nt: K mismatch")
    if out is None:
        _chk(not accumulate, "accumulate needs out")
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    Mo, No, ldc = _rows2d(out, "out", act=True)
    _chk((Mo, No) == (M, N), "gemm_nt: out shape")
    a16 = _a16(A, out)
    if a16:
        _chk(_nt_suffix() == "_bf16", "bfloat16 activations need the bf16 mixed-precision mode")
        _call("pe_gemm_nt_bf16_a16", A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
              _lib.ptr(bias0), _lib.ptr(bias1), int(bool(accumulate)), _s(), work=2.0 * M * N * K)
        return out
    for b in (bias0, bias1):
        if b is not None:
            _chk(_dense(b, "bias").numel() == N, "bias size")
    sfx = _nt_suffix()
    if sfx == "_h2":
        amax_a = absmax(A) if amax_a is None else amax_a
        amax_b = absmax(B) if amax_b is None else amax_b
        _call("pe_gemm_nt_h2", A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
              _lib.ptr(bias0), _lib.ptr(bias1), int(bool(accumulate)), amax_a.data_ptr(), amax_b.data_ptr(), _s(),
              work=2.0 * M * N * K)
        return out
    _call("pe_gemm_nt" + sfx, A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
          _lib.ptr(bias0), _lib.ptr(bias1), int(bool(accumulate)), _s(), work=2.0 * M * N * K)
    return out


def gemm_tn(A, B, out=None, accumulate=False, amax_a=None, amax_b=None):
    """out[M,N] = A[K,M]^T @ B[K,N] (+ out); deterministic split-K."""
    K, M, lda = _rows2d(A, "A", act=True)
    K2, N, ldb = _rows2d(B, "B", act=True)
    _chk(K == K2, "gemm_tn: K mismatch")
    if out is None:
        _chk(not accumulate, "accumulate needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    Mo, No, ldc = _rows2d(out, "out")
    _chk((Mo, No) == (M, N), "gemm_tn: out shape")
    lib = _lib.load()
    need = lib.pe_gemm_tn_workspace_bytes(M, N, K)
    ws = workspace(need, A.device)
    if _a16(A, B):
        _chk(_tn_suffix() == "_bf16", "bfloat16 activations need the bf16 mixed-precision mode")
        _call("pe_gemm_tn_bf16_a16", A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
              int(bool(accumulate)), ws.data_ptr(), ws.numel(), _s(), work=2.0 * M * N * K)
        return out
    if _tn_suffix() == "_h2":
        amax_a = absmax(A) if amax_a is None else amax_a
        amax_b = absmax(B) if amax_b is None else amax_b
        _call("pe_gemm_tn_h2", A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
              int(bool(accumulate)), ws.data_ptr(), ws.numel(), amax_a.data_ptr(), amax_b.data_ptr(), _s(),
              work=2.0 * M * N * K)
        return out
    _call("pe_gemm_tn" + _tn_suffix(), A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc, M, N, K,
          int(bool(accumulate)), ws.data_ptr(), ws.numel(), _s(), work=2.0 * M * N * K)
    return out


def transpose2d(x, out=None):
    x = _dense(x, "x")
    _chk(x.dim() == 2, "transpose2d: 2-D")
    if out is None:
        out = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float32, device=x.device)
    _chk(_dense(out, "out").numel() == x.numel(), "transpose2d: out size")
    _call("pe_transpose2d", x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _s())
    return out


# ------------------------------------------------------------------ conv
# x3 / bf16 modes: 3x3 weights are packed once per step into MFMA B-fragment order (three exact bf16 terms, or
# one rounded term) and the halo kernel loads them straight from L2 into registers (csrc/conv.hip).
CONV_WFRAG = os.environ.get("PE_CONV_WFRAG", "1") == "1"


class PackedWeight:
    """One packed 3x3 weight: ``fp32`` [N, 9*C] (native MFMA path, fallback shapes) and, in the x3 / h2 / bf16 / f16
    modes, ``frag`` = the same matrix as 16-bit MFMA fragments (``terms`` = 3 exact bf16 terms, 2 scaled fp16 terms
    with ``amax`` = the weight's absmax word, or 1 rounded term of ``half``)."""

    def __init__(self, fp32, frag=None, terms=0, half=None, amax=None):
        self.fp32, self.frag, self.terms, self.half, self.amax = fp32, frag, terms, half, amax

    @property
    def shape(self):
        return self.fp32.shape


def wfrag_pack(w2d, terms, amax=None):
    """[N, K] float32 (K % 16 == 0) -> fragment-ordered 16-bit terms (uint8 buffer); terms == 2 needs the weight's
    absmax word."""
    N, K, ld = _rows2d(w2d, "w")
    lib = _lib.load()
    nbytes = lib.pe_wfrag_bytes(N, K, terms)
    _chk(nbytes > 0, "wfrag_pack: K must be a multiple of 16 and terms 1, 2 or 3")
    out = torch.empty((nbytes,), dtype=torch.uint8, device=w2d.device)
    if terms == 2:
        _chk(amax is not None, "wfrag_pack: two-term fragments need amax")
        _call("pe_wfrag_pack_h2", w2d.data_ptr(), ld, N, K, amax.data_ptr(), out.data_ptr(), _s())
    elif terms == 1 and MATMUL_BF16 and HALF_DTYPE == "f16":
        _call("pe_wfrag_pack_f16", w2d.data_ptr(), ld, N, K, out.data_ptr(), _s())
    else:
        _call("pe_wfrag_pack", w2d.data_ptr(), ld, N, K, int(terms), out.data_ptr(), _s())
    return out


def _mode_terms():
    sfx = _nt_suffix()
    return 3 if sfx == "_x3" else 2 if sfx == "_h2" else 1 if sfx in ("_bf16", "_f16") else 0


def conv3x3_repack(w, want_fwd=True, want_dgrad=True, amax=None):
    """OIHW (Cout,Cin,3,3) -> (w_fwd [Cout, 9*Cin], w_dgrad [Cin, 9*Cout]) as ``PackedWeight``s.  ``amax``: the
    weight's absmax word when the caller has it ("h2" mode)."""
    w = _dense(w, "w")
    _chk(w.dim() == 4 and w.shape[2:] == (3, 3), "conv3x3_repack: OIHW 3x3")
    co, ci = w.shape[0], w.shape[1]
    wf = torch.empty((co, 9 * ci), dtype=torch.float32, device=w.device) if want_fwd else None
    wd = torch.empty((ci, 9 * co), dtype=torch.float32, device=w.device) if want_dgrad else None
    _call("pe_conv3x3_repack", w.data_ptr(), _lib.ptr(wf), _lib.ptr(wd), co, ci, _s())
    terms = _mode_terms() if CONV_WFRAG else 0
    if _mode_terms() != 2:
        amax = None
    elif amax is None:
        amax = absmax(w.view(co, ci * 9))                               # forward and data-gradient forms share it
    out = []
    for t in (wf, wd):
        if t is None:
            out.append(None)
        elif terms and t.shape[1] % 16 == 0:
            out.append(PackedWeight(t, wfrag_pack(t, terms, amax), terms, HALF_DTYPE if terms == 1 else None, amax))
        else:
            out.append(PackedWeight(t, amax=amax))
    return out[0], out[1]


def conv3x3_fwd(x, w_packed, out=None, accumulate=False, bn_stats=None, amax=None):
    """x [B,T,F,C], w_packed [N, 9*C] (tensor or PackedWeight) -> y [B,T,F,N] (+= when accumulate).
    ``amax``: x's absmax word when the caller has it ("h2" mode; computed here otherwise).
    ``bn_stats`` True / False (not None) returns (y, partials): with True the fragment-fed kernel leaves the BatchNorm
    column sums of its final outputs behind ([tiles, 2, N] float64, for ``bn_train_stats(..., partials=)``);
    partials is None when not asked for or when another kernel ran."""
    x = _actd(x, "x")
    pw = w_packed if isinstance(w_packed, PackedWeight) else PackedWeight(w_packed)
    w32 = _dense(pw.fp32, "w_packed")
    _chk(x.dim() == 4 and w32.dim() == 2, "conv3x3_fwd: ranks")
    B, T, F, Cc = x.shape
    N = w32.shape[0]
    _chk(w32.shape[1] == 9 * Cc, "conv3x3_fwd: weight K")
    if out is None:
        _chk(not accumulate, "accumulate needs out")
        out = torch.empty((B, T, F, N), dtype=x.dtype, device=x.device)
    _chk(_actd(out, "out").shape == (B, T, F, N), "conv3x3_fwd: out shape")
    a16 = _a16(x, out)
    sfx = _nt_suffix()
    _chk(not a16 or sfx == "_bf16", "bfloat16 activations need the bf16 mixed-precision mode")
    sfx += a16
    h2 = ()
    if sfx == "_h2":
        amax_w = pw.amax if pw.amax is not None else absmax(w32)
        h2 = ((absmax(x) if amax is None else amax).data_ptr(), amax_w.data_ptr())
    if (pw.frag is not None and pw.terms == _mode_terms() and CONV_WFRAG
            and (pw.terms != 1 or pw.half == HALF_DTYPE) and _lib.load().pe_conv3x3_wf_supported(F, Cc, N)):
        parts = None
        if bn_stats:
            parts = torch.empty((_lib.load().pe_conv3x3_wf_stat_parts(B, T, F), 2, N), dtype=torch.float64,
                                device=x.device)
        _call("pe_conv3x3_fwd_wf" + sfx, x.data_ptr(), pw.frag.data_ptr(), out.data_ptr(), B, T, F, Cc, N,
              int(bool(accumulate)), _lib.ptr(parts), *h2, _s(), work=2.0 * B * T * F * N * 9 * Cc)
        return (out, parts) if bn_stats is not None else out
    _call("pe_conv3x3_fwd" + sfx, x.data_ptr(), w32.data_ptr(), out.data_ptr(), B, T, F, Cc, N,
          int(bool(accumulate)), *h2, _s(), work=2.0 * B * T * F * N * 9 * Cc)
    return (out, None) if bn_stats is not None else out


def conv3x3_wgrad(x, dy, dw, amax_x=None, amax_dy=None):
    """dw (OIHW view, contiguous) = grad of conv3x3 wrt weights."""
    x = _actd(x, "x")
    dy = _actd(dy, "dy")
    dw = _dense(dw, "dw")
    B, T, F, Ci = x.shape
    Co = dy.shape[3]
    _chk(dy.shape[:3] == (B, T, F), "conv3x3_wgrad: dy shape")
    if _a16(x, dy):
        _chk(_tn_suffix() == "_bf16", "bfloat16 activations need the bf16 mixed-precision mode")
        ws = workspace(_lib.load().pe_conv3x3_wgrad_workspace_bytes(B, T, F, Ci, Co), x.device)
        _call("pe_conv3x3_wgrad_bf16_a16", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, T, F, Ci, Co, ws.data_ptr(),
              ws.numel(), _s(), work=2.0 * B * T * F * Co * 9 * Ci)
        return dw
    _chk(dw.shape == (Co, Ci, 3, 3), "conv3x3_wgrad: dw shape")
    lib = _lib.load()
    ws = workspace(lib.pe_conv3x3_wgrad_workspace_bytes(B, T, F, Ci, Co), x.device)
    h2 = ()
    if _tn_suffix() == "_h2":
        h2 = ((absmax(x) if amax_x is None else amax_x).data_ptr(), (absmax(dy) if amax_dy is None else amax_dy).data_ptr())
    _call("pe_conv3x3_wgrad" + _tn_suffix(), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, T, F, Ci, Co,
          ws.data_ptr(), ws.numel(), *h2, _s(), work=2.0 * B * T * F * Co * 9 * Ci)
    return dw


def _btf_view(x):
    """(B,T,F) float32 view with arbitrary strides (the mel batch after transpose(-1,-2))."""
    _f32c(x, "x")
    _chk(x.dim() == 3, "expected (B,T,F)")
    return x.shape, x.stride()


def conv3x3_c1_fwd(x_btf, w, out=None, bn_stats=None):
    """First convolution.  bn_stats is None: returns out.  Otherwise returns (out, partials): with bn_stats true the
    kernel also leaves the BatchNorm batch statistics of its output as [parts][2][64] float64 partial sums (else None),
    as conv3x3_fwd does."""
    (B, T, F), (sb, st, sf) = _btf_view(x_btf)
    w = _dense(w, "w")
    _chk(w.shape == (64, 1, 3, 3), "first conv is 1 -> 64")
    if out is None:
        out = torch.empty((B, T, F, 64), dtype=act_dtype(), device=x_btf.device)
    _chk(_actd(out, "out").shape == (B, T, F, 64), "conv3x3_c1_fwd: out shape")
    parts = None
    if bn_stats:
        parts = torch.empty((_lib.load().pe_conv3x3_c1_stat_parts(B, T, F), 2, 64), dtype=torch.float64,
                            device=x_btf.device)
    _call("pe_conv3x3_c1_fwd" + _a16(out), x_btf.data_ptr(), sb, st, sf, w.data_ptr(), out.data_ptr(), B, T, F, _lib.ptr(parts),
          _s())
    return out if bn_stats is None else (out, parts)


def conv3x3_c1_wgrad(x_btf, dy, dw):
    (B, T, F), (sb, st, sf) = _btf_view(x_btf)
    dy = _actd(dy, "dy")
    dw = _dense(dw, "dw")
    _chk(dy.shape == (B, T, F, 64) and dw.shape == (64, 1, 3, 3), "conv3x3_c1_wgrad: shapes")
    lib = _lib.load()
    ws = workspace(lib.pe_conv3x3_wgrad_workspace_bytes(B, T, F, 1, 64), dy.device)
    _call("pe_conv3x3_c1_wgrad" + _a16(dy), x_btf.data_ptr(), sb, st, sf, dy.data_ptr(), dw.data_ptr(), B, T, F,
          ws.data_ptr(), ws.numel(), _s())
    return dw


# ------------------------------------------------------------------ BN / pooling / dropout
class BnState:
    """Per-call BN tensors: mean, invstd, scale, shift (all [C])."""

    def __init__(self, C_, device):
        buf = torch.empty((4, C_), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_train_stats(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, partials=None):
    """Train-mode BatchNorm statistics of x [.., C].  ``partials`` ([parts, 2, C] float64 column sums / sums of
    squares left behind by the kernel that produced x) skips the pass over x."""
    x = _actd(x, "x")
    Cc = x.shape[-1]
    for t, n in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        if t is not None:
            _chk(_dense(t, n).numel() == Cc, f"{n}: size")
    st = BnState(Cc, x.device)
    lib = _lib.load()
    if partials is not None:
        _chk(partials.is_cuda and partials.dtype == torch.float64 and partials.is_contiguous() and partials.dim() == 3
             and partials.shape[1:] == (2, Cc), "bn_train_stats: partials [parts, 2, C] float64")
        ws = workspace(lib.pe_bn_workspace_bytes(Cc), x.device)
        _call("pe_bn_finalize_stats", partials.data_ptr(), partials.shape[0], x.numel() // Cc, Cc, gamma.data_ptr(),
              beta.data_ptr(), eps, momentum, _lib.ptr(running_mean), _lib.ptr(running_var), st.mean.data_ptr(),
              st.invstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), ws.data_ptr(), ws.numel(), _s())
        return st
    ws = workspace(lib.pe_bn_workspace_bytes(Cc), x.device)
    _call("pe_bn_train_stats" + _a16(x), x.data_ptr(), x.numel() // Cc, Cc, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
          _lib.ptr(running_mean), _lib.ptr(running_var), st.mean.data_ptr(), st.invstd.data_ptr(),
          st.scale.data_ptr(), st.shift.data_ptr(), ws.data_ptr(), ws.numel(), _s())
    return st


def bn_eval_affine(gamma, beta, running_mean, running_var, eps=1e-5):
    Cc = gamma.numel()
    st = BnState(Cc, gamma.device)
    for t in (gamma, beta, running_mean, running_var):
        _chk(_dense(t, "bn tensor").numel() == Cc, "bn tensor size")
    _call("pe_bn_eval_affine", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
          eps, Cc, st.scale.data_ptr(), st.shift.data_ptr(), _s())
    return st


def _slice_target(out, B, T, Fo, Cc, coff):
    """out is a dense [B,T,Fo,Ctot] tensor; we write channels [coff, coff+C)."""
    _actd(out, "out")
    _chk(out.dim() == 4 and out.shape[:3] == (B, T, Fo), "output pixel grid mismatch")
    ld = out.shape[3]
    _chk(0 <= coff and coff + Cc <= ld, "channel slice out of range")
    return ld


def bn_act_pool_fwd(x, st: BnState, pool=1, slope=0.01, out=None, coff=0, amax_out=None):
    """``amax_out``: a zeroed word from ``amax_word()``; the pass leaves the absmax of its output there."""
    x = _actd(x, "x")
    B, T, F, Cc = x.shape
    Fo = F // pool
    if out is None:
        out = torch.empty((B, T, Fo, Cc), dtype=x.dtype, device=x.device)
    ld = _slice_target(out, B, T, Fo, Cc, coff)
    if _a16(x, out):
        _call("pe_bn_act_pool_fwd_a16", x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), slope, out.data_ptr(),
              B * T, F, Cc, pool, ld, coff, _s())
        return out
    _call("pe_bn_act_pool_fwd", x.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), slope, out.data_ptr(),
          B * T, F, Cc, pool, ld, coff, _lib.ptr(amax_out), _s())
    return out


def bn_act_pool_bwd(x, dy, st: BnState, dgamma, dbeta, pool=1, slope=0.01, coff=0, dx=None, amax_out=None):
    x = _actd(x, "x")
    B, T, F, Cc = x.shape
    ld = _slice_target(dy, B, T, F // pool, Cc, coff)
    if dx is None:
        dx = torch.empty_like(x)
    _chk(_actd(dx, "dx").shape == x.shape, "dx shape")
    _chk(_dense(dgamma, "dgamma").numel() == Cc and _dense(dbeta, "dbeta").numel() == Cc, "dgamma/dbeta size")
    lib = _lib.load()
    ws = workspace(lib.pe_bn_workspace_bytes(Cc) + 8 * Cc, x.device)
    if _a16(x, dy, dx):
        _call("pe_bn_act_pool_bwd_a16", x.data_ptr(), dy.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
              st.mean.data_ptr(), st.invstd.data_ptr(), slope, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
              B * T, F, Cc, pool, ld, coff, ws.data_ptr(), ws.numel(), _s())
        return dx
    _call("pe_bn_act_pool_bwd", x.data_ptr(), dy.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
          st.mean.data_ptr(), st.invstd.data_ptr(), slope, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
          B * T, F, Cc, pool, ld, coff, ws.data_ptr(), ws.numel(), _lib.ptr(amax_out), _s())
    return dx


def maxpool_fwd(x, pool, out=None, coff=0, want_argmax=False):
    """MaxPool2d((1, pool)) into a channel slice.  ``want_argmax``: also return the uint8 window positions of the
    maxima ([B, T, F // pool, C]) for ``maxpool_bwd_add(argmax=...)``: returns (out, argmax)."""
    x = _actd(x, "x")
    B, T, F, Cc = x.shape
    Fo = F // pool
    if out is None:
        out = torch.empty((B, T, Fo, Cc), dtype=x.dtype, device=x.device)
    ld = _slice_target(out, B, T, Fo, Cc, coff)
    arg = torch.empty((B, T, Fo, Cc), dtype=torch.uint8, device=x.device) if want_argmax else None
    _call("pe_maxpool_fwd" + _a16(x, out), x.data_ptr(), out.data_ptr(), B * T, F, Cc, pool, ld, coff, _lib.ptr(arg), _s())
    return (out, arg) if want_argmax else out


def maxpool_bwd_add(x, dy, dx, pool, coff=0, amax_out=None, argmax=None):
    """dx[first maximum of each window] += dy.  With ``argmax`` (from ``maxpool_fwd``) x is only used for its shape."""
    B, T, F, Cc = x.shape
    if argmax is None:
        x = _actd(x, "x")
    else:
        _chk(argmax.is_cuda and argmax.dtype == torch.uint8 and argmax.is_contiguous()
             and argmax.shape == (B, T, F // pool, Cc), "maxpool_bwd_add: argmax shape")
    ld = _slice_target(dy, B, T, F // pool, Cc, coff)
    _chk(_actd(dx, "dx").shape == x.shape, "dx shape")
    if _a16(dy, dx, x if argmax is None else None):
        _call("pe_maxpool_bwd_add_a16", x.data_ptr() if argmax is None else 0, _lib.ptr(argmax), dy.data_ptr(),
              dx.data_ptr(), B * T, F, Cc, pool, ld, coff, _s())
        return dx
    _call("pe_maxpool_bwd_add", x.data_ptr() if argmax is None else 0, _lib.ptr(argmax), dy.data_ptr(), dx.data_ptr(),
          B * T, F, Cc, pool, ld, coff, _lib.ptr(amax_out), _s())
    return dx


def dropout(x2d, p, out2d=None, mask_in=None, want_mask=True, seed=0, offset=0):
    """Row-strided 2-D dropout.  Returns (out, mask uint8 [rows, cols] or None)."""
    rows, cols, ldx = _rows2d(x2d, "x", act=True)
    if out2d is None:
        out2d = torch.empty((rows, cols), dtype=x2d.dtype, device=x2d.device)
    r2, c2, ldy = _rows2d(out2d, "out", act=True)
    _chk((r2, c2) == (rows, cols), "dropout: out shape")
    mask_out = None
    if mask_in is not None:
        _chk(mask_in.is_cuda and mask_in.dtype == torch.uint8 and mask_in.is_contiguous()
             and mask_in.numel() == rows * cols, "dropout: mask_in")
    elif want_mask:
        mask_out = torch.empty((rows, cols), dtype=torch.uint8, device=x2d.device)
    _call("pe_dropout_fwd" + _a16(x2d, out2d), x2d.data_ptr(), ldx, out2d.data_ptr(), ldy, _lib.ptr(mask_in), _lib.ptr(mask_out),
          rows, cols, float(p), int(seed), int(offset), _s())
    return out2d, (mask_in if mask_in is not None else mask_out)


def nhwc_to_seq(x, Cc, coff=0, out=None):
    """x dense [B,T,2,Ctot] (channels [coff,coff+C)) -> seq [B,T,2C] with feature c*2+w."""
    x = _actd(x, "x")
    B, T, two, ld = x.shape
    _chk(two == 2 and coff + Cc <= ld, "nhwc_to_seq: shape")
    if out is None:
        out = torch.empty((B, T, 2 * Cc), dtype=torch.float32, device=x.device)
    _chk(_dense(out, "out").shape == (B, T, 2 * Cc), "nhwc_to_seq: out shape")       # the temporal heads read fp32
    _call("pe_nhwc_to_seq" + _a16(x), x.data_ptr(), ld, coff, out.data_ptr(), B * T, Cc, _s())
    return out


def seq_to_nhwc(seq, out, Cc, coff=0, accumulate=False):
    seq = _dense(seq, "seq")
    out = _actd(out, "out")
    B, T, two, ld = out.shape
    _chk(two == 2 and coff + Cc <= ld and seq.shape == (B, T, 2 * Cc), "seq_to_nhwc: shape")
    _call("pe_seq_to_nhwc" + _a16(out), seq.data_ptr(), out.data_ptr(), ld, coff, B * T, Cc, int(bool(accumulate)), _s())
    return out


def copy2d(src2d, dst2d, accumulate=False):
    r, c, lds = _rows2d(src2d, "src", act=True)
    r2, c2, ldd = _rows2d(dst2d, "dst", act=True)
    _chk((r, c) == (r2, c2), "copy2d: shape")
    _call("pe_copy2d" + _a16(src2d, dst2d), src2d.data_ptr(), lds, dst2d.data_ptr(), ldd, r, c, int(bool(accumulate)), _s())
    return dst2d


# ------------------------------------------------------------------ LSTM
def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _int_array(vals):
    arr = (C.c_int * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


_SYNC: dict = {}
USE_PERSISTENT_LSTM = True


def _lstm_sync(ncells, B, device):
    """Zero-initialised barrier words for the persistent LSTM kernels (word 0 = sticky error flag)."""
    lib = _lib.load()
    need = lib.pe_lstm_persistent_sync_bytes(4, max(B, 256)) // 4
    key = torch.device(device)
    buf = _SYNC.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros(need, dtype=torch.int32, device=key)
        _SYNC[key] = buf
    return buf


def persistent_lstm_error(device) -> bool:
    """True if a persistent-LSTM group barrier ever timed out on this device (results invalid)."""
    buf = _SYNC.get(torch.device(device))
    return bool(buf is not None and int(buf[0].item()) != 0)


def persistent_lstm_error_word(device):
    """The sticky barrier-timeout word as a 1-element device tensor (None before any persistent launch): a
    data-parallel trainer max-reduces it across ranks and reads it once (distributed.GradientAllReduce.any_rank_word)."""
    buf = _SYNC.get(torch.device(device))
    return None if buf is None else buf[0:1]


def clear_persistent_lstm_error(device) -> None:
    """Reset the sticky barrier-timeout word (the persistent kernels skip their waits while it is set)."""
    buf = _SYNC.get(torch.device(device))
    if buf is not None:
        buf[0:1].zero_()


def _persistent_ok(ncells, B, H, device, which="fwd"):
    """The persistent recurrence kernels serve this configuration: enabled, a 16-bit-term product form selected (x3 /
    bf16 / f16: there is no native-fp32 persistent kernel) and the shape / device admitted by the library."""
    if not USE_PERSISTENT_LSTM or _lstm_suffix(which) == "":
        return False
    with torch.cuda.device(device):
        return bool(_lib.load().pe_lstm_persistent_supported(ncells, B, H))


def lstm_fwd(whh, gates, y_slices, cbuf, reverse, B, T, H):
    """Advance len(whh) cells through all T steps.  y_slices[i] is a view [B,T,H] into a dense
    [B,T,ldy] output (ldy = 2H for bidirectional)."""
    n = len(whh)
    _chk(1 <= n <= 4 and len(gates) == len(y_slices) == len(cbuf) == len(reverse) == n, "lstm_fwd: cell lists")
    ldy = None
    for i in range(n):
        _chk(_dense(whh[i], "whh").shape == (4 * H, H), "whh shape")
        _chk(_dense(gates[i], "gates").shape == (B, T, 4 * H), "gates shape")
        _chk(_dense(cbuf[i], "cbuf").shape == (B, T, H), "cbuf shape")
        ys = _f32c(y_slices[i], "y")
        _chk(ys.shape == (B, T, H) and ys.stride(2) == 1 and ys.stride(0) == T * ys.stride(1), "y slice layout")
        _chk(ldy in (None, ys.stride(1)), "all y slices share ldy")
        ldy = ys.stride(1)
    dev = gates[0].device
    if _persistent_ok(n, B, H, dev, "fwd"):
        sync = _lstm_sync(n, B, dev)
        _call("pe_lstm_fwd_persistent" + _lstm_suffix("fwd"), n, _ptr_array(whh), _ptr_array(gates), _ptr_array(y_slices),
              _ptr_array(cbuf), _int_array(reverse), ldy, B, T, H, sync.data_ptr(), _s(),
              work=2.0 * n * B * (T - 1) * 4 * H * H)
        return
    _call("pe_lstm_fwd", n, _ptr_array(whh), _ptr_array(gates), _ptr_array(y_slices), _ptr_array(cbuf),
          _int_array(reverse), ldy, B, T, H, _s(), work=2.0 * n * B * (T - 1) * 4 * H * H)


def lstm_bwd(whh_t, gates, cbuf, dy_slices, dcarry, reverse, B, T, H, dbias_rows=None, amax_out=None):
    """Backward recurrences of len(whh_t) cells; gates become d(pre-activation gates) in place.  dbias_rows: optional
    list of [lstm_bwd_dbias_rows(...)][4H] buffers, one per cell, that receive the per-batch-tile column sums of the
    gate gradients (only pass it when that query is non-zero); returns True if they were written.  amax_out: optional
    list of zeroed words (``amax_word()``), one per cell, filled with the gate gradients' absmax under the same
    condition (else untouched: take ``absmax`` of the tensors)."""
    n = len(whh_t)
    _chk(1 <= n <= 4 and len(gates) == len(dy_slices) == len(cbuf) == len(reverse) == len(dcarry) == n,
         "lstm_bwd: cell lists")
    ld = None
    for i in range(n):
        _chk(_dense(whh_t[i], "whh_t").shape == (H, 4 * H), "whh_t shape")
        _chk(_dense(gates[i], "gates").shape == (B, T, 4 * H), "gates shape")
        _chk(_dense(cbuf[i], "cbuf").shape == (B, T, H), "cbuf shape")
        _chk(_dense(dcarry[i], "dcarry").shape == (B, H), "dcarry shape")
        d = _f32c(dy_slices[i], "dy")
        _chk(d.shape == (B, T, H) and d.stride(2) == 1 and d.stride(0) == T * d.stride(1), "dy slice layout")
        _chk(ld in (None, d.stride(1)), "all dy slices share ld")
        ld = d.stride(1)
    dev = gates[0].device
    if _persistent_ok(n, B, H, dev, "bwd"):
        sync = _lstm_sync(n, B, dev)
        rows = lstm_bwd_dbias_rows(n, B, T, H, ld, dev) if dbias_rows is not None else 0
        if rows:
            _chk(len(dbias_rows) == n and all(_dense(d, "dbias_rows").shape == (rows, 4 * H) for d in dbias_rows),
                 "lstm_bwd: dbias_rows shape")
        _call("pe_lstm_bwd_persistent" + _lstm_suffix("bwd"), n, _ptr_array(whh_t), _ptr_array(gates), _ptr_array(cbuf),
              _ptr_array(dy_slices), _int_array(reverse), ld, B, T, H, _ptr_array(dbias_rows) if rows else None,
              _ptr_array(amax_out) if (rows and amax_out is not None) else None,
              sync.data_ptr(), _s(), work=2.0 * n * B * (T - 1) * 4 * H * H)
        return bool(rows)
    _call("pe_lstm_bwd", n, _ptr_array(whh_t), _ptr_array(gates), _ptr_array(cbuf), _ptr_array(dy_slices),
          _ptr_array(dcarry), _int_array(reverse), ld, B, T, H, _s(), work=2.0 * n * B * (T - 1) * 4 * H * H)
    return False


def lstm_bwd_dbias_rows(n, B, T, H, ld, device) -> int:
    """Rows of the per-cell [rows][4H] bias-gradient partials lstm_bwd can emit for this configuration (0: it cannot;
    sum the gate gradients with colsum instead)."""
    if not _persistent_ok(n, B, H, device, "bwd"):
        return 0
    return int(_lib.load().pe_lstm_bwd_persistent_dbias_rows(n, B, T, H, ld))


def lstm_whh_grad(dgates, y_slice, dwhh, reverse, B, T, H, amax_dg=None, amax_y=None):
    _chk(_dense(dgates, "dgates").shape == (B, T, 4 * H), "dgates shape")
    ys = _f32c(y_slice, "y")
    _chk(ys.shape == (B, T, H) and ys.stride(2) == 1 and ys.stride(0) == T * ys.stride(1), "y slice layout")
    _chk(_dense(dwhh, "dwhh").shape == (4 * H, H), "dwhh shape")
    lib = _lib.load()
    ws = workspace(lib.pe_lstm_whh_grad_workspace_bytes(B, T, H), dgates.device)
    h2 = ()
    if _tn_suffix() == "_h2":
        if amax_y is None:                 # |h| < 1 by construction (o * tanh(c)): a constant bound is a valid amax
            amax_y = _unit_amax(dgates.device)
        h2 = ((absmax(dgates) if amax_dg is None else amax_dg).data_ptr(), amax_y.data_ptr())
    _call("pe_lstm_whh_grad" + _tn_suffix(), dgates.data_ptr(), ys.data_ptr(), ys.stride(1), dwhh.data_ptr(), B, T, H,
          int(bool(reverse)), ws.data_ptr(), ws.numel(), *h2, _s(), work=2.0 * B * T * 4 * H * H)
    return dwhh


def colsum(x2d, out0, out1=None):
    rows, cols, ld = _rows2d(x2d, "x")
    _chk(_dense(out0, "out0").numel() == cols, "colsum: out size")
    if out1 is not None:
        _chk(_dense(out1, "out1").numel() == cols, "colsum: out1 size")
    lib = _lib.load()
    ws = workspace(lib.pe_colsum_workspace_bytes(cols), x2d.device)
    _call("pe_colsum", x2d.data_ptr(), rows, cols, ld, out0.data_ptr(), _lib.ptr(out1), ws.data_ptr(), ws.numel(),
          _s())
    return out0


# ------------------------------------------------------------------ heads / loss / optimiser
def head_fwd(x2d, w, bias, out=None):
    R, D, ldx = _rows2d(x2d, "x")
    w = _dense(w, "w")
    bias = _dense(bias, "bias")
    n_out = w.shape[0]
    _chk(w.shape == (n_out, D) and bias.numel() == n_out, "head_fwd: weight shape")
    if out is None:
        out = torch.empty((R,), dtype=torch.float32, device=x2d.device)
    _chk(_dense(out, "out").numel() == R, "head_fwd: out size")
    _call("pe_head_fwd", x2d.data_ptr(), ldx, w.data_ptr(), bias.data_ptr(), n_out, out.data_ptr(), R, D, _s())
    return out


def head_bwd(x2d, w, dy, dw, db, dx=None):
    R, D, ldx = _rows2d(x2d, "x")
    n_out = w.shape[0]
    _chk(_dense(dy, "dy").numel() == R, "head_bwd: dy size")
    _chk(_dense(dw, "dw").shape == (n_out, D) and _dense(db, "db").numel() == n_out, "head_bwd: grads shape")
    if dx is None:
        dx = torch.empty((R, D), dtype=torch.float32, device=x2d.device)
    R2, D2, lddx = _rows2d(dx, "dx")
    _chk((R2, D2) == (R, D), "head_bwd: dx shape")
    lib = _lib.load()
    ws = workspace(lib.pe_head_bwd_workspace_bytes(D), x2d.device)
    _call("pe_head_bwd", x2d.data_ptr(), ldx, w.data_ptr(), dy.data_ptr(), n_out, dx.data_ptr(), lddx,
          dw.data_ptr(), db.data_ptr(), R, D, ws.data_ptr(), ws.numel(), _s())
    return dx


def f0_sil_loss(f0_pred, f0, sil_pred, sil, lambda_f0, grad_scale=1.0, want_grads=True):
    R = f0.numel()
    for t, n in ((f0_pred, "f0_pred"), (f0, "f0"), (sil_pred, "sil_pred"), (sil, "sil")):
        _chk(_dense(t, n).numel() == R, f"{n}: size")
    out3 = torch.empty((3,), dtype=torch.float32, device=f0.device)
    d_f0 = torch.empty((R,), dtype=torch.float32, device=f0.device) if want_grads else None
    d_sil = torch.empty((R,), dtype=torch.float32, device=f0.device) if want_grads else None
    _call("pe_f0_sil_loss", f0_pred.data_ptr(), f0.data_ptr(), sil_pred.data_ptr(), sil.data_ptr(),
          float(lambda_f0), R, float(grad_scale), out3.data_ptr(), _lib.ptr(d_f0), _lib.ptr(d_sil), _s())
    return out3, d_f0, d_sil


def f0_bins_ce_loss(logits, f0, sil_pred, sil, lambda_f0, grad_scale=1.0, want_grads=True):
    """360-bin (CREPE-style) F0 classification loss + silence BCE (build-defined).
    logits (R, C), f0 / sil_pred / sil (R,).  Returns (out4 = [total, lambda*CE, BCE, n_voiced], d_logits, d_sil)."""
    logits = _dense(logits, "logits")
    _chk(logits.dim() == 2, "logits: expected (R, C)")
    R, C = logits.shape
    for t, n in ((f0, "f0"), (sil_pred, "sil_pred"), (sil, "sil")):
        _chk(_dense(t, n).numel() == R, f"{n}: size")
    lib = _lib.load()
    out4 = torch.empty((4,), dtype=torch.float32, device=f0.device)
    d_logits = torch.empty((R, C), dtype=torch.float32, device=f0.device) if want_grads else None
    d_sil = torch.empty((R,), dtype=torch.float32, device=f0.device) if want_grads else None
    nbytes = lib.pe_f0_bins_ce_workspace_bytes(R)
    ws = workspace(nbytes, f0.device)
    _call("pe_f0_bins_ce_loss", logits.data_ptr(), C, C, f0.data_ptr(), sil_pred.data_ptr(), sil.data_ptr(),
          float(lambda_f0), R, float(grad_scale), out4.data_ptr(), _lib.ptr(d_logits), C, _lib.ptr(d_sil),
          ws.data_ptr(), ws.numel(), _s())
    return out4, d_logits, d_sil


def nonfinite_flag(x, flag=None):
    """int32 device scalar: 1 if any element of the flat float32 tensor ``x`` is inf / nan (GradScaler's test)."""
    x = _dense(x, "x")
    if flag is None:
        flag = torch.empty((1,), dtype=torch.int32, device=x.device)
    _chk(flag.is_cuda and flag.dtype == torch.int32 and flag.numel() == 1, "flag: int32 device scalar")
    _call("pe_nonfinite_flag", x.data_ptr(), x.numel(), flag.data_ptr(), _s())
    return flag


def adamw_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0,
               skip_flag=None):
    """``skip_flag``: optional 1-element float32 device tensor; non-zero at execution time = update nothing."""
    n = param.numel()
    for t, nme in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(_dense(t, nme).numel() == n, f"{nme}: size")
    bc1 = 1.0 - float(beta1) ** int(step)
    bc2 = 1.0 - float(beta2) ** int(step)
    _call("pe_adamw_step", param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
          float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), bc1, bc2, float(grad_scale),
          _lib.ptr(skip_flag), _s())


# ------------------------------------------------------------------ transformer pieces
def bgemm(mode, A, a_view, B, b_view, C, c_view, inner, batch, M, N, K, alpha=1.0, accumulate=False):
    """Batched 64x64-tiled GEMM.  ``*_view`` = (ld, outer_stride, inner_stride) in elements; matrix b
    of an operand starts at element (b // inner) * outer + (b % inner) * inner_stride of its tensor
    (tensors may be views: the data pointer carries any extra offset)."""
    for t, n in ((A, "A"), (B, "B"), (C, "C")):
        _f32c(t, n)

    def _extent(view, rows, cols):
        ld, so, si = view
        return ((batch - 1) // inner) * so + ((batch - 1) % inner if batch > 0 else 0) * si + (rows - 1) * ld + cols

    a_rows, a_cols = (M, K) if mode != 2 else (K, M)
    b_rows, b_cols = (N, K) if mode == 0 else (K, N)
    for t, view, r, c, n in ((A, a_view, a_rows, a_cols, "A"), (B, b_view, b_rows, b_cols, "B"),
                             (C, c_view, M, N, "C")):
        avail = t.untyped_storage().nbytes() // 4 - t.storage_offset()
        _chk(_extent(view, r, c) <= avail, f"bgemm: operand {n} exceeds its storage")
    _call("pe_bgemm", int(mode), A.data_ptr(), *map(int, a_view), B.data_ptr(), *map(int, b_view), C.data_ptr(),
          *map(int, c_view), int(inner), int(batch), int(M), int(N), int(K), float(alpha), int(bool(accumulate)),
          _s(), work=2.0 * batch * M * N * K)
    return C


FUSED_ATTENTION = os.environ.get("PE_FUSED_ATTENTION", "1") == "1"


def attn_supported(T, dh):
    return bool(FUSED_ATTENTION and _lib.load().pe_attn_supported(int(T), int(dh)))


def _attn_suffix():
    """Mixed precision with bf16 operands runs the attention matmuls on the bf16 MFMA as autocast does (softmax, the
    log-sum-exp and every tensor in memory stay fp32); the fp16 mode and the fp32 modes keep the exact-fp32 MFMAs."""
    return "_bf16" if (MATMUL_BF16 and HALF_DTYPE == "bf16") else ""


def attn_fwd(qkv, B, T, H, scale, p=0.0, mask_in=None, seed=0, offset=0):
    """Fused softmax(Q K^T * scale) -> dropout(p) -> . V for packed projections qkv [B*T, 3*H*dh].
    Returns (o [B*T, H*dh], lse [B*H*T], keep mask uint8 [B*H*T, T] or None)."""
    qkv = _dense(qkv, "qkv")
    R, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    _chk(R == B * T and D3 == 3 * H * dh, "attn_fwd: qkv shape")
    o = torch.empty((R, D), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B * H * T,), dtype=torch.float32, device=qkv.device)
    mask_out = None
    if p > 0.0:
        if mask_in is not None:
            _chk(mask_in.is_cuda and mask_in.dtype == torch.uint8 and mask_in.is_contiguous()
                 and mask_in.numel() == B * H * T * T, "attn_fwd: mask_in")
        else:
            mask_out = torch.empty((B * H * T, T), dtype=torch.uint8, device=qkv.device)
    _call("pe_attn_fwd" + _attn_suffix(), qkv.data_ptr(), D3, o.data_ptr(), D, lse.data_ptr(), _lib.ptr(mask_in if p > 0.0 else None),
          _lib.ptr(mask_out), B, T, H, dh, float(scale), float(p), int(seed), int(offset), _s(),
          work=4.0 * B * H * T * T * dh)
    return o, lse, (mask_in if (p > 0.0 and mask_in is not None) else mask_out)


def attn_bwd(qkv, o, d_o, lse, mask, B, T, H, scale, p=0.0):
    """Gradient of attn_fwd wrt the packed projections: dqkv [B*T, 3*H*dh]."""
    qkv, o, d_o, lse = _dense(qkv, "qkv"), _dense(o, "o"), _dense(d_o, "d_o"), _dense(lse, "lse")
    R, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    _chk(o.shape == (R, D) and d_o.shape == (R, D) and lse.numel() == B * H * T, "attn_bwd: shapes")
    if p > 0.0:
        _chk(mask is not None and mask.is_cuda and mask.dtype == torch.uint8 and mask.numel() == B * H * T * T,
             "attn_bwd: mask")
    dqkv = torch.empty_like(qkv)
    _call("pe_attn_bwd" + _attn_suffix(), qkv.data_ptr(), D3, o.data_ptr(), d_o.data_ptr(), D, lse.data_ptr(),
          _lib.ptr(mask if p > 0.0 else None), dqkv.data_ptr(), B, T, H, dh, float(scale), float(p), _s(),
          work=10.0 * B * H * T * T * dh)
    return dqkv


def softmax_fwd_(s2d, scale):
    rows, L, ld = _rows2d(s2d, "scores")
    _chk(ld == L, "softmax: dense rows")
    _call("pe_softmax_fwd", s2d.data_ptr(), rows, L, float(scale), _s())
    return s2d


def softmax_bwd_(p2d, dp2d, scale):
    rows, L, ld = _rows2d(p2d, "p")
    r2, L2, ld2 = _rows2d(dp2d, "dp")
    _chk((rows, L, ld) == (r2, L2, ld2) and ld == L, "softmax_bwd: shapes")
    _call("pe_softmax_bwd", p2d.data_ptr(), dp2d.data_ptr(), rows, L, float(scale), _s())
    return dp2d


class LnState:
    def __init__(self, z, mean, rstd):
        self.z, self.mean, self.rstd = z, mean, rstd


def layernorm_fwd(a2d, gamma, beta, b2d=None, pe=None, eps=1e-5, keep_z=True):
    """y = LN(a + b + pe[row % period]); returns (y, LnState).  ``pe`` is [period, D] contiguous."""
    a2d = _dense(a2d, "a")
    R, D = a2d.shape
    if b2d is not None:
        _chk(_dense(b2d, "b").shape == (R, D), "layernorm: b shape")
    period = 0
    if pe is not None:
        _chk(_dense(pe, "pe").dim() == 2 and pe.shape[1] == D, "layernorm: pe shape")
        period = pe.shape[0]
    _chk(_dense(gamma, "gamma").numel() == D and _dense(beta, "beta").numel() == D, "layernorm: affine size")
    need_z = keep_z and (b2d is not None or pe is not None)
    z = torch.empty_like(a2d) if need_z else None
    y = torch.empty_like(a2d)
    mean = torch.empty((R,), dtype=torch.float32, device=a2d.device)
    rstd = torch.empty((R,), dtype=torch.float32, device=a2d.device)
    _call("pe_layernorm_fwd", a2d.data_ptr(), _lib.ptr(b2d), _lib.ptr(pe), period, gamma.data_ptr(), beta.data_ptr(),
          float(eps), _lib.ptr(z), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, D, _s())
    return y, LnState(z if need_z else (a2d if keep_z else None), mean, rstd)


def layernorm_bwd(dy2d, st: LnState, gamma, dgamma, dbeta, dz=None, dy_add=None, drop_mask=None, p=0.0):
    """dz = dLN(dy [+ dy_add]).  With ``drop_mask`` (uint8 [R, D], rate ``p``) returns (dz, dropout_bwd(dz)) from the
    same pass."""
    dy2d = _dense(dy2d, "dy")
    R, D = dy2d.shape
    _chk(_dense(st.z, "z").shape == (R, D), "layernorm_bwd: z shape")
    if dz is None:
        dz = torch.empty_like(dy2d)
    _chk(_dense(dz, "dz").shape == (R, D), "layernorm_bwd: dz shape")
    _chk(_dense(dgamma, "dgamma").numel() == D and _dense(dbeta, "dbeta").numel() == D, "layernorm_bwd: grads")
    lib = _lib.load()
    ws = workspace(lib.pe_layernorm_bwd_workspace_bytes(D), dy2d.device)
    if dy_add is None and drop_mask is None:
        _call("pe_layernorm_bwd", dy2d.data_ptr(), st.z.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(),
              gamma.data_ptr(), dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), R, D, ws.data_ptr(), ws.numel(),
              _s())
        return dz
    if dy_add is not None:
        _chk(_dense(dy_add, "dy_add").shape == (R, D), "layernorm_bwd: dy_add shape")
    dz_drop = None
    if drop_mask is not None:
        _chk(drop_mask.is_cuda and drop_mask.dtype == torch.uint8 and drop_mask.is_contiguous()
             and drop_mask.numel() == R * D, "layernorm_bwd: drop_mask")
        dz_drop = torch.empty_like(dy2d)
    _call("pe_layernorm_bwd_fused", dy2d.data_ptr(), _lib.ptr(dy_add), st.z.data_ptr(), st.mean.data_ptr(),
          st.rstd.data_ptr(), gamma.data_ptr(), dz.data_ptr(), _lib.ptr(drop_mask), float(p), _lib.ptr(dz_drop),
          dgamma.data_ptr(), dbeta.data_ptr(), R, D, ws.data_ptr(), ws.numel(), _s())
    return dz if drop_mask is None else (dz, dz_drop)


def layernorm_dropout_fwd(a2d, b2d, gamma, beta, p, mask_in=None, seed=0, offset=0, eps=1e-5):
    """y = LN(a + dropout(b)); returns (y, LnState, mask uint8 [R, D]).  Same masks and values as ``dropout`` followed
    by ``layernorm_fwd(a, b2d=...)``."""
    a2d, b2d = _dense(a2d, "a"), _dense(b2d, "b")
    R, D = a2d.shape
    _chk(b2d.shape == (R, D), "layernorm_dropout: b shape")
    _chk(_dense(gamma, "gamma").numel() == D and _dense(beta, "beta").numel() == D, "layernorm_dropout: affine size")
    mask_out = None
    if mask_in is not None:
        _chk(mask_in.is_cuda and mask_in.dtype == torch.uint8 and mask_in.is_contiguous()
             and mask_in.numel() == R * D, "layernorm_dropout: mask_in")
    else:
        mask_out = torch.empty((R, D), dtype=torch.uint8, device=a2d.device)
    z = torch.empty_like(a2d)
    y = torch.empty_like(a2d)
    mean = torch.empty((R,), dtype=torch.float32, device=a2d.device)
    rstd = torch.empty((R,), dtype=torch.float32, device=a2d.device)
    _call("pe_layernorm_dropout_fwd", a2d.data_ptr(), b2d.data_ptr(), None, 0, gamma.data_ptr(), beta.data_ptr(),
          float(eps), z.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, D, _lib.ptr(mask_in),
          _lib.ptr(mask_out), float(p), int(seed), int(offset), _s())
    return y, LnState(z, mean, rstd), (mask_in if mask_in is not None else mask_out)


def gelu_dropout_fwd(x, p, mask_in=None, seed=0, offset=0):
    """dropout(gelu(x)) in one pass; returns (out, mask uint8 of x's shape)."""
    x = _dense(x, "x")
    out = torch.empty_like(x)
    mask_out = None
    if mask_in is not None:
        _chk(mask_in.is_cuda and mask_in.dtype == torch.uint8 and mask_in.is_contiguous()
             and mask_in.numel() == x.numel(), "gelu_dropout: mask_in")
    else:
        mask_out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _call("pe_gelu_dropout_fwd", x.data_ptr(), out.data_ptr(), x.numel(), _lib.ptr(mask_in), _lib.ptr(mask_out),
          float(p), int(seed), int(offset), _s())
    return out, (mask_in if mask_in is not None else mask_out)


def gelu_dropout_bwd(x, dy, mask, p, out=None):
    x, dy = _dense(x, "x"), _dense(dy, "dy")
    _chk(x.numel() == dy.numel() == mask.numel() and mask.dtype == torch.uint8 and mask.is_contiguous(),
         "gelu_dropout_bwd: sizes")
    if out is None:
        out = torch.empty_like(x)
    _call("pe_gelu_dropout_bwd", x.data_ptr(), dy.data_ptr(), mask.data_ptr(), float(p), out.data_ptr(), x.numel(),
          _s())
    return out


def gelu_fwd(x, out=None):
    x = _dense(x, "x")
    if out is None:
        out = torch.empty_like(x)
    _chk(_dense(out, "out").numel() == x.numel(), "gelu: out size")
    _call("pe_gelu_fwd", x.data_ptr(), out.data_ptr(), x.numel(), _s())
    return out


def gelu_bwd(x, dy, out=None):
    x, dy = _dense(x, "x"), _dense(dy, "dy")
    _chk(x.numel() == dy.numel(), "gelu_bwd: sizes")
    if out is None:
        out = torch.empty_like(x)
    _call("pe_gelu_bwd", x.data_ptr(), dy.data_ptr(), out.data_ptr(), x.numel(), _s())
    return out
