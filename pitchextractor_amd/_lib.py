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
_z = C.c_size_t
_u64 = C.c_ulonglong
_pp = C.POINTER(C.c_void_p)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes).  Mirrors include/pitchextractor_hip.h one to one;
# tests/test_abi.py checks the two against each other and against the .so.
PROTOTYPES = {
    "pe_abi_version": (_i, []),
    "pe_device_count": (_i, []),
    "pe_stream_create_low_priority": (_i, [C.POINTER(_p)]),
    "pe_mel_plan_create": (_i, [C.POINTER(_p), _i, _i, _i, _i, _i, _f, _f]),
    "pe_mel_plan_destroy": (_i, [_p]),
    "pe_mel_num_frames": (_i, [_p, _i]),
    "pe_mel_forward": (_i, [_p, _p, _i, _i, _l, _p, _l, _l, _l, _i, _i, _f, _f, _f, _f, _p]),
    "pe_mel_forward_ragged": (_i, [_p, _p, _i, _i, _l, _p, _p, _p, _l, _l, _l, _i, _i, _f, _f, _f, _f, _p]),
    "pe_gemm_nt": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "pe_gemm_nt_bf16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "pe_gemm_nt_x3": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "pe_gemm_nt_h2": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p, _p, _p]),
    "pe_absmax": (_i, [_p, _l, _i, _l, _p, _p]),
    "pe_absmax_segments": (_i, [_p, _p, _p, _i, _p, _p]),
    "pe_gemm_tn_workspace_bytes": (_z, [_i, _i, _i]),
    "pe_gemm_tn": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p]),
    "pe_gemm_tn_x3": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p]),
    "pe_gemm_tn_bf16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p]),
    "pe_gemm_tn_h2": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p, _p, _p]),
    "pe_transpose2d": (_i, [_p, _p, _i, _i, _p]),
    "pe_conv3x3_repack": (_i, [_p, _p, _p, _i, _i, _p]),
    "pe_conv3x3_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pe_conv3x3_fwd_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pe_conv3x3_fwd_x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pe_conv3x3_fwd_h2": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "pe_wfrag_bytes": (_z, [_i, _i, _i]),
    "pe_wfrag_pack_h2": (_i, [_p, _l, _i, _i, _p, _p, _p]),
    "pe_conv3x3_fwd_wf_h2": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "pe_conv3x3_wgrad_h2": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p, _p, _p]),
    "pe_lstm_whh_grad_h2": (_i, [_p, _p, _l, _p, _i, _i, _i, _i, _p, _z, _p, _p, _p]),
    "pe_wfrag_pack": (_i, [_p, _l, _i, _i, _i, _p, _p]),
    "pe_conv3x3_wf_supported": (_i, [_i, _i, _i]),
    "pe_conv3x3_fwd_wf_x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "pe_conv3x3_fwd_wf_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "pe_conv3x3_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "pe_conv3x3_wgrad": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "pe_conv3x3_wgrad_x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "pe_conv3x3_wgrad_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "pe_conv3x3_c1_stat_parts": (_i, [_i, _i, _i]),
    "pe_conv3x3_c1_fwd": (_i, [_p, _l, _l, _l, _p, _p, _i, _i, _i, _p, _p]),
    "pe_conv3x3_c1_wgrad": (_i, [_p, _l, _l, _l, _p, _p, _i, _i, _i, _p, _z, _p]),
    "pe_bn_workspace_bytes": (_z, [_i]),
    "pe_bn_train_stats": (_i, [_p, _l, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "pe_conv3x3_wf_stat_parts": (_i, [_i, _i, _i]),
    "pe_bn_finalize_stats": (_i, [_p, _i, _l, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "pe_bn_eval_affine": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "pe_bn_act_pool_fwd": (_i, [_p, _p, _p, _f, _p, _l, _i, _i, _i, _l, _i, _p, _p]),
    "pe_bn_act_pool_bwd": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _l, _i, _i, _i, _l, _i, _p, _z, _p, _p]),
    "pe_maxpool_fwd": (_i, [_p, _p, _l, _i, _i, _i, _l, _i, _p, _p]),
    "pe_maxpool_bwd_add": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _l, _i, _p, _p]),
    "pe_dropout_fwd": (_i, [_p, _l, _p, _l, _p, _p, _l, _i, _f, _u64, _u64, _p]),
    "pe_nhwc_to_seq": (_i, [_p, _l, _i, _p, _l, _i, _p]),
    "pe_seq_to_nhwc": (_i, [_p, _p, _l, _i, _l, _i, _i, _p]),
    "pe_copy2d": (_i, [_p, _l, _p, _l, _l, _i, _i, _p]),
    "pe_lstm_fwd": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _p]),
    "pe_lstm_bwd": (_i, [_i, _pp, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _p]),
    "pe_lstm_persistent_sync_bytes": (_z, [_i, _i]),
    "pe_lstm_persistent_supported": (_i, [_i, _i, _i]),
    "pe_lstm_fwd_persistent_x3": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _p, _p]),
    "pe_lstm_bwd_persistent_dbias_rows": (_i, [_i, _i, _i, _i, _l]),
    "pe_lstm_configure_stamps": (_i, [_i]),
    "pe_lstm_bwd_persistent_x3": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _pp, _pp, _p, _p]),
    "pe_lstm_fwd_persistent_bf16": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _p, _p]),
    "pe_lstm_bwd_persistent_bf16": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _pp, _pp, _p, _p]),
    "pe_lstm_whh_grad_workspace_bytes": (_z, [_i, _i, _i]),
    "pe_lstm_whh_grad": (_i, [_p, _p, _l, _p, _i, _i, _i, _i, _p, _z, _p]),
    "pe_lstm_whh_grad_x3": (_i, [_p, _p, _l, _p, _i, _i, _i, _i, _p, _z, _p]),
    "pe_lstm_whh_grad_bf16": (_i, [_p, _p, _l, _p, _i, _i, _i, _i, _p, _z, _p]),
    "pe_colsum_workspace_bytes": (_z, [_i]),
    "pe_colsum": (_i, [_p, _l, _i, _l, _p, _p, _p, _z, _p]),
    "pe_head_fwd": (_i, [_p, _l, _p, _p, _i, _p, _l, _i, _p]),
    "pe_head_bwd_workspace_bytes": (_z, [_i]),
    "pe_head_bwd": (_i, [_p, _l, _p, _p, _i, _p, _l, _p, _p, _l, _i, _p, _z, _p]),
    "pe_f0_sil_loss": (_i, [_p, _p, _p, _p, _f, _l, _f, _p, _p, _p, _p]),
    "pe_bgemm": (_i, [_i, _p, _l, _l, _l, _p, _l, _l, _l, _p, _l, _l, _l, _i, _i, _i, _i, _i, _f, _i, _p]),
    "pe_attn_supported": (_i, [_i, _i]),
    "pe_attn_fwd": (_i, [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u64, _u64, _p]),
    "pe_attn_bwd": (_i, [_p, _l, _p, _p, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "pe_attn_fwd_bf16": (_i, [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _u64, _u64, _p]),
    "pe_attn_bwd_bf16": (_i, [_p, _l, _p, _p, _l, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "pe_softmax_fwd": (_i, [_p, _l, _i, _f, _p]),
    "pe_softmax_bwd": (_i, [_p, _p, _l, _i, _f, _p]),
    "pe_layernorm_fwd": (_i, [_p, _p, _p, _i, _p, _p, _f, _p, _p, _p, _p, _l, _i, _p]),
    "pe_layernorm_bwd_workspace_bytes": (_z, [_i]),
    "pe_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _z, _p]),
    "pe_gelu_fwd": (_i, [_p, _p, _l, _p]),
    "pe_gelu_bwd": (_i, [_p, _p, _p, _l, _p]),
    "pe_layernorm_dropout_fwd": (_i, [_p, _p, _p, _i, _p, _p, _f, _p, _p, _p, _p, _l, _i, _p, _p, _f, _u64, _u64, _p]),
    "pe_layernorm_bwd_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _l, _i, _p, _z, _p]),
    "pe_gelu_dropout_fwd": (_i, [_p, _p, _l, _p, _p, _f, _u64, _u64, _p]),
    "pe_gelu_dropout_bwd": (_i, [_p, _p, _p, _f, _p, _l, _p]),
    "pe_resample_plan_create": (_i, [C.POINTER(_p), _i, _i, _i, _f]),
    "pe_resample_plan_destroy": (_i, [_p]),
    "pe_resample_out_len": (_l, [_p, _l]),
    "pe_resample_forward": (_i, [_p, _p, _i, _i, _l, _p, _l, _i, _p]),
    "pe_f0_bins_ce_workspace_bytes": (_z, [_l]),
    "pe_f0_bins_ce_loss": (_i, [_p, _l, _i, _p, _p, _p, _f, _l, _f, _p, _p, _l, _p, _p, _z, _p]),
    "pe_gemm_nt_f16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "pe_gemm_tn_f16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p]),
    "pe_conv3x3_fwd_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pe_conv3x3_fwd_wf_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "pe_conv3x3_wgrad_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "pe_lstm_fwd_persistent_f16": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _p, _p]),
    "pe_lstm_bwd_persistent_f16": (_i, [_i, _pp, _pp, _pp, _pp, _ip, _l, _i, _i, _i, _pp, _pp, _p, _p]),
    "pe_lstm_whh_grad_f16": (_i, [_p, _p, _l, _p, _i, _i, _i, _i, _p, _z, _p]),
    "pe_wfrag_pack_f16": (_i, [_p, _l, _i, _i, _p, _p]),
    "pe_conv3x3_c1_fwd_a16": (_i, [_p, _l, _l, _l, _p, _p, _i, _i, _i, _p, _p]),
    "pe_conv3x3_c1_wgrad_a16": (_i, [_p, _l, _l, _l, _p, _p, _i, _i, _i, _p, _z, _p]),
    "pe_conv3x3_fwd_bf16_a16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "pe_conv3x3_fwd_wf_bf16_a16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "pe_conv3x3_wgrad_bf16_a16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _z, _p]),
    "pe_gemm_nt_bf16_a16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p]),
    "pe_gemm_tn_bf16_a16": (_i, [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _z, _p]),
    "pe_bn_train_stats_a16": (_i, [_p, _l, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "pe_bn_act_pool_fwd_a16": (_i, [_p, _p, _p, _f, _p, _l, _i, _i, _i, _l, _i, _p]),
    "pe_bn_act_pool_bwd_a16": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _l, _i, _i, _i, _l, _i, _p, _z, _p]),
    "pe_maxpool_fwd_a16": (_i, [_p, _p, _l, _i, _i, _i, _l, _i, _p, _p]),
    "pe_maxpool_bwd_add_a16": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _l, _i, _p]),
    "pe_dropout_fwd_a16": (_i, [_p, _l, _p, _l, _p, _p, _l, _i, _f, _u64, _u64, _p]),
    "pe_nhwc_to_seq_a16": (_i, [_p, _l, _i, _p, _l, _i, _p]),
    "pe_seq_to_nhwc_a16": (_i, [_p, _p, _l, _i, _l, _i, _i, _p]),
    "pe_copy2d_a16": (_i, [_p, _l, _p, _l, _l, _i, _i, _p]),
    "pe_nonfinite_flag": (_i, [_p, _l, _p, _p]),
    "pe_adamw_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _d, _d, _f, _p, _p]),
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
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's, but its
    # libraries ask for it by the unversioned file name, so it is loaded even when /opt/rocm's copy already
    # is).  Loading this library first therefore left TWO runtimes in the process, and the second to open the
    # device reported hipErrorNoDevice (seen with build() followed by smoke() in one interpreter).  With torch
    # imported first, this library's DT_NEEDED libamdhip64.so.7 resolves to the copy torch already loaded.
    import torch  # noqa: F401
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
