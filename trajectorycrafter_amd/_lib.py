"""ctypes binding of libtcx_hip.so (the C ABI declared in include/tcx_hip.h).

The HIP library is the product path: there is NO fallback.  If the shared object is missing or a
call fails, a `TcxError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCX_LIB", os.path.join(_HERE, "libtcx_hip.so"))   # TCX_LIB: experiment builds only

TCX_BF16, TCX_F32 = 0, 1
TCX_ATTN_LOG2_SCORES = 1
TCX_ATTN_BOUND_PROVEN = 2
TCX_STEP_EULER, TCX_STEP_DPMPP_2M = 0, 1
TCX_PNDM_PRK_FIRST, TCX_PNDM_PRK_MID, TCX_PNDM_PRK_LAST, TCX_PNDM_PLMS4 = 0, 1, 2, 3

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (restype is int unless noted); mirrors include/tcx_hip.h one to one
SIGNATURES = {
    "tcx_version": [],
    "tcx_last_error_string": [],
    "tcx_device_info": [C.c_int, C.POINTER(_i32)],
    "tcx_attn_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32] + [_i64] * 12 + [_f32, _i32, _vp, _i32, _vp],
    "tcx_attn_fwd_ws": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32] + [_i64] * 12 + [_f32, _i32, _vp, _i32, _vp, _i64, _vp],
    "tcx_attn_fwd_workspace_bytes": [_i32] * 8,
    "tcx_qk_layernorm_rope": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp],
    "tcx_layernorm_modulate": [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp],
    "tcx_gated_residual": [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _vp, _vp, _i64, _i32, _vp],
    "tcx_bias_gelu_tanh": [_vp, _vp, _vp, _i64, _i32, _vp],
    "tcx_scale_bf16": [_vp, _vp, _i64, _f32, _vp],
    "tcx_silu_bf16": [_vp, _vp, _i64, _vp],
    "tcx_patchify": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "tcx_unpatchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "tcx_cfg_ddim_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _vp],
    "tcx_cfg_ddim_cog_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _vp],
    "tcx_cfg_ddim_eta_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _i32, _vp],
    "tcx_cfg_sigma_step": [_vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp, _vp, _vp, _vp, _i32, _vp],
    "tcx_div_bf16": [_vp, _vp, _i64, _f32, _vp],
    "tcx_cfg_pndm_step": [_vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "tcx_conv3d_cl": [_vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 16 + [_vp, _vp],
    "tcx_conv3d_route": [_i32] * 14,
    "tcx_avgpool_t": [_vp, _vp, _i32, _i32, _i64, _i32, _vp],
    "tcx_blend_ramp_bf16": [_vp, _vp, _i64, _i32, _i64, _i64, _i64, _i64, _i64, _vp],
    "tcx_scale_sqmax_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _f32, _vp, _vp],
    "tcx_gemm_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _vp],
    "tcx_warp_forward": [_vp] * 10 + [_i32, _i32, _i32, _i32, _vp],
    "tcx_bilinear_splat": [_vp] * 7 + [_i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "tcx_groupnorm_stats": [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _i32, _vp],
    "tcx_groupnorm_spatialnorm_silu": [_vp] * 7 + [_i32] * 9 + [_vp, _i32, _vp],
    "tcx_ncthw_to_cl": [_vp, _vp, _i32, _i32, _i64, _f32, _vp],
    "tcx_cl_to_ncthw_frames": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp],
}


class TcxError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libtcx_hip.so, failing loudly (no CPU / torch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch must load ITS HIP runtime first: libtcx_hip.so then binds to that already-loaded libamdhip64 (same
        # soname) instead of pulling a second runtime from /opt/rocm into the process ("no ROCm-capable device").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise TcxError(
                f"{LIB_PATH} is missing: the HIP extension is the only compute path. "
                "Build it with `python -m trajectorycrafter_amd.build` (needs hipcc, gfx950).")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise TcxError(f"libtcx_hip.so does not export {name}; rebuild it") from e
            fn.argtypes = argtypes
            fn.restype = (C.c_char_p if name == "tcx_last_error_string" else
                          C.c_int64 if name.endswith("_workspace_bytes") else C.c_int)
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tcx_last_error_string()
        raise TcxError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
