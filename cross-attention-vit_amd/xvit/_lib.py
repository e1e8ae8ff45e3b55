"""ctypes binding of libxvit_hip.so (the C ABI declared in include/xvit.h).

There is NO fallback: if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XVIT_LIB") or os.path.join(_HERE, "libxvit_hip.so")   # XVIT_LIB: A/B-test another build of the same ABI

i32, i64, f32, u64, vp = C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_void_p


class GemmArgs(C.Structure):
    """struct xvit_gemm_args (include/xvit.h)."""
    _fields_ = [(n, i32) for n in (
        "layout", "M", "N", "K", "batch", "c_dtype", "act", "accumulate", "split_k",
        "res_row_mod", "res_row_off", "out_seg_rows", "out_seg_skip", "out_row_off", "aux_mode")] + [
        (n, vp) for n in ("A", "B", "C", "bias", "residual", "aux")] + [
        (n, i64) for n in ("lda", "ldb", "ldc", "ldr", "ldaux",
                           "stride_a", "stride_b", "stride_c", "stride_bias", "stride_r", "stride_aux")] + [
        ("workspace", vp), ("workspace_bytes", i64), ("colsum", vp), ("dropout_p", f32), ("reserved2", i32), ("dropout_seed", u64)]


class PatchGeom(C.Structure):
    """struct xvit_patch_geom (include/xvit.h)."""
    _fields_ = [(n, i32) for n in ("B", "M", "D", "H", "W", "dp", "hp", "wp", "cls_rows")]


# name -> argtypes; every function returns int except the two noted below
SIGNATURES = {
    "xvit_gemm": [C.POINTER(GemmArgs), vp],
    "xvit_small_linear_fwd": [vp, i64, vp, vp, vp, i32, i32, i32, vp],
    "xvit_small_linear_bwd": [vp, vp, i64, vp, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, vp],
    "xvit_layernorm_fwd": [vp, vp, i64, i32, i64, vp, vp, f32, vp, i64, vp, i64, vp, vp, i32, i32, vp],
    "xvit_linear_f32": [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp, i64, vp, i64, f32, u64, vp, i64, vp],
    "xvit_layernorm_bwd": [vp, i64, vp, vp, i64, i32, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, vp, i64, vp],
    "xvit_attn_fwd": [vp, vp, vp, i64, i64, vp, i64, i64, vp, i32, i32, i32, i32, f32, f32, u64, vp, i64, vp],
    "xvit_attn_bwd": [vp, vp, vp, i64, i64, vp, vp, i64, i64, vp, vp, i64, vp, vp, vp, i32, i32, i32, i32, f32, f32, u64, vp],
    "xvit_attn_fwd_fp8": [vp, vp, vp, i64, i64, vp, i64, i64, vp, i32, i32, i32, i32, f32, vp, i64, vp],
    "xvit_cls_xattn_fwd": [vp, i64, vp, i64, vp, vp, i64, i64, vp, i64, vp, i64, vp, i32, i32, i32, i32, f32, f32, u64, vp],
    "xvit_cls_xattn_bwd": [vp, i64, vp, vp, i64, i64, vp, vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, f32, f32, u64, vp],
    "xvit_head_rows": [vp, i64, vp, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp],
    "xvit_head_cols": [vp, i64, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "xvit_head_bias_grad": [vp, i64, vp, i64, vp, i32, i32, i32, vp],
    "xvit_head_wgrad": [vp, i64, vp, i64, i64, vp, i64, vp, i64, i32, i32, i32, vp],
    "xvit_cls_softmax_fwd": [vp, i64, vp, i64, vp, i32, i32, i32, f32, vp, f32, u64, vp],
    "xvit_cls_softmax_bwd": [vp, i64, vp, vp, i64, vp, vp, i64, i32, i32, i32, f32, f32, u64, vp],
    "xvit_xattn_kv_dgrad": [vp, vp, vp, i64, i32, i32, i32, i32, vp],
    "xvit_patchify": [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i32, i32, i64, vp],
    "xvit_patch_embed_supported": [C.POINTER(PatchGeom), i32],
    "xvit_patch_embed_fwd": [vp, C.POINTER(PatchGeom), vp, i64, vp, vp, i64, vp, i64, i32, vp],
    "xvit_patch_embed_wgrad": [vp, C.POINTER(PatchGeom), vp, i64, vp, i64, i32, vp, i64, vp],
    "xvit_cls_row_fwd": [vp, vp, vp, i32, i32, i32, vp],
    "xvit_embed_bwd": [vp, vp, vp, i32, i32, i32, vp],
    "xvit_cast_f32_bf16": [vp, vp, i64, vp],
    "xvit_add_cast_f32_bf16": [vp, vp, vp, vp, i64, vp],
    "xvit_rows_combine": [vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, i32, i64, i32, i32, vp],
    "xvit_colsum": [vp, i32, i64, vp, i32, i32, i32, vp, i64, vp],
    "xvit_dropout": [vp, vp, i32, i64, f32, u64, vp],
    "xvit_cu_trace": [vp, i32, i32, vp],
    "xvit_binary_metrics_step": [vp, i64, vp, i32, i32, vp, vp],
    "xvit_mean_ce": [vp, vp, f32, vp, vp, vp, i32, i32, i32, vp],
    "xvit_resize_pad_crop_i16": [vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp],
    "xvit_set_option": [C.c_char_p, i32],
    "xvit_set_dropout_epoch": [C.c_void_p],
    "xvit_adam_step": [vp, vp, i32, f32, f32, f32, f32, f32, i32, f32, vp],
}
EXPORTS = sorted(list(SIGNATURES) + ["xvit_version", "xvit_last_error_string", "xvit_gemm_workspace_bytes", "xvit_linear_f32_workspace_bytes",
                                    "xvit_colsum_workspace_bytes", "xvit_layernorm_bwd_workspace_bytes", "xvit_patch_embed_wgrad_workspace_bytes", "xvit_attn_fp8_workspace_bytes",
                                    "xvit_attn_fwd_workspace_bytes", "xvit_attn_bwd_workspace_bytes"])

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"xvit: {LIB_PATH} is missing. Build it with `python cross-attention-vit_amd/build.py` "
                "(hipcc, gfx950). There is no CPU or PyTorch fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        lib.xvit_gemm_workspace_bytes.argtypes = [C.POINTER(GemmArgs)]
        lib.xvit_gemm_workspace_bytes.restype = C.c_int64
        lib.xvit_linear_f32_workspace_bytes.argtypes = [i32, i32, i32]
        lib.xvit_linear_f32_workspace_bytes.restype = C.c_int64
        for name in ("xvit_colsum_workspace_bytes", "xvit_layernorm_bwd_workspace_bytes"):
            getattr(lib, name).argtypes = [i32, i32]
            getattr(lib, name).restype = C.c_int64
        lib.xvit_attn_fp8_workspace_bytes.argtypes = [i32, i32, i32, i32]
        lib.xvit_attn_fp8_workspace_bytes.restype = C.c_int64
        for name in ("xvit_attn_fwd_workspace_bytes", "xvit_attn_bwd_workspace_bytes"):
            getattr(lib, name).argtypes = [i32, i32, i32]
            getattr(lib, name).restype = C.c_int64
        lib.xvit_patch_embed_wgrad_workspace_bytes.argtypes = [C.POINTER(PatchGeom), i32]
        lib.xvit_patch_embed_wgrad_workspace_bytes.restype = C.c_int64
        lib.xvit_version.restype = C.c_int
        lib.xvit_last_error_string.restype = C.c_char_p
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().xvit_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
