"""ctypes binding of librt_reptext_hip.so — the C ABI declared in include/reptext_hip.h.

This is the only way compute leaves Python: there is NO CPU or eager-PyTorch fallback. If the shared
library has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C csrc``)
importing a symbol raises ``NativeLibraryMissing``; if a call is rejected ``NativeCallError`` carries the
RT_E_* / hipError code.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "librt_reptext_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

RT_GEMM_MAX_GROUPS = 4
ABI_VERSION = 8


class NativeLibraryMissing(RuntimeError):
    pass


class NativeCallError(RuntimeError):
    def __init__(self, fn: str, code: int):
        names = {-1: "RT_E_BADARG", -2: "RT_E_ALIGN", -3: "RT_E_SHAPE"}
        super().__init__(f"{fn} failed: {names.get(code, 'hipError ' + str(code))} ({code})")
        self.code = code


class GemmGroup(C.Structure):
    """Mirror of ``rt_gemm_group`` (include/reptext_hip.h). Field order and widths are ABI."""

    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("gate", C.c_void_p), ("res", C.c_void_p), ("add2", C.c_void_p), ("rowscale", C.c_void_p),
        ("lda", C.c_int64), ("ldw", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("ld2", C.c_int64), ("gate_ld", C.c_int64),
        ("strideA", C.c_int64), ("strideC", C.c_int64), ("strideR", C.c_int64), ("stride2", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("rows_per_batch", C.c_int32), ("gelu_from", C.c_int32), ("out_f32", C.c_int32),
        ("alpha", C.c_float),
        ("a_scale", C.c_void_p), ("w_scale", C.c_void_p),
        ("stride_rowscale", C.c_int64),
        ("a_bscale", C.c_void_p), ("a_bscale_plane", C.c_int64), ("a_bscale_rows", C.c_int64),
        ("c8", C.c_void_p), ("c_bscale", C.c_void_p),
        ("ldc8", C.c_int64), ("stride_c8", C.c_int64), ("c_bscale_plane", C.c_int64), ("c_bscale_rows", C.c_int64),
        ("c8_from", C.c_int32), ("c_bscale_k0", C.c_int32),
        ("conv_ks", C.c_int32), ("conv_cin", C.c_int32), ("conv_w2", C.c_int32), ("conv_h2", C.c_int32),
        ("conv_inv_w2", C.c_float), ("conv_inv_h2", C.c_float),
    ]


_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

# name -> argtypes (restype is int except rt_version). Every symbol include/reptext_hip.h declares.
SIGNATURES = {
    "rt_abi_version": [],
    "rt_gemm_bf16": [C.POINTER(GemmGroup), _i32, _vp],
    "rt_gemm_fp8": [C.POINTER(GemmGroup), _i32, _vp],
    "rt_gemm_tile_mode": [_i32],
    "rt_quantize_rows_fp8": [_vp, _i64, _i32, _vp, _i64, _vp, _i32, _i32, _vp],
    "rt_quantize_mx_fp8": [_vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "rt_layernorm_modulate_fp8": [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "rt_gemv_bf16w": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "rt_timestep_embedding": [_vp, _vp, _i32, _i32, _vp],
    "rt_rope_table": [_vp, _vp, _vp, _i32, C.POINTER(_i32), _f32, _vp],
    "rt_layernorm_modulate": [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "rt_qk_rmsnorm_rope": [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "rt_attention_ws_bytes": [_i32, _i32, _i32],
    "rt_attention_variant": [_i32],
    "rt_attention_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp, _i64, _vp],
    "rt_attention_fp8_vt_bytes": [_i32, _i32, _i32],
    "rt_attention_fp8_prep": [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "rt_attention_fp8_fwd": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _vp],
    "rt_attention_fp8_fwd_mx": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp],
    "rt_embedding_gather": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "rt_rmsnorm_rows": [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _f32, _vp],
    "rt_softmax_rows_bias": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "rt_gated_mul": [_vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "rt_quick_gelu": [_vp, _i64, _vp],
    "rt_euler_step": [_vp, _vp, _f32, _i64, _vp],
    "rt_euler_step_f32": [_vp, _vp, _vp, _f32, _i64, _vp],
    "rt_cfg_mix": [_vp, _vp, _vp, _f32, _i64, _vp],
    "rt_pack_latents": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "rt_unpack_latents": [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    "rt_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "rt_cast_bf16_to_f32": [_vp, _vp, _i64, _vp],
    "rt_silu_split_bf16": [_vp, _vp, _vp, _i64, _i32, _vp],
    "rt_masked_accumulate": [_vp, _vp, _vp, _f32, _i32, _i32, _i32, _i32, _i32, _vp],
}
# AutoencoderKL entries (csrc/vae.hip)
SIGNATURES.update({
    "rt_groupnorm_ws_bytes": [_i32, _i32, _i32, _i32],
    "rt_groupnorm_silu_nhwc": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "rt_conv2d_nhwc": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "rt_softmax_rows": [_vp, _vp, _i32, _i32, _f32, _vp],
    "rt_vae_attention": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _vp],
    "rt_transpose_bf16": [_vp, _vp, _i32, _i32, _i64, _i64, _vp],
    "rt_image_out": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "rt_nchw_to_haloed_nhwc": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "rt_haloed_nhwc_to_nchw": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "rt_unpack_latents_haloed": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp],
    "rt_resize2d": [_vp, _i32, _f32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp],
    "rt_glyph_blend": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
})
# hint preparation on the device (csrc/hints.hip)
SIGNATURES.update({
    "rt_conv2d_variant": [_i32],
    "rt_canny_ws_bytes": [_i32, _i32],
    "rt_canny_u8": [_vp, _i32, _i32, _i32, _f32, _f32, _vp, _i32, _i32, _vp, _i64, _vp],
    "rt_preprocess_u8": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
})

# entries that do not return a status code
RESTYPES = {"rt_canny_ws_bytes": C.c_int64, "rt_groupnorm_ws_bytes": C.c_int64, "rt_attention_fp8_vt_bytes": C.c_int64, "rt_attention_ws_bytes": C.c_int64}

_lib = None


def library_present() -> bool:
    return os.path.isfile(LIB_PATH)


def load():
    """Load (once) and type the shared library. Raises NativeLibraryMissing when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not library_present():
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path."
        )
    lib = C.CDLL(LIB_PATH)
    lib.rt_version.restype = C.c_char_p
    lib.rt_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, C.c_int)
    if lib.rt_abi_version() != ABI_VERSION:
        raise NativeLibraryMissing(f"{LIB_NAME} ABI {lib.rt_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(fn: str, code: int) -> None:
    if code != 0:
        raise NativeCallError(fn, code)
