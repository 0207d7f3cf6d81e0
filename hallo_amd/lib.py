"""ctypes binding of libhallo_amd.so (C ABI declared in include/hallo_amd.h).

The product path has no CPU or PyTorch fallback: if the HIP library cannot be loaded this
module raises, and every operator wrapper raises on a non-zero status code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhallo_amd.so")

F16, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU, ACT_GELU_PRE = 0, 1, 2, 3, 4


class HalloLibraryError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("batch", C.c_int),
        ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64), ("stride_r", C.c_int64),
        ("bias", C.c_void_p),
        ("bias_per_row", C.c_int),
        ("bias2", C.c_void_p),
        ("bias2_rows_per_group", C.c_int),
        ("bias2_ld", C.c_int64),
        ("rowscale", C.c_void_p),
        ("residual", C.c_void_p),
        ("ldr", C.c_int64),
        ("alpha", C.c_float),
        ("act", C.c_int),
        ("geglu", C.c_int),
        ("out_f32", C.c_int),
        ("dtype", C.c_int),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("lead_cols", C.c_int),
        ("lead_alpha", C.c_float),
        ("ln_colsum", C.c_void_p),
        ("ln_eps", C.c_float),
        ("ln_stats", C.c_void_p),
        ("row_parts", C.c_void_p),
        ("ln_parts", C.c_int),
        ("workspace_zeroed", C.c_int),
        ("kv_out", C.c_void_p), ("kv_col0", C.c_int), ("kv_rows_per_image", C.c_int), ("kv_tensor_stride", C.c_int64),
    ]


class GemmFp8Desc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("a_scale", C.c_void_p), ("w_scale", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("alpha", C.c_float), ("lead_cols", C.c_int), ("lead_alpha", C.c_float), ("dtype", C.c_int),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
        ("n_img", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int),
        ("stride", C.c_int), ("pad_t", C.c_int), ("pad_l", C.c_int), ("upsample", C.c_int),
        ("bias", C.c_void_p),
        ("bias2", C.c_void_p),
        ("bias2_rows_per_group", C.c_int),
        ("bias2_ld", C.c_int64),
        ("residual", C.c_void_p),
        ("ldr", C.c_int64),
        ("ldy", C.c_int64),
        ("alpha", C.c_float),
        ("act", C.c_int),
        ("dtype", C.c_int),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k1", C.c_void_p), ("v1", C.c_void_p), ("k2", C.c_void_p), ("v2", C.c_void_p),
        ("o", C.c_void_p),
        ("batch", C.c_int), ("heads", C.c_int), ("head_dim", C.c_int), ("Lq", C.c_int), ("Lkv1", C.c_int),
        ("Lkv2", C.c_int),
        ("q_bs", C.c_int64), ("q_rs", C.c_int64), ("k1_bs", C.c_int64), ("k1_rs", C.c_int64),
        ("v1_bs", C.c_int64), ("v1_rs", C.c_int64), ("k2_bs", C.c_int64), ("k2_rs", C.c_int64),
        ("v2_bs", C.c_int64), ("v2_rs", C.c_int64), ("o_bs", C.c_int64), ("o_rs", C.c_int64),
        ("kv2_batch_div", C.c_int), ("kv2_batch_mod", C.c_int), ("kv2_first_batch", C.c_int),
        ("scale", C.c_float),
        ("dtype", C.c_int),
        ("o_rowscale", C.c_void_p), ("o_rowscale_head_div", C.c_int), ("o_rowscale_stride", C.c_int64),
        ("q_prescaled", C.c_int),
        ("kv1_hs", C.c_int64), ("kv2_hs", C.c_int64),
    ]


# symbol -> (restype, argtypes); every symbol include/hallo_amd.h declares is listed here and
# tests/test_abi.py checks that the built library exports each of them.
SYMBOLS = {
    "hallo_abi_version": (C.c_int, []),
    "hallo_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "hallo_get_option": (C.c_int, [C.c_char_p]),
    "hallo_option_names": (C.c_char_p, []),
    "hallo_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "hallo_gemm_fp8": (C.c_int, [C.POINTER(GemmFp8Desc), C.c_void_p]),
    "hallo_quant_rows_fp8": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_float, C.c_int, C.c_void_p]),
    "hallo_conv3x3_nhwc": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "hallo_attention": (C.c_int, [C.POINTER(AttnDesc), C.c_void_p]),
    "hallo_temporal_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_int, C.c_void_p]),
    "hallo_temporal_attention_lead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_int, C.c_void_p]),
    "hallo_groupnorm_chunks": (C.c_int, [C.c_int]),
    "hallo_groupnorm_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "hallo_groupnorm_nhwc2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "hallo_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hallo_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "hallo_copy2d": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                               C.c_void_p]),
    "hallo_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "hallo_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float,
                                         C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "hallo_row_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "hallo_gemm_fuses_row_stats": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "hallo_gemm_kv_split_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "hallo_gemm4_schedule": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    "hallo_face_xattn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "hallo_face_xattn_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "hallo_ff320_pack_bytes": (C.c_int64, []),
    "hallo_ff320": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                              C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "hallo_frames_to_uint8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "hallo_w2v_conv0_workspace": (C.c_int64, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "hallo_w2v_conv0_gn_gelu": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "hallo_lerp_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hallo_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hallo_cfg_ddim_step": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]),
}

_lib = None


def load(path=None):
    """Load the HIP operator library.  Raises HalloLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HalloLibraryError(
            f"{p} not found: build it with `python -m hallo_amd.build` (hipcc --offload-arch=gfx950). "
            "hallo_amd has no CPU / PyTorch fallback.")
    # The process must hold ONE HIP runtime: torch bundles its own libamdhip64.so.7 and the streams / device
    # pointers handed to this library come from it.  Importing torch first makes the dynamic loader bind
    # libhallo_amd.so's NEEDED libamdhip64.so.7 to the already-loaded copy instead of /opt/rocm's.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the host's ROCm install
        raise HalloLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HalloLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise HalloLibraryError(f"{what} failed with status {status}")
