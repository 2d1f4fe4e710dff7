"""ctypes binding of libttvdm.so (include/ttvdm.h).  The product path has NO fallback: if the
library is missing or a call fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("TT_LIBTTVDM") or os.path.join(CSRC, "libttvdm.so")      # TT_LIBTTVDM: A/B builds of the same ABI

TT_BF16, TT_F16, TT_F32 = 0, 1, 2


class TtGemmArgs(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("k0", C.c_int32), ("k1", C.c_int32),
        ("lda0", C.c_int64), ("lda1", C.c_int64), ("w", C.c_void_p), ("ldw", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32), ("mode", C.c_int32),
        ("nimg", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("frames", C.c_int32), ("hw", C.c_int32),
        ("bias", C.c_void_p), ("acc_scale", C.c_float),
        ("rowvec", C.c_void_p), ("rowvec_rows", C.c_int32), ("ld_rowvec", C.c_int64),
        ("geglu", C.c_int32), ("residual", C.c_void_p), ("ld_res", C.c_int64),
        ("blend", C.c_void_p), ("ld_blend", C.c_int64), ("alpha", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int64), ("out_f32", C.c_int32),
        ("out_col_hw", C.c_int32), ("out_col_hwp", C.c_int32), ("dtype", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("ln_fold", C.c_int32), ("ln_eps", C.c_float), ("out_fp8", C.c_int32),
        ("rowvec_mod", C.c_int32),          # ABI 7: periodic row vector
        ("stats_out", C.c_void_p),          # ABI 8: per (row tile, column) sum / sum of squares of the stored output
        ("stats_seg", C.c_int32),           # ... rows of the consumer's GroupNorm segment (hint for the statistics tile height)
        ("gn_out", C.c_void_p), ("ld_gn", C.c_int64), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p),      # ABI 9: GroupNorm in the split-K reduction
        ("gn_eps", C.c_float), ("gn_silu", C.c_int32),
        ("presplit", C.c_int32),            # ABI 11: bit 0 a0 / a1, bit 1 w hold pre-split fp16 pairs (TT_F32 split16 mode)
    ]


class TtAttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("ldk", C.c_int64),
        ("vt", C.c_void_p), ("ldvt", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
        ("nseq", C.c_int32), ("lq", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("mask", C.c_int32), ("lk", C.c_int32), ("k_seq_stride", C.c_int32), ("v_seq_stride", C.c_int32),
        ("frames", C.c_int32), ("ctx_batches", C.c_int32), ("dtype", C.c_int32), ("batch0", C.c_int32), ("fp8", C.c_int32),
        # ABI 6: fused query projection of the cross-attention (qx != NULL: Q = LN(qx rows) wq^T + bq computed by the kernel)
        ("qx", C.c_void_p), ("ldqx", C.c_int64), ("wq", C.c_void_p), ("ldwq", C.c_int64), ("bq", C.c_void_p),
        ("qc", C.c_int32), ("ln_eps", C.c_float),
        ("v_rows", C.c_int32),              # ABI 10: `vt` is V itself, [key rows, ldvt]
    ]


class TtConvArgs(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("x1", C.c_void_p), ("c0", C.c_int32), ("c1", C.c_int32), ("ld0", C.c_int64), ("ld1", C.c_int64),
        ("w", C.c_void_p), ("ldw", C.c_int64),
        ("nimg", C.c_int32), ("h", C.c_int32), ("w_img", C.c_int32), ("n", C.c_int32),
        ("gn_scale", C.c_void_p), ("gn_shift", C.c_void_p), ("silu", C.c_int32),
        ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("rowvec_rows", C.c_int32), ("ld_rowvec", C.c_int64),
        ("residual", C.c_void_p), ("ld_res", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("dtype", C.c_int32),
    ]


_i32, _i64, _f32, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes): every symbol include/ttvdm.h declares
SIGNATURES = {
    "tt_abi_version": (C.c_int, []),
    "tt_target_arch": (C.c_char_p, []),
    "tt_last_error": (C.c_char_p, []),
    "tt_gemm": (C.c_int, [C.POINTER(TtGemmArgs), _vp]),
    "tt_gemm_set_streaming_square": (C.c_int, [C.c_int32]),
    "tt_gemm_set_big_tile": (C.c_int, [C.c_int32]),
    "tt_gemm_set_f32_split": (C.c_int, [C.c_int32]),
    "tt_gemm_plan": (C.c_int, [C.POINTER(TtGemmArgs), C.POINTER(C.c_int32)]),
    "tt_gemm_set_tile_override": (C.c_int, [_i32]),
    "tt_gemm_ws_bytes": (_sz, [C.POINTER(TtGemmArgs)]),
    "tt_attention": (C.c_int, [C.POINTER(TtAttnArgs), _vp]),
    "tt_conv3x3_supported": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "tt_conv3x3": (C.c_int, [C.POINTER(TtConvArgs), _vp]),
    "tt_temporal_attention": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tt_groupnorm_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "tt_groupnorm_small_supported": (C.c_int, [_i32, _i32, _i32]),
    "tt_groupnorm_small": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _i64, _i32, _vp]),
    "tt_groupnorm_stats": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _f32, _vp, _vp, _vp, _sz, _i32, _vp]),
    "tt_groupnorm_apply": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _vp]),
    "tt_groupnorm_tiles_supported": (C.c_int, [_i32, _i32, _i32, _i32]),
    "tt_groupnorm_tiles": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _i64, _i32, _vp]),
    "tt_gemm_stats_rows": (C.c_int32, [C.POINTER(TtGemmArgs)]),
    "tt_gemm_gn_fused": (C.c_int32, [C.POINTER(TtGemmArgs)]),
    "tt_layernorm": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp]),
    "tt_zero_sum_round": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "tt_small_linear": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "tt_timestep_embedding": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _vp]),
    "tt_prep_model_input": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "tt_cfg_euler_step": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "tt_cfg3_euler_step": (C.c_int, [_vp, _i32, _vp, _vp, C.c_float, _vp, _i32, _i32, _i32, _i32, _vp]),
    "tt_nchw_to_tokens": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "tt_tokens_to_nchw": (C.c_int, [_vp, _i32, _i64, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "tt_add_scaled": (C.c_int, [_vp, _vp, _f32, _vp, _i64, _i32, _vp]),
    "tt_add_rowvec": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "tt_softmax_rows": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _vp]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libttvdm.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libttvdm.so failed:\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}")
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `make -C {CSRC}` (or __graft_entry__.build()). "
                           "There is no CPU/PyTorch fallback for the denoise path.")
    # torch first: its wheel bundles its own HIP runtime (libamdhip64).  If libttvdm.so is dlopen'ed before torch, the system
    # runtime it links against is the one that initialises, torch then brings a second copy, and kernels launched through this
    # library fail with "no ROCm-capable device is detected" (seen with build() + smoke() in one process).  With torch loaded
    # first the library's DT_NEEDED entry resolves to the runtime that owns torch's streams and allocations.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI mismatch with include/ttvdm.h
        fn.restype, fn.argtypes = res, args
    if lib.tt_abi_version() != 11:
        raise RuntimeError("libttvdm.so ABI version mismatch")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().tt_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed ({code}): {msg}")
