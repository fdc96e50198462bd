"""ctypes binding of libcalmvit_hip.so (the C-ABI declared in include/calm_vit.h).

There is NO fallback: if the shared object is missing or a symbol cannot be resolved, loading
raises and every op of the package fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CALM_VIT_LIB: A/B another build of the same library (kernel tuning); never a different implementation
LIB_PATH = os.environ.get("CALM_VIT_LIB") or os.path.join(HERE, "libcalmvit_hip.so")

ACT_NONE, ACT_GELU, ACT_GELU_BWD = 0, 1, 2
F32 = 0
ST_F32, ST_BF16, ST_FP8_E4M3, ST_FP8_E5M2 = 0, 1, 2, 3   # storage type of a tensor in HBM (CALM_ST_*)
E_INVAL, E_LAYOUT, E_UNSUPP = -1, -2, -3      # CALM_E_*
ABI_VERSION = 7          # CALM_ABI_VERSION of include/calm_vit.h

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class GemmArgs(C.Structure):
    """struct calm_gemm_args (include/calm_vit.h)."""
    _fields_ = [
        ("A", _p), ("B", _p), ("C", _p),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("batch0", _i32), ("batch1", _i32),
        ("a_rs", _i64), ("a_cs", _i64), ("a_b0", _i64), ("a_b1", _i64),
        ("b_rs", _i64), ("b_cs", _i64), ("b_b0", _i64), ("b_b1", _i64),
        ("c_rs", _i64), ("c_b0", _i64), ("c_b1", _i64),
        ("alpha", _f32),
        ("inv_scale", _p), ("bias", _p), ("col_scale", _p),
        ("residual", _p), ("r_rs", _i64), ("r_b0", _i64), ("r_b1", _i64),
        ("C_pre", _p), ("aux", _p),
        ("act", _i32), ("accumulate", _i32), ("reduce_batch", _i32), ("split_k", _i32), ("dtype", _i32),
        ("n_group", _i32),
        ("A_group", _p * 4), ("B_group", _p * 4), ("C_group", _p * 4), ("inv_scale_group", _p * 4),
        ("workspace", _p), ("workspace_bytes", _i64),
        ("a_type", _i32), ("b_type", _i32), ("c_type", _i32), ("aux_type", _i32), ("r_type", _i32), ("reserved_", _i32),
        ("a_dq", _p), ("b_dq", _p),
    ]


class GemmPlan(C.Structure):
    """struct calm_gemm_plan (calm_gemm_describe, ABI v7)."""
    _fields_ = [(n, _i32) for n in ("family", "tile_m", "tile_n", "tile_k", "tiles_m", "tiles_n", "k_slices", "items",
                                    "grid", "epi_unit", "uses_workspace", "threads")]


RED_LAYERNORM_BWD, RED_ROPE_BWD, RED_LATENT_FWD, RED_COLSUM, RED_CNN_BWD = range(5)     # CALM_RED_*


class OptimTensor(C.Structure):
    """struct calm_optim_tensor (64 bytes)."""
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("sn_u", _p), ("sn_v", _p),
                ("sn_sigma", _p), ("numel", _i64), ("rows", _i32), ("cols", _i32), ("chunk0", _i32), ("reserved", _i32)]


class OptimHparams(C.Structure):
    """struct calm_optim_hparams."""
    _fields_ = [("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32),
                ("max_norm", _f32), ("step", _i32)]


class CastEntry(C.Structure):
    """struct calm_cast_entry."""
    _fields_ = [("src", _p), ("dst", _p), ("numel", _i64), ("chunk0", _i32), ("reserved", _i32)]


class SnLayer(C.Structure):
    """struct calm_sn_layer."""
    _fields_ = [("w", _p), ("u", _p), ("v", _p), ("sigma", _p), ("rows", _i32), ("cols", _i32)]


class SnPlanInfo(C.Structure):
    """struct calm_sn_plan_info."""
    _fields_ = [("blob_bytes", _i64), ("scratch_floats", _i64), ("n_layers", _i32), ("n_work", _i32),
                ("n_work_a", _i32), ("reserved", _i32)]


# name -> (restype, argtypes); every symbol include/calm_vit.h declares
SIGNATURES = {
    "calm_abi_version": (_i32, []),
    "calm_build_info": (C.c_char_p, []),
    "calm_gemm": (_i32, [C.POINTER(GemmArgs), _p]),
    "calm_gemm_workspace_bytes": (_i64, [C.POINTER(GemmArgs)]),
    "calm_gemm_set_option": (C.c_int, [_i32, _i32]),
    "calm_gemm_describe": (_i32, [C.POINTER(GemmArgs), C.POINTER(GemmPlan)]),
    "calm_reduce_scratch_floats": (_i64, [_i32, _i64, _i32]),
    "calm_layernorm_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _f32, _i32, _p]),
    "calm_layernorm_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    "calm_cast_chunk_elems": (_i32, []),
    "calm_cast_bf16": (_i32, [_p, _p, _i32, _p]),
    "calm_cast_bf16_one": (_i32, [_p, _p, _i64, _p]),
    "calm_cast_f32_one": (_i32, [_p, _p, _i64, _p]),
    "calm_quantize_fp8": (_i32, [_p, _i32, _i64, _p, _i32, _p, _p]),
    "calm_transpose_u8": (_i32, [_p, _p, _i32, _i32, _p]),
    "calm_rope_fwd": (_i32, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "calm_rope_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "calm_softmax_fwd": (_i32, [_p, _i64, _i32, _p]),
    "calm_softmax_bwd": (_i32, [_p, _p, _i64, _i32, _p]),
    "calm_softmax_bwd_heads": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "calm_sum_heads": (_i32, [_p, _p, _i32, _i32, _i64, _p]),
    "calm_attention_fwd_supported": (_i32, [_i32, _i32, _i32, _i32]),
    "calm_attention_fwd": (_i32, [_p] * 15 + [_i32] * 5 + [_p]),
    "calm_attention_bwd_preferred": (_i32, [_i32, _i32, _i32, _i32]),
    "calm_attention_bwd": (_i32, [_p] * 10 + [_i32] * 5 + [_p]),
    "calm_attention16_supported": (_i32, [_i32, _i32, _i32]),
    "calm_attention16_fwd": (_i32, [_p] * 16 + [_i32] * 4 + [_p]),
    "calm_attention16_bwd": (_i32, [_p] * 13 + [_i32] * 4 + [_p]),
    "calm_latent_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i32, _p, _p]),
    "calm_latent_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p]),
    "calm_sn_plan": (_i32, [C.POINTER(SnLayer), _i32, _p, C.POINTER(SnPlanInfo)]),
    "calm_sn_power_iter": (_i32, [_p, C.POINTER(SnPlanInfo), _i32, _f32, _p, _p]),
    "calm_sn_weight_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p, _p]),
    "calm_optim_chunk_elems": (_i32, []),
    "calm_optim_step": (_i32, [_p, _i32, _p, _i32, _p, C.POINTER(OptimHparams), _p, _p, _p, _p, _p]),
    "calm_collate_mix": (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _f32, _p, _p, _p, _p]),
    "calm_collate_crop_mix": (_i32, [_p, _i32, _i32, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f32, _p, _p, _p, _p]),
    "calm_image_to_rows": (_i32, [_p, _p, _i32, _i32, _p]),
    "calm_rows_to_image": (_i32, [_p, _p, _i32, _i32, _p]),
    "calm_grid_transpose": (_i32, [_p, _p, _i32, _i32, _p]),
    "calm_dwconv3x3_fwd": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p]),
    "calm_dwconv3x3_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "calm_cnn_residual_fwd": (_i32, [_p] * 11 + [_i32, _i32, _i32, _i32, _p]),
    "calm_cnn_residual_bwd": (_i32, [_p] * 18 + [_i32, _i32, _i32, _i32, _p, _p]),
    "calm_add": (_i32, [_p, _p, _p, _i64, _p]),
    "calm_gelu_fwd": (_i32, [_p, _p, _i64, _p]),
    "calm_gelu_bwd": (_i32, [_p, _p, _p, _i64, _p]),
    "calm_colsum": (_i32, [_p, _p, _i64, _i32, _i32, _p, _p]),
    "calm_row_scale": (_i32, [_p, _p, _p, _i32, _i32, _i32, _p]),
    "calm_mean_seq_fwd": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
    "calm_mean_seq_bwd": (_i32, [_p, _p, _i32, _i32, _i32, _p]),
}

_lib = None


def load():
    """dlopen the library once and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The CALM-ViT path has no non-HIP fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.calm_abi_version() != ABI_VERSION:
        raise RuntimeError("libcalmvit_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = "invalid argument/unsupported shape" if rc < 0 else "hipError_t"
        raise RuntimeError(f"{what} failed: code {rc} ({kind})")
