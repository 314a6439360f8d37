"""ctypes binding of libtad_mi355x.so (the C ABI declared in include/tad_mi355x.h).

The product path has no CPU or PyTorch fallback: if the shared library is missing
or a symbol cannot be resolved, loading raises and every op fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAD_LIB") or os.path.join(_HERE, "libtad_mi355x.so")  # TAD_LIB: an experiment build (simple_tad_amd/build.py)

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/tad_mi355x.h one-to-one
SIGNATURES = {
    "tad_abi_version": (_i, []),
    "tad_last_error_string": (C.c_char_p, []),
    "tad_cast_f32_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "tad_transpose_cast_f32_bf16": (_i, [_vp, _vp, _i, _i, _vp]),
    "tad_patch_embed_ldk": (_i, [_i, _i, _i]),
    "tad_im2col_tubelets": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tad_patch_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tad_patch_embed_fwd_implicit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tad_im2col_tubelets_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _i, _i, _vp]),
    "tad_patch_embed_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "tad_patch_embed_bwd_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tad_patch_embed_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i64, _i, _i, _vp]),
    "tad_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _f, _vp]),
    "tad_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "tad_layernorm_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _i64, _i, _vp]),
    "tad_linear_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i64, _i, _i, _vp]),
    "tad_linear_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tad_linear_tuning": (_i, [C.c_char_p, _i]),
    "tad_linear_tuning_get": (_i, [C.c_char_p, C.POINTER(_i)]),
    "tad_linear_fwd_qkv": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i64, _i, _i, _vp]),
    "tad_linear_bwd_weight_qkv": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _i64, _i, _i, _vp]),
    "tad_linear_bwd_weight_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _i64, _i, _vp]),
    "tad_linear_debug_stamps": (_i, [_vp]),
    "tad_linear_kernel_launches": (C.c_longlong, []),
    "tad_linear_bwd_input": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _sz, _i64, _i, _i, _vp]),
    "tad_linear_bwd_weight_workspace_bytes": (_sz, [_i64, _i, _i]),
    "tad_linear_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _sz, _i64, _i, _i, _vp]),
    "tad_attn_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, C.c_uint32, _vp]),
    "tad_attn_tuning": (_i, [C.c_char_p, _i]),
    "tad_attn_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "tad_attn_debug_stamps": (_i, [_vp]),
    "tad_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, C.c_uint32, _vp]),
    "tad_meanpool_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tad_meanpool_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tad_colsum_workspace_bytes": (_sz, [_i64, _i]),
    "tad_colsum_bf16": (_i, [_vp, _vp, _i, _vp, _sz, _i64, _i, _vp]),
    "tad_colsum_window_f32": (_i, [_vp, _vp, _i, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "tad_scale_cast_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp]),
    "tad_sumsq_workspace_bytes": (_sz, []),
    "tad_sumsq_f32": (_i, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "tad_grad_norm_coef": (_i, [_vp, _i64, _f, _f, _vp, _vp, _sz, _vp]),
    "tad_transpose_bf16_batched": (_i, [_vp, _vp, _vp, _i, _vp]),
    "tad_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_f), C.POINTER(_f), _i, C.POINTER(C.c_int32), _f, _f, _f, _vp, _vp,
                       _vp]),
    "tad_gather_rows_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "tad_scatter_rows_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "tad_mae_assemble": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tad_mae_target": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _i, _vp]),
    "tad_mse_loss_blocks": (_i, [_i64]),
    "tad_mse_loss": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "tad_threshold_histogram": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "tad_split_bf16x3": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    "tad_im2col_tubelets_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tad_attn_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "tad_attn_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, C.c_uint32, _vp]),
    "tad_gelu_f32": (_i, [_vp, _vp, _i64, _vp]),
    "tad_gelu_bwd_f32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "tad_colsum_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "tad_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "tad_rccl_unique_id": (_i, [_vp]),
    "tad_rccl_init": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "tad_rccl_world_size": (_i, [_vp, C.POINTER(_i)]),
    "tad_rccl_allreduce": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "tad_rccl_broadcast": (_i, [_vp, _vp, _sz, _i, _i, _vp]),
    "tad_rccl_destroy": (_i, [_vp]),
}

TAD_F32, TAD_BF16, TAD_F16 = 0, 1, 2

# IEEE-half twins (include/tad_mi355x.h, "IEEE half operand twins"): same signature as the bf16 entry point they mirror
F16_TWINS = {
    "tad_cast_f32_bf16": "tad_cast_f32_f16", "tad_transpose_cast_f32_bf16": "tad_transpose_cast_f32_f16",
    "tad_scale_cast_bf16": "tad_scale_cast_f16", "tad_colsum_bf16": "tad_colsum_f16", "tad_split_bf16x3": "tad_split_f16x3",
    "tad_im2col_tubelets": "tad_im2col_tubelets_f16", "tad_im2col_tubelets_u8": "tad_im2col_tubelets_u8_f16",
    "tad_patch_embed_fwd": "tad_patch_embed_fwd_f16", "tad_patch_embed_gemm": "tad_patch_embed_gemm_f16",
    "tad_patch_embed_fwd_implicit": "tad_patch_embed_fwd_implicit_f16",
    "tad_patch_embed_bwd": "tad_patch_embed_bwd_f16", "tad_layernorm_fwd": "tad_layernorm_fwd_f16",
    "tad_layernorm_bwd": "tad_layernorm_bwd_f16", "tad_linear_fwd": "tad_linear_fwd_f16", "tad_linear_fwd_qkv": "tad_linear_fwd_qkv_f16",
    "tad_linear_bwd_input": "tad_linear_bwd_input_f16", "tad_linear_bwd_weight": "tad_linear_bwd_weight_f16",
    "tad_linear_bwd_weight_qkv": "tad_linear_bwd_weight_qkv_f16", "tad_linear_bwd_weight_pair": "tad_linear_bwd_weight_pair_f16",
    "tad_attn_fwd": "tad_attn_fwd_f16", "tad_attn_bwd": "tad_attn_bwd_f16",
    "tad_meanpool_bwd": "tad_meanpool_bwd_f16", "tad_adamw_step": "tad_adamw_step_f16",
}
for _bf, _h in F16_TWINS.items():
    SIGNATURES[_h] = SIGNATURES[_bf]
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL = 0, 1, 2
ABI_VERSION = 4
ADAMW_CHUNK = 4096
ADAMW_MAX_GROUPS = 128
POOL_SPLIT = 8

_lib = None


class TadError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and return the shared library; raise if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TadError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run `python -m simple_tad_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback for the MI355X path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise TadError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.tad_abi_version()
    if v != ABI_VERSION:
        raise TadError(f"ABI mismatch: library reports {v}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().tad_last_error_string()
        raise TadError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
