// Names of the exported entry points in the IEEE-half compilation pass (-DTAD_OPND_F16): every entry point that takes or produces
// 16-bit GEMM / attention operands exists twice in libtad_mi355x.so, tad_<name> for bfloat16 and tad_<name>_f16 for half
// (include/tad_mi355x.h, section "IEEE half operand twins").  The sources are written once under the bf16 names; this header, included
// before anything else in the half pass, renames definitions, declarations and internal cross-calls alike.
#pragma once
#define tad_cast_f32_bf16 tad_cast_f32_f16
#define tad_transpose_cast_f32_bf16 tad_transpose_cast_f32_f16
#define tad_scale_cast_bf16 tad_scale_cast_f16
#define tad_colsum_bf16 tad_colsum_f16
#define tad_split_bf16x3 tad_split_f16x3
#define tad_im2col_tubelets tad_im2col_tubelets_f16
#define tad_im2col_tubelets_u8 tad_im2col_tubelets_u8_f16
#define tad_patch_embed_fwd tad_patch_embed_fwd_f16
#define tad_patch_embed_gemm tad_patch_embed_gemm_f16
#define tad_patch_embed_fwd_implicit tad_patch_embed_fwd_implicit_f16
#define tad_patch_embed_bwd tad_patch_embed_bwd_f16
#define tad_layernorm_fwd tad_layernorm_fwd_f16
#define tad_layernorm_bwd tad_layernorm_bwd_f16
#define tad_linear_fwd tad_linear_fwd_f16
#define tad_linear_fwd_qkv tad_linear_fwd_qkv_f16
#define tad_linear_bwd_input tad_linear_bwd_input_f16
#define tad_linear_bwd_weight tad_linear_bwd_weight_f16
#define tad_linear_bwd_weight_qkv tad_linear_bwd_weight_qkv_f16
#define tad_linear_bwd_weight_pair tad_linear_bwd_weight_pair_f16
#define tad_attn_fwd tad_attn_fwd_f16
#define tad_attn_bwd tad_attn_bwd_f16
#define tad_meanpool_bwd tad_meanpool_bwd_f16
#define tad_adamw_step tad_adamw_step_f16
