/*
 * tad_mi355x.h -- C ABI of libtad_mi355x.so: the MI355X (gfx950) kernels behind
 * simple-tad's Video-ViT forward/backward path.
 *
 * The reference (tue-mps/simple-tad) has no FFI of its own: its hot path is the
 * Python nn.Module surface of modeling_finetune.py, whose arithmetic runs inside
 * third-party wheels (ATen / cuDNN / cuBLAS / flash-attn).  Each entry point below
 * names the reference call site whose native arithmetic it replaces.  The nearest
 * existing analogue of a C-level boundary in the reference is
 *   flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_s, dropout_p, softmax_scale, causal)
 * (flash_attention_class.py:47-50).
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + explicit sizes; no torch / C++ types.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).
 *   - return 0 on success, a negative TAD_E* code otherwise; never throw; the text of
 *     the last error on the calling thread is available from tad_last_error_string().
 *   - never allocate device memory, never synchronise the device, re-entrant per stream.
 *     Scratch memory is passed in by the caller; sizes come from the *_workspace_bytes()
 *     queries.
 *   - "bf16" = bfloat16 stored as uint16_t; "f32" = IEEE float.  Every uint16_t operand below is bf16 in the tad_* entry points and
 *     IEEE half in their tad_*_f16 twins (same kernels compiled for the other operand format; see the end of this header).
 *   - row-major everywhere; leading dimension == number of columns unless stated.
 */
#ifndef TAD_MI355X_H
#define TAD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAD_ABI_VERSION 4

enum tad_status {
  TAD_OK = 0,
  TAD_EINVAL = -1,   /* bad argument / unsupported shape */
  TAD_ELAUNCH = -2,  /* hipLaunch / runtime error         */
  TAD_ENOSPACE = -3  /* workspace too small               */
};

/* Element types.  TAD_BF16 / TAD_F16 name the 16-bit OPERAND FORMAT of a call: the entry points below take bfloat16 operands and accept
 * TAD_F32 or TAD_BF16 where an output type is selectable; their IEEE-half twins (tad_*_f16, end of this header) take half operands
 * and accept TAD_F32 or TAD_F16. */
enum tad_dtype { TAD_F32 = 0, TAD_BF16 = 1, TAD_F16 = 2 };

/* Linear-layer epilogues (tad_linear_fwd). */
enum tad_epilogue {
  TAD_EPI_BIAS = 0,          /* y = x W^T + b                                  (F.linear)            */
  TAD_EPI_BIAS_GELU = 1,     /* y = gelu_erf(x W^T + b); optional pre-activation copy (Mlp.fc1+act)  */
  TAD_EPI_BIAS_RESIDUAL = 2  /* y = res + rowscale[m/rows_per_scale] * gamma[n] * (x W^T + b)       */
};

typedef void* tad_stream_t;

int tad_abi_version(void);
const char* tad_last_error_string(void);

/* ---- weight preparation -------------------------------------------------------------
 * fp32 master weights stay owned by the caller (torch nn.Parameter); the kernels consume
 * bf16 copies: W [N,K] for forward, W^T [K,N] for the input-gradient GEMM. */
int tad_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, tad_stream_t stream);
int tad_transpose_cast_f32_bf16(const float* src /*[R,C]*/, uint16_t* dst /*[C,R]*/, int R, int C,
                                tad_stream_t stream);

/* ---- PatchEmbed: nn.Conv3d(3,D,k=s=(tub,p,p)) + flatten(2).transpose(1,2)
 *      (modeling_finetune.py:181-183,190) fused with "+ pos_embed" (:312-313).
 * x [B,C,T,H,W] f32 contiguous.  cols [B*N, C*tub*p*p] bf16 is the tubelet patch matrix
 * (token n = t'*H'*W' + h'*W' + w', k = ((c*tub+kt)*p+kh)*p+kw) -- kept for backward.
 * w_bf16 [D,K], bias [D] f32 (nullable), pos [N,D] f32 (nullable), out [B*N,D] f32. */
/* Row stride (elements) of the patch matrix and of the bf16 weight the patch-embed GEMM reads: K = C*tubelet*patch^2 rounded up to
 * the GEMM's K-tile of 64.  Equal to K for patch sizes that are multiples of 8 (/16: 1536); for even patch sizes that are not
 * (ViT-L/14: K = 1176 -> 1216) tad_im2col_tubelets writes rows of that stride with zeros in the padding columns and the caller
 * passes a zero-padded weight [D, ldk] -- the product is the un-padded one exactly, and /14 is a pure configuration change. */
int tad_patch_embed_ldk(int C, int tubelet, int patch);
int tad_im2col_tubelets(const float* x, uint16_t* cols, int B, int C, int T, int H, int W, int tubelet,
                        int patch, tad_stream_t stream);
int tad_patch_embed_fwd(const float* x, const uint16_t* w_bf16, const float* bias, const float* pos,
                        float* out, uint16_t* cols, int B, int C, int T, int H, int W, int tubelet,
                        int patch, int D, tad_stream_t stream);
/* The same forward as an IMPLICIT GEMM: the x operand is read straight from the f32 clip (register-staged, rounded to the
 * operand format as tad_im2col_tubelets rounds it), no patch matrix is written -- for forwards that keep nothing for a backward pass (eval,
 * no_grad, inference); the training step uses tad_patch_embed_fwd, whose patch matrix the weight gradient reads again.  Bit-identical to
 * tad_patch_embed_fwd.  patch must be 16 (TAD_EINVAL otherwise: callers fall back to the explicit form); clip and weight < 2 GiB each. */
int tad_patch_embed_fwd_implicit(const float* x, const uint16_t* w_bf16, const float* bias, const float* pos,
                                 float* out, int B, int C, int T, int H, int W, int tubelet, int patch, int D,
                                 tad_stream_t stream);
/* Input stage: the same patch matrix straight from uint8 frames [B,T,H,W,3] (decoder / cv2 layout) with the
 * reference's normalisation v = (u8/255 - mean[c]) / std[c] in f32 (run_inference.py:15-34 prepare_image; dota.py:443-460
 * tensor_normalize) -- bit-identical to tad_im2col_tubelets on the normalised f32 clip.  mean3 / std3: HOST arrays in RGB order;
 * bgr != 0: channel c is stored at position 2-c (cv2 frames, replaces cv2.cvtColor(BGR2RGB)); t_offset: frame t of the clip is
 * slot (t + t_offset) % T of the buffer (ring buffer for the sliding window of run_inference.py:86-93).  Any even patch size: rows have
 * the stride tad_patch_embed_ldk(3, tubelet, patch) (= K for multiples of 8; /14: 1216, padding columns zeroed). */
int tad_im2col_tubelets_u8(const uint8_t* frames, uint16_t* cols, int B, int T, int H, int W, int tubelet, int patch,
                           const float* mean3, const float* std3, int bgr, int t_offset, tad_stream_t stream);
/* GEMM part of tad_patch_embed_fwd on an existing patch matrix: out = cols * w^T + bias (+ pos broadcast over the batch) */
int tad_patch_embed_gemm(const uint16_t* cols, const uint16_t* w_bf16, const float* bias, const float* pos, float* out,
                         int64_t M, int ntok, int D, int K, tad_stream_t stream);
/* dW [D,K] f32 (overwritten), db [D] f32 (overwritten, nullable) from dy [B*N,D] bf16 and cols. */
size_t tad_patch_embed_bwd_workspace_bytes(int64_t M, int D, int K);
int tad_patch_embed_bwd(const uint16_t* dy_bf16, const uint16_t* cols, float* dW, float* db, void* ws,
                        size_t ws_bytes, int64_t M, int D, int K, tad_stream_t stream);

/* ---- nn.LayerNorm(D, eps) (modeling_finetune.py:143,149,270; eps=1e-6 at :342) -------
 * x [rows,D] f32; y [rows,D] in y_dtype; mean/rstd [rows] f32 saved for backward (nullable). */
int tad_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int64_t rows, int D, float eps, tad_stream_t stream);
/* dx = (dres ? dres : 0) + LN'(dy).  Optional outputs: dx_bf16 (copy of dx), colsum_dx [D]
 * (sum over rows of dx: the bias gradient of the Linear that produced x's residual branch).
 * rowscale (nullable, [ceil(rows / rows_per_scale)]): dx_bf16 and colsum_dx are taken of rowscale[row / rows_per_scale] * dx --
 * the gradient entering a residual branch whose output was scaled per sample (DropPath, modeling_finetune.py:23-34, 159-163).
 * accumulate != 0: dgamma, dbeta and colsum_dx are added to (gradient accumulation in place), else overwritten.
 * ws: tad_layernorm_bwd_workspace_bytes(rows, D). */
size_t tad_layernorm_bwd_workspace_bytes(int64_t rows, int D);
int tad_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                      const float* rstd, const float* dres, float* dx, uint16_t* dx_bf16, float* dgamma,
                      float* dbeta, float* colsum_dx, const float* rowscale, int rows_per_scale, int accumulate,
                      void* ws, size_t ws_bytes, int64_t rows, int D, tad_stream_t stream);

/* ---- Linear: F.linear(x, W, b) with fused epilogues ---------------------------------
 * replaces qkv (modeling_finetune.py:88-92), proj (:104), fc1+GELU (:48-49), fc2 (:52) and the
 * residual adds of Block.forward (:161-162, incl. gamma_1/2 and per-sample drop-path scale).
 * x [M,K] bf16, w [N,K] bf16, bias [N] f32 (nullable), y [M,N] y_dtype.
 * preact [M,N] bf16 (nullable): x W^T + b before GELU (saved for backward).
 * residual [M,N] f32, gamma [N] f32 (nullable), rowscale f32 [ceil(M/rows_per_scale)] (nullable). */
/* ws (nullable; tad_linear_workspace_bytes(M, N, K) bytes, private to the call's stream until the launches have run): scratch for the
 * split-K form of an under-filled last round of tiles -- a Linear whose 256 x 256 tiles do not fill whole rounds of one workgroup per
 * CU may run its last tiles as several shares of their K-tiles each, whose f32 partial tiles travel through this buffer.  Two plans:
 * the default (tad_linear_tuning("splitk_defer", 1)) is THREE launches -- partial tiles, the whole rounds, combine + epilogue -- ordered
 * by kernel boundaries, with no residency requirement: safe beside other kernels (an overlapped RCCL exchange, side streams).  The
 * in-launch combine ("splitk_defer", 0; experiments) waits for the other shares of a tile inside ONE launch and needs its whole grid
 * resident at once: if a share never arrives the kernel gives up, flags it in host-visible memory, and the NEXT tad_linear_* call on
 * the process returns TAD_ELAUNCH (the output of the affected Linear is invalid).  Results differ from the ws == NULL plan only in the
 * summation order over K.  tad_linear_workspace_bytes returns 0 for shapes whose plan never splits along K under the current
 * tad_linear_tuning knobs (every Linear of ViT-B by default; ViT-L's fc2 / dX(fc1) tails: 16 tiles x 8 shares = 33.5 MB). */
size_t tad_linear_workspace_bytes(int64_t M, int N, int K);
int tad_linear_fwd(const uint16_t* x, const uint16_t* w, const float* bias, void* y, int y_dtype,
                   int epilogue, uint16_t* preact, const float* residual, const float* gamma,
                   const float* rowscale, int rows_per_scale, void* ws, size_t ws_bytes, int64_t M, int N, int K,
                   tad_stream_t stream);
/* Input gradient: dx [M,K] = (dy [M,N] @ W)  using wT [K,N] bf16.
 * If gelu_preact [M,K] is given, dx *= gelu'(preact) (backward through the GELU that fed this Linear).
 * colscale [N] / rowscale are applied to dy on the fly is NOT supported; scale dy beforehand. */
int tad_linear_bwd_input(const uint16_t* dy, const uint16_t* wT, void* dx, int dx_dtype,
                         const uint16_t* gelu_preact, void* ws, size_t ws_bytes, int64_t M, int N, int K, tad_stream_t stream);
/* The qkv Linear of Attention (modeling_finetune.py:64-76, 89-92): bias = cat(q_bias, zeros, v_bias) without materialising it.
 * N = 3 * all_head_dim; q_bias / v_bias [N/3] f32 (both or neither).  Forward: y = x W^T + bias.  Weight gradient: as
 * tad_linear_bwd_weight, with the column sums of the first / last third of dy going to dq_bias / dv_bias (workspace: the same
 * tad_linear_bwd_weight_workspace_bytes). */
/* q_prescale (> 0; 1 = the plain Linear): the q third of y (columns [0, N/3)) is multiplied by it before the one rounding to
 * y_dtype.  With q_prescale = scale * log2(e) the attention kernels (q_prescaled != 0 below) get their scores from the matrix pipe
 * in log2 units and spend no vector instruction on the softmax scale (modeling_finetune.py:96: `q = q * self.scale`, done here). */
int tad_linear_fwd_qkv(const uint16_t* x, const uint16_t* w, const float* q_bias, const float* v_bias, void* y,
                       int y_dtype, float q_prescale, int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_bwd_weight_qkv(const uint16_t* dy, const uint16_t* x, float* dW, float* dq_bias, float* dv_bias,
                              int accumulate, void* ws, size_t ws_bytes, int64_t M, int N, int K,
                              tad_stream_t stream);
/* TWO weight gradients over the same M rows with the same K in one call -- the qkv and proj Linears of a Block (modeling_finetune.py:89-92,
 * 104): dW1 [N1, K] (+)= dy1^T x1 with its bias column sums (db1 [N1], or -- db1b != NULL -- the first / last third of them to db1 / db1b
 * [N1/3] as in tad_linear_bwd_weight_qkv; db1 NULL: none), dW2 [N2, K] (+)= dy2^T x2 (no bias sums).  When N1 is a multiple of 256 and K
 * runs on the 256-wide tiles both run as ONE kernel launch (a small problem alone pays for filling the chip with many reduction shares:
 * ViT-B's proj gradient is 9 tiles x 28 shares; the pair is 36 tiles x 7), otherwise as two; tad_linear_tuning("tn_pair", 0) forces two.
 * ws: at least the largest of tad_linear_bwd_weight_workspace_bytes(M, N1, K), (M, N2, K) and (M, N1 + N2, K).  The sums over the M rows
 * are taken in another order than the single calls take them (a different share count), deterministic for a given shape. */
int tad_linear_bwd_weight_pair(const uint16_t* dy1, const uint16_t* x1, float* dW1, float* db1, float* db1b, int N1,
                               const uint16_t* dy2, const uint16_t* x2, float* dW2, int N2, int accumulate, void* ws,
                               size_t ws_bytes, int64_t M, int K, tad_stream_t stream);
/* Scheduling knobs of the Linear GEMMs (process-wide; results never depend on them, only timing): tad_linear_tuning(key, value).
 *   "persistent"      1 = one workgroup per CU walks the tile list (default), 0 = one workgroup per tile
 *   "direct_epilogue" 2 = epilogue on the accumulator registers, stores straight from the MFMA layout; 0 = accumulators
 *                     transposed through the LDS first (whole rows per store instruction); 1 = per epilogue kind (default)
 *   "split_tail"      1 = a Linear whose 256 x 256 tiles do not fill whole rounds of one workgroup per CU may run as two
 *                     launches (whole rounds + remaining rows) when the cost model says so (default); 0 = never; 2 = always
 *   "splitk_tail"     1 = with a workspace, the second of those launches may split its tiles along K (default); 0 = never;
 *                     2 = whenever eligible (this one changes the summation order over K of the rows it covers)
 *   "splitk_defer"    1 = a split-K tail runs as three launches -- its partial tiles, the whole rounds, its combine + epilogue -- so
 *                     that the partial tiles travel through memory beside the whole rounds (default); 0 = one launch that combines inside
 *   "variant"         0 = tile configuration planned per shape (default); 1 / 3 / 2 / 4 / 5 = 256 x 256, 256 x 128, 128 x 128, 128 x 64,
 *                     64 x 64 tiles for every launch; 7 = the four-wave 256 x 256 kernels (128 x 128 outputs per wave, one wave per SIMD;
 *                     csrc/gemm_w4.hip); 8 / 9 = 192 x 128 as eight / four waves -- all bit-identical
 *   "short_k"         1 = Linears with K <= 512, whose epilogue is as long as their K loop, run on tiles that put two workgroups on a CU so
 *                     that one's epilogue runs beside the other's K loop: 192 x 128 as four waves (variant 9) for the GELU / GELU' Linears and,
 *                     at K < 512, the bias-only ones; 128 x 128 for the residual ones at K < 512 (default: the MAE decoder's fc1 / dX(fc2) -11 %,
 *                     ViT-S 2-12 % per shape); 0 = planned as the rest
 *   "tail_192"        1 = the second launch of the split plan may use 192 x 128 tiles where they put more CUs to work than 256 x 128
 *                     (default: ViT-B's 6656-row tails run 210 workgroups instead of 156, 9 % faster); 0 = 256 x 128 / 256 x 256 only
 *   "w4_plain"        K_min > 0: Linears whose reduction is at least K_min long run their whole rounds of 256 x 256 tiles on the
 *                     four-wave kernel (default 640: bit-identical, 5-8 % faster on the input-gradient Linears of ViT-B / L; at K = 384 / 512
 *                     its slower epilogue costs what its K loop gains); 0 = eight waves.  Applies to bias-only epilogues and to those of
 *   "w4_epilogues"    bit mask of the other epilogues whose whole rounds take the four-wave kernel: bit 1 GELU, 2 residual with f32 output
 *                     (default: 4), 3 GELU backward
 *   "tn_pair"         1 = tad_linear_bwd_weight_pair runs its two problems as one launch when they fit (default); 0 = always two launches
 *   "tn_w4"           1 = the 256 x 256 weight-gradient GEMM runs as four waves of 128 x 128 outputs (default; bit-identical, 8 % faster);
 *                     0 = eight waves of 128 x 64
 *   "tn_pdeep"        1 = the weight-gradient GEMM requests its dy operand two reduction tiles ahead (three-slot ring, the whole
 *                     160 KiB of LDS); 0 = two-stage ring (default: measured equal) */
int tad_linear_tuning(const char* key, int value);
/* Current value of a tad_linear_tuning knob (what a scoped change has to put back: simple_tad_amd.kernels.TuningScope). */
int tad_linear_tuning_get(const char* key, int* value);
/* Number of gemm_nt kernel launches issued so far by tad_linear_fwd* / tad_linear_bwd_input / tad_patch_embed_* (a call is one
 * launch, or two when the split-tail plan is taken): lets a profiler attribute event time to kernel launches. */
long long tad_linear_kernel_launches(void);
/* Debug timeline of the Linear GEMM kernels: while buf (device memory, >= gridDim * 64 * 32 * 8 bytes, caller-owned) is set,
 * every workgroup records s_memrealtime (100 MHz) for each of its first 64 tiles: slot 0 tile start, 1 K-loop end, 2 epilogue
 * issued, 3 stores acknowledged, 4 + 2q / 5 + 2q epilogue chunk q transposed / stored.  NULL switches it off (default).  Costs a store drain per tile: never leave it on.  Only libraries built with
 * -DTAD_GEMM_ABLATION (TAD_BUILD_ABLATION=1) carry the stamps; others return TAD_EINVAL for a non-NULL buffer. */
int tad_linear_debug_stamps(void* buf);
/* Weight gradient: dW [N,K] f32 = dy^T [N,M] @ x [M,K]; db [N] f32 = column sums of dy (nullable).
 * Outputs are overwritten (accumulate==0) or added to (accumulate!=0). */
size_t tad_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K);
int tad_linear_bwd_weight(const uint16_t* dy, const uint16_t* x, float* dW, float* db, int accumulate,
                          void* ws, size_t ws_bytes, int64_t M, int N, int K, tad_stream_t stream);

/* ---- Space-time attention: softmax(q k^T * scale) v, non-causal, no mask --------------
 * replaces Attention._naive_attn's q@k^T/softmax/attn@v (modeling_finetune.py:96-103) and
 * FlashAttention.forward -> flash_attn_varlen_qkvpacked_func (flash_attention_class.py:47-50)
 * with equal-length sequences (cu_seqlens = arange(0,(B+1)N,N)).
 * qkv [B,N,3,H,d] bf16 packed (the qkv Linear's output, column order [3][H][d]); d = 64 or 80 (the `d` argument).
 * out [B,N,H,d] in out_dtype; lse [B,H,N] f32 = log(sum_j exp(scale * q.k_j)) (natural log).
 * out_lo (nullable, 16-bit outputs only): [B,N,H,d] = what the rounding of out dropped (out + out_lo carries 16 / 22 significant
 * bits), for tad_attn_bwd's delta.
 * q_prescaled != 0: the q third of qkv already carries the factor scale * log2(e) (tad_linear_fwd_qkv's q_prescale); `scale` is
 * still the softmax scale.  0: plain q, as flash_attn_varlen_qkvpacked_func takes it (the kernels then scale their Q fragments
 * themselves: on the f32 scores).
 * dropout_p in [0, 1), seed: attention dropout (nn.Dropout on the softmax matrix, modeling_finetune.py:99-101; dropout_p of
 * flash_attn_varlen_qkvpacked_func, flash_attention_class.py:56-61): element (b, h, query, key) is kept iff
 * hash((b H + h) N + query, key, seed) >= dropout_p * 2^32 (csrc/common.h: drop_keep; oracle/vit_oracle.py:
 * attention_dropout_keep regenerates the mask) and kept probabilities are scaled by 1 / (1 - dropout_p); lse is that of the full
 * softmax.  tad_attn_bwd must be given the same dropout_p and seed.  0 = no dropout (evaluation). */
int tad_attn_fwd(const uint16_t* qkv, void* out, int out_dtype, uint16_t* out_lo, float* lse, int B, int N, int H, int d,
                 float scale, int q_prescaled, float dropout_p, uint32_t seed, tad_stream_t stream);
/* dqkv [B,N,3,H,d] bf16 (fully overwritten).  delta: scratch of tad_attn_bwd_scratch_bytes(B, N, H) bytes = 2*B*H*N floats (the
 * first kernel leaves -rowsum(dout*out) in [0, BHN) and -lse/scale (-lse*log2(e) with q_prescaled) in [BHN, 2 BHN) for the second
 * one, which takes them as the initial values of its accumulators).  The q slot of dqkv is the gradient of the PLAIN q in either case. */
/* Knob of the three attention kernels.  "dma_mode": 0 = production; 2 / 3 = timing-only ablations (wrong results) that only
 * ablation builds (TAD_BUILD_ABLATION=1) accept. */
int tad_attn_tuning(const char* key, int value);
size_t tad_attn_bwd_scratch_bytes(int B, int N, int H);
/* Diagnostic, ablation builds only (see tad_linear_debug_stamps): while buf (device memory, 32 bytes per workgroup of the dK/dV grid
 * = ceil(N/128)*H*B workgroups) is set, every dK/dV workgroup records {s_memrealtime, s_memtime} at the start and at the end of its
 * tile loop: (d memtime / d memrealtime) x 100 MHz = the clock held inside the loop.  NULL switches it off. */
int tad_attn_debug_stamps(void* buf);
/* out_lo (nullable): the forward's rounding residual; with it delta = rowsum(dout * (out + out_lo)), i.e. of the unrounded output, which
 * is what cancels against the dP the kernels recompute (without it the q / k gradients of near-uniform attention rows carry the
 * rounding of out amplified by |delta| / |dP - delta|). */
int tad_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* out_lo, const uint16_t* dout, const float* lse,
                 uint16_t* dqkv, float* delta, int B, int N, int H, int d, float scale, int q_prescaled, float dropout_p,
                 uint32_t seed, tad_stream_t stream);

/* ---- token mean-pool x.mean(1) (modeling_finetune.py:325-326) -------------------------
 * x [B,N,D] f32 -> y [B,D] f32.  ws: B*TAD_POOL_SPLIT*D floats. */
#define TAD_POOL_SPLIT 8
int tad_meanpool_fwd(const float* x, float* y, float* ws, int B, int N, int D, tad_stream_t stream);
/* dx[b,n,:] = dy[b,:]/N  (f32, plus optional bf16 copy) */
int tad_meanpool_bwd(const float* dy, float* dx, uint16_t* dx_bf16, int B, int N, int D,
                     tad_stream_t stream);

/* ---- small helpers used by the training step ------------------------------------------
 * column sums of a bf16 [M,N] matrix into f32 [N] (bias gradients: q_bias/v_bias/fc1.bias). */
size_t tad_colsum_workspace_bytes(int64_t M, int N);
/* Column sums over a row window of an f32 [B, R, N] tensor: out[n] (+)= sum_b sum_{r0 <= r < r0 + rc} a[b, r, n] (workspace:
 * tad_colsum_workspace_bytes(B * rc, N)).  The mask-token gradient of the MAE decoder input. */
int tad_colsum_window_f32(const float* a, float* out, int accumulate, void* ws, size_t ws_bytes, int B, int R, int N,
                          int r0, int rc, tad_stream_t stream);
int tad_colsum_bf16(const uint16_t* a, float* out, int accumulate, void* ws, size_t ws_bytes, int64_t M,
                    int N, tad_stream_t stream);
/* y_bf16 = bf16(rowscale[m/rows_per_scale] * gamma[n] * x_f32)  (backward of the residual-branch scale) */
int tad_scale_cast_bf16(const float* x, uint16_t* y, const float* gamma, const float* rowscale,
                        int rows_per_scale, int64_t M, int N, tad_stream_t stream);
/* sum of squares of an f32 vector, accumulated into *out (f32, device) -- get_grad_norm_ (utils.py:415-427).  Deterministic: block
 * partials in ws (tad_sumsq_workspace_bytes()), added in a fixed order; no atomics. */
size_t tad_sumsq_workspace_bytes(void);
int tad_sumsq_f32(const float* x, int64_t n, float* out, void* ws, size_t ws_bytes, tad_stream_t stream);
/* NativeScalerWithGradNormCount's unscale + clip + overflow decision (utils.py:386-412: GradScaler.unscale_, clip_grad_norm_, the
 * "found inf" skip) as two launches on the flat gradient buffer, nothing read back by the host:
 *   out3[0] = ||x||_2 * inv_scale                                  (the gradient norm with the loss scale removed)
 *   out3[1] = inv_scale * min(1, max_norm / (out3[0] + 1e-6))       (max_norm <= 0: inv_scale) -- tad_adamw_step's grad_scale operand;
 *             0 when the norm is inf / NaN: the optimizer kernel then skips the step
 *   out3[2] = 1 when the norm is inf / NaN, else 0                  (read by the host one step late)
 * ws: tad_sumsq_workspace_bytes().  Deterministic (fixed order, no atomics). */
int tad_grad_norm_coef(const float* x, int64_t n, float inv_scale, float max_norm, float* out3, void* ws, size_t ws_bytes,
                       tad_stream_t stream);

/* Batched bf16 transpose: every [R,C] matrix of a flat buffer -> [C,R] at the same offset of a second flat buffer, one launch (the
 * transposed operand copies W^T that the input-gradient GEMMs of all Linear layers read; refreshed once per optimizer step).
 * table (device, int32 [n_tiles][8]): per 64x64 tile {src offset, dst offset, C, R, valid rows, valid cols, 0, 0} in elements,
 * offsets < 2^32; R % 8 == C % 8 == 0 and 16-byte aligned matrices. */
int tad_transpose_bf16_batched(const uint16_t* src, uint16_t* dst, const int32_t* table, int n_tiles, tad_stream_t stream);

/* ---- optimizer tail -------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW over FLAT buffers: replaces torch.optim.AdamW as configured by optim_factory.create_optimizer
 * (optim_factory.py:91-127) with the layer-decay parameter groups of optim_factory.get_parameter_groups (:49-88) and the per-step
 * lr / weight-decay assignment of engine_for_finetuning.train_one_epoch (:49-54); the sum of g^2 that utils.get_grad_norm_
 * (utils.py:415-427) needs comes out of the same pass.
 * param / grad / exp_avg / exp_avg_sq: f32 [n], one shared layout in which every tensor starts on a TAD_ADAMW_CHUNK boundary.
 * chunk_group[c] (device, uint8, ceil(n / CHUNK) entries) = parameter-group index of chunk c, or 255 = leave the chunk untouched.
 * group_lr / group_wd / group_step: HOST arrays [n_groups] (this step's lr and weight decay; the 1-based count of updates the
 * group's tensors will have received after this call -- torch keeps it per parameter).  Update rule = torch.optim.AdamW:
 *   p *= 1 - lr*wd;  m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * grad_scale (device scalar or NULL): g is multiplied by *grad_scale first (gradient clipping and loss-scale removal without a host
 *   sync).  *grad_scale == 0 or not finite SKIPS the update (parameters, moments and the operand copy stay untouched; sumsq_partials
 *   is still written): the device-side form of GradScaler's "found inf -> skip the step" (utils.py:386-412).
 * param_bf16 (or NULL): receives bf16(p_new) -- the operand copy the next forward's GEMMs read.
 * sumsq_partials (or NULL): [ceil(n / CHUNK)] per-chunk sums of the UNSCALED g^2 (deterministic order). */
#define TAD_ADAMW_CHUNK 4096
#define TAD_ADAMW_MAX_GROUPS 128
int tad_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint16_t* param_bf16,
                   const uint8_t* chunk_group, int64_t n, const float* group_lr, const float* group_wd, int n_groups,
                   const int32_t* group_step, float beta1, float beta2, float eps, const float* grad_scale,
                   float* sumsq_partials, tad_stream_t stream);

/* ---- MAE pre-training path: what modeling_pretrain.py / engine_for_pretraining.py add around the Block stack ----
 * Rows are D f32, D % 4 == 0.  idx arrays are int32 on the device. */
/* out[r] = src[idx[r]], r < n_out: x[~mask].reshape(B,-1,C) (modeling_pretrain.py:98) with idx = b*N + visible token */
int tad_gather_rows_f32(const float* src, const int32_t* idx, float* out, int64_t n_out, int D, tad_stream_t stream);
/* out[idx[r]] = src[r], r < n_in (unique indices; the caller zero-fills out): backward of the gather */
int tad_scatter_rows_f32(const float* src, const int32_t* idx, float* out, int64_t n_in, int D, tad_stream_t stream);
/* decoder input (modeling_pretrain.py:283-288): out [B, n_vis+n_mask, D] = cat(x_vis + pos[vis_idx], mask_token + pos[mask_idx]);
 * x_vis [B*n_vis, D], mask_token [D], pos [N, D], vis_idx [B*n_vis] / mask_idx [B*n_mask] = token index inside the clip */
int tad_mae_assemble(const float* x_vis, const float* mask_token, const float* pos, const int32_t* vis_idx, const int32_t* mask_idx,
                     float* out, int B, int n_vis, int n_mask, int D, tad_stream_t stream);
/* reconstruction target (engine_for_pretraining.py:51-66): videos [B,3,T,H,W] f32 (normalised clip) -> labels [B*n_mask,
 * tub*p*p*3] for the masked tokens: v = x*std + mean, patch layout 'b n (p0 p1 p2) c', and with normalize_target each
 * (patch, channel) is standardised over its pixels: (v - mean) / (sqrt(unbiased var) + 1e-6).  mean3 / std3: HOST arrays. */
int tad_mae_target(const float* videos, const int32_t* mask_idx, float* labels, int B, int n_mask, int T, int H, int W, int tubelet,
                   int patch, const float* mean3, const float* std3, int normalize_target, tad_stream_t stream);
/* nn.MSELoss() (engine_for_pretraining.py:27,70): partials[tad_mse_loss_blocks(n)] = block sums of (pred-target)^2 (the loss is
 * their sum / n); grad (nullable) = 2 (pred - target) / n */
int tad_mse_loss_blocks(int64_t n);
int tad_mse_loss(const float* pred, const float* target, int64_t n, float* partials, float* grad, tad_stream_t stream);

/* ---- evaluation path -----------------------------------------------------------------------------------------
 * Exact integer counts behind every thresholded metric of engine_for_frame_finetuning.calculate_metrics (:593-636) and
 * anaysis/metrics.calculate_MORE_metrics (:127-208), which test `pred >= t` for t in np.arange(0, 1.001, 0.01).
 * thresholds: ascending f32 [n_thresholds <= 255] (device); labels int32 (non-zero = positive);
 * hist int64 [2][n_thresholds+1] (device, overwritten): hist[l][k] = number of samples with label l and k = #{t : p >= t}.
 * The confusion matrix at threshold index i is then  TP = sum_{k>i} hist[1][k], FN = sum_{k<=i} hist[1][k], FP / TN likewise. */
int tad_threshold_histogram(const float* probs, const int32_t* labels, const float* thresholds, int n_thresholds, int64_t n,
                            int64_t* hist, tad_stream_t stream);

/* ---- "precise" mode (parity gate, not throughput): f32-accurate Linear via split-bf16 operands, f32 attention ----------
 * x = hi + lo (bf16 each).  concat mode: out [M,3K] = [hi|hi|lo] (role_b=0) or [hi|lo|hi] (role_b=1); stack mode: out [3M,K]
 * with the three parts stacked along rows.  Feeding tad_linear_* with both operands split this way (K or M tripled) gives the
 * f32 product to ~2^-17 using the same MFMA kernels. */
int tad_split_bf16x3(const float* x, uint16_t* out, int64_t M, int K, int role_b, int stack, tad_stream_t stream);
int tad_im2col_tubelets_f32(const float* x, float* cols, int B, int C, int T, int H, int W, int tubelet, int patch,
                            tad_stream_t stream);
/* f32 attention on the matrix pipe (exact-f32 MFMA), head dim d = 64 or 80: qkv [B,N,3,H,d] f32 -> out [B,N,H,d] f32, lse [B,H,N]
 * (nullable).  dropout_p in [0, 1): nn.Dropout on the softmax matrix (modeling_finetune.py:99-101; flash_attention_class.py:59-61) --
 * element (b, h, query, key) is kept iff lowbias32(row * 0x9E3779B1 + key * 0x85EBCA77 + seed) >= dropout_p * 2^32 with
 * row = (b H + h) N + query, kept probabilities scaled by 1 / (1 - dropout_p); the backward regenerates the same bits from (p, seed). */
int tad_attn_fwd_f32(const float* qkv, float* out, float* lse, int B, int N, int H, int d, float scale, float dropout_p, uint32_t seed,
                     tad_stream_t stream);

/* precise-mode backward pieces: f32 attention backward (dqkv [B,N,3,H,64] f32 fully overwritten; delta [B,H,N] scratch),
 * erf-GELU forward / backward on f32, column sums of an f32 matrix (bias gradients). */
int tad_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta,
                     int B, int N, int H, int d, float scale, float dropout_p, uint32_t seed, tad_stream_t stream);
int tad_gelu_f32(const float* h, float* a, int64_t n, tad_stream_t stream);
int tad_gelu_bwd_f32(const float* dy, const float* h, float* dh, int64_t n, tad_stream_t stream);
int tad_colsum_f32(const float* a, float* out, int64_t M, int N, tad_stream_t stream);

/* ---- device info ------------------------------------------------------------------------ */
int tad_device_info(int* cu_count, int* clock_khz, int* lds_bytes_per_cu, char* name, int name_len);

/* ---- gradient exchange over RCCL / xGMI for C hosts --------------------------------------------------------------------------
 * Replaces, for a host without torch, the two collectives of the reference's data-parallel wrap: DistributedDataParallel's
 * gradient all-reduce (run_class_finetuning.py:446-448; process group from utils.init_distributed_mode, utils.py:283-333,
 * backend 'nccl' at :325) and its initial parameter broadcast.  The Python host keeps torch.distributed (whose "nccl" backend is
 * RCCL on ROCm): parallel.DataParallel.  One communicator per process, one process per GPU; rank 0 creates the 128-byte id with
 * tad_rccl_unique_id and hands it to the other ranks by any host channel (file, environment, socket); every rank then calls
 * tad_rccl_init with the device it will use already current (hipSetDevice).  All-reduce and broadcast are in place and ordered on
 * `stream`; `average` != 0 divides by the number of ranks (ncclAvg).  librccl.so.1 is opened on first use (no link-time
 * dependency): on a host without it these calls return TAD_ELAUNCH and the rest of the library is unaffected. */
#define TAD_RCCL_UNIQUE_ID_BYTES 128
typedef void* tad_comm_t;
int tad_rccl_unique_id(void* id128);
int tad_rccl_init(const void* id128, int nranks, int rank, tad_comm_t* comm);
int tad_rccl_world_size(tad_comm_t comm, int* nranks);
int tad_rccl_allreduce(tad_comm_t comm, void* buf, size_t count, int dtype, int average, tad_stream_t stream);
int tad_rccl_broadcast(tad_comm_t comm, void* buf, size_t count, int dtype, int root, tad_stream_t stream);
int tad_rccl_destroy(tad_comm_t comm);

/* ---- IEEE half operand twins ------------------------------------------------------------------------------------------------------
 * The reference trains under torch.cuda.amp.autocast() -- float16 on CUDA -- with a GradScaler (engine_for_finetuning.py:67,
 * utils.py:386-412).  Every entry point above that takes or produces 16-bit GEMM / attention operands therefore exists a second
 * time with IEEE half in place of bfloat16: identical signature, semantics, kernels and schedules (the sources are compiled twice,
 * csrc/common.h), TAD_F16 in place of TAD_BF16 wherever a dtype argument selects the 16-bit type.  Half carries 11 significant bits
 * against bfloat16's 8 (operand rounding 2^-12 instead of 2^-9) at the same MFMA rate, and a narrower range (6e-8 .. 65504): the
 * backward pass runs on loss-scaled gradients exactly as in the reference (engine.NativeScalerWithGradNormCount; the scale is removed,
 * and an overflowed step skipped, inside tad_adamw_step_f16 through its grad_scale argument). */
int tad_cast_f32_f16(const float* src, uint16_t* dst, int64_t n, tad_stream_t stream);
int tad_transpose_cast_f32_f16(const float* src /*[R,C]*/, uint16_t* dst /*[C,R]*/, int R, int C, tad_stream_t stream);
int tad_scale_cast_f16(const float* x, uint16_t* y, const float* gamma, const float* rowscale, int rows_per_scale, int64_t M, int N,
                       tad_stream_t stream);
int tad_colsum_f16(const uint16_t* a, float* out, int accumulate, void* ws, size_t ws_bytes, int64_t M, int N, tad_stream_t stream);
int tad_split_f16x3(const float* x, uint16_t* out, int64_t M, int K, int role_b, int stack, tad_stream_t stream);
int tad_im2col_tubelets_f16(const float* x, uint16_t* cols, int B, int C, int T, int H, int W, int tubelet, int patch,
                            tad_stream_t stream);
int tad_im2col_tubelets_u8_f16(const uint8_t* frames, uint16_t* cols, int B, int T, int H, int W, int tubelet, int patch,
                               const float* mean3, const float* std3, int bgr, int t_offset, tad_stream_t stream);
int tad_patch_embed_fwd_f16(const float* x, const uint16_t* w_f16, const float* bias, const float* pos, float* out, uint16_t* cols,
                            int B, int C, int T, int H, int W, int tubelet, int patch, int D, tad_stream_t stream);
int tad_patch_embed_fwd_implicit_f16(const float* x, const uint16_t* w_f16, const float* bias, const float* pos, float* out, int B, int C,
                                     int T, int H, int W, int tubelet, int patch, int D, tad_stream_t stream);
int tad_patch_embed_gemm_f16(const uint16_t* cols, const uint16_t* w_f16, const float* bias, const float* pos, float* out, int64_t M,
                             int ntok, int D, int K, tad_stream_t stream);
int tad_patch_embed_bwd_f16(const uint16_t* dy_f16, const uint16_t* cols, float* dW, float* db, void* ws, size_t ws_bytes, int64_t M,
                            int D, int K, tad_stream_t stream);
int tad_layernorm_fwd_f16(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, float* mean, float* rstd,
                          int64_t rows, int D, float eps, tad_stream_t stream);
int tad_layernorm_bwd_f16(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, uint16_t* dx_f16, float* dgamma, float* dbeta, float* colsum_dx,
                          const float* rowscale, int rows_per_scale, int accumulate, void* ws, size_t ws_bytes, int64_t rows, int D,
                          tad_stream_t stream);
int tad_linear_fwd_f16(const uint16_t* x, const uint16_t* w, const float* bias, void* y, int y_dtype, int epilogue, uint16_t* preact,
                       const float* residual, const float* gamma, const float* rowscale, int rows_per_scale, void* ws, size_t ws_bytes,
                       int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_fwd_qkv_f16(const uint16_t* x, const uint16_t* w, const float* q_bias, const float* v_bias, void* y, int y_dtype,
                           float q_prescale, int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_bwd_input_f16(const uint16_t* dy, const uint16_t* wT, void* dx, int dx_dtype, const uint16_t* gelu_preact, void* ws,
                             size_t ws_bytes, int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_bwd_weight_f16(const uint16_t* dy, const uint16_t* x, float* dW, float* db, int accumulate, void* ws, size_t ws_bytes,
                              int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_bwd_weight_qkv_f16(const uint16_t* dy, const uint16_t* x, float* dW, float* dq_bias, float* dv_bias, int accumulate,
                                  void* ws, size_t ws_bytes, int64_t M, int N, int K, tad_stream_t stream);
int tad_linear_bwd_weight_pair_f16(const uint16_t* dy1, const uint16_t* x1, float* dW1, float* db1, float* db1b, int N1,
                                   const uint16_t* dy2, const uint16_t* x2, float* dW2, int N2, int accumulate, void* ws,
                                   size_t ws_bytes, int64_t M, int K, tad_stream_t stream);
int tad_attn_fwd_f16(const uint16_t* qkv, void* out, int out_dtype, uint16_t* out_lo, float* lse, int B, int N, int H, int d,
                     float scale, int q_prescaled, float dropout_p, uint32_t seed, tad_stream_t stream);
int tad_attn_bwd_f16(const uint16_t* qkv, const uint16_t* out, const uint16_t* out_lo, const uint16_t* dout, const float* lse,
                     uint16_t* dqkv, float* delta, int B, int N, int H, int d, float scale, int q_prescaled, float dropout_p,
                     uint32_t seed, tad_stream_t stream);
int tad_meanpool_bwd_f16(const float* dy, float* dx, uint16_t* dx_f16, int B, int N, int D, tad_stream_t stream);
int tad_adamw_step_f16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint16_t* param_f16,
                       const uint8_t* chunk_group, int64_t n, const float* group_lr, const float* group_wd, int n_groups,
                       const int32_t* group_step, float beta1, float beta2, float eps, const float* grad_scale,
                       float* sumsq_partials, tad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TAD_MI355X_H */
