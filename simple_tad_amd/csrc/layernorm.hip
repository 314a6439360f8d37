// LayerNorm forward / backward for rows of D f32 elements (D % 4 == 0, D <= 2048).
// HBM-bound: one wave per row, the row lives in registers (float4 per lane per 256 columns),
// two-pass mean / variance in registers (same numerics as torch: biased variance around the mean),
// wave reductions by cross-lane shuffles -- no LDS on the forward path.
#include "common.h"

TAD_NAMESPACE_BEGIN

constexpr int LN_MAX_V = 8;  // float4 per lane -> D <= 64*4*8 = 2048
#ifndef TAD_LN_NT
#define TAD_LN_NT 0  // (experiment) cache policy of LayerNorm's once-touched streams, a bit mask: 1 = the backward's loads of x, dy and the
                     // incoming residual gradient (each read for the last time here) non-temporal; 2 = the forward's load of x; 4 = the backward's
                     // store of the residual gradient (next read a whole block later); 8 = the forward's store.  Round 2 measured the single bits null on
                     // the round-2 step; re-measured at step level in round 6 (experiments/README.md r06)
#endif
constexpr bool LN_NT_BWD_LD = (TAD_LN_NT & 1) != 0, LN_NT_FWD_LD = (TAD_LN_NT & 2) != 0, LN_NT_BWD_ST = (TAD_LN_NT & 4) != 0, LN_NT_FWD_ST = (TAD_LN_NT & 8) != 0;

int launch_reduce_partials(const float* partial, float* out, int splits, int64_t n, int accumulate, hipStream_t st);
int launch_reduce_cols(const float* partial, float* out0, float* out1, float* out2, int nq, int splits, int n, int accumulate,
                       hipStream_t st);

template <int NV, bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int D4 = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * D);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < D4) ? ldg_f4<LN_NT_FWD_LD>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c];
      const float4 b = reinterpret_cast<const float4*>(beta)[c];
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      if (OUT_BF16) {
        uint2 p;
        p.x = pack_op16x2(o.x, o.y);
        p.y = pack_op16x2(o.z, o.w);
        stg_u2<LN_NT_FWD_ST>(reinterpret_cast<uint2*>((uint16_t*)y + row * D) + c, p);
      } else {
        stg_f4<LN_NT_FWD_ST>(reinterpret_cast<float4*>((float*)y + row * D) + c, o);
      }
    }
  }
}

// Backward.  Each block owns a contiguous chunk of rows; each wave walks rows of the chunk with stride 4 and keeps
// per-lane column partials of dgamma, dbeta and (optionally) colsum(dx) in registers; they are combined across the
// block's 4 waves in LDS and written as one partial row per block -> reduced by reduce_partials_kernel.
// FULL: D == 256 * NV (768, 1024, 1280, 512 ...): no per-lane column guard, so the row loop is one basic block up to the optional bf16
// store (the guards split every chunk into its own exec-masked region with its own waits).
__device__ const float ln_one = 1.0f;
template <int NV, bool DY_BF16, bool FULL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dx, uint16_t* __restrict__ dxb,
                                                            float* __restrict__ partial /*[grid][3][D]*/, int64_t rows, int D,
                                                            int rows_per_block, int want_colsum,
                                                            const float* __restrict__ rowscale, int rows_per_scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][3][D]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D4 = D >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float4 g[NV], dg[NV], db[NV], cs[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    g[i] = (c < D4) ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = db[i] = cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // One row per wave and iteration.  Everything the row needs (x, dy, the residual gradient, its statistics and its drop-path scale) is
  // requested in ONE batch at the top -- the first version waited for each 256-column chunk on its own and fetched `dres` only behind
  // the two reductions, six dependent HBM round trips per row, and its `s_waitcnt vmcnt(0)` in front of every chunk's store also drained
  // the stores issued before it (CDNA4 counts stores in vmcnt) -- and the two row sums go through DPP / permlane steps instead of
  // twelve ds_bpermute round trips (wave_sum_dpp, common.h).  Optional operands are made unconditional so that the loads stay in one
  // block: without a residual gradient the row of x is read a second time and multiplied by 0, without row scales the scale comes from a
  // constant.
  const float* const rsrc = dres ? dres : x;
  const float rmul = dres ? 1.f : 0.f;
  const float* const scp = rowscale ? rowscale : &ln_one;
  // index of the row's drop-path scale, kept incrementally (one division per wave instead of one 64-bit division per row and chunk)
  int sidx = 0, srem = 0;
  if (rowscale) {
    sidx = (int)((r0 + wave) / rows_per_scale);
    srem = (int)((r0 + wave) - (int64_t)sidx * rows_per_scale);
  }
  for (int64_t row = r0 + wave; row < r1; row += 4) {
    float4 xv[NV], dyv[NV], rv[NV];
    uint2 dyp[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < D4) {
        xv[i] = ldg_f4<LN_NT_BWD_LD>(reinterpret_cast<const float4*>(x + row * D) + c);
        if (DY_BF16) dyp[i] = ldg_u2<LN_NT_BWD_LD>(reinterpret_cast<const uint2*>((const uint16_t*)dy + row * D) + c);
        else dyv[i] = ldg_f4<LN_NT_BWD_LD>(reinterpret_cast<const float4*>((const float*)dy + row * D) + c);
        rv[i] = ldg_f4<LN_NT_BWD_LD>(reinterpret_cast<const float4*>(rsrc + row * D) + c);
      }
    }
    const float mu = mean[row], rs = rstd[row];
    const float sc = scp[sidx];
    if (rowscale) {
      srem += 4;
      while (srem >= rows_per_scale) { srem -= rows_per_scale; ++sidx; }
    }
    // keep the whole batch of loads above the arithmetic: the scheduler otherwise sinks them chunk by chunk (one exposed round trip each)
    __builtin_amdgcn_sched_barrier(0);
    float4 xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < D4) {
        if (DY_BF16)
          dyv[i] = make_float4(op16_lo_f32(dyp[i].x), op16_hi_f32(dyp[i].x), op16_lo_f32(dyp[i].y), op16_hi_f32(dyp[i].y));
        xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
        const float4 t = make_float4(dyv[i].x * g[i].x, dyv[i].y * g[i].y, dyv[i].z * g[i].z, dyv[i].w * g[i].w);
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += (t.x * xh[i].x + t.y * xh[i].y) + (t.z * xh[i].z + t.w * xh[i].w);
        dg[i].x += dyv[i].x * xh[i].x; dg[i].y += dyv[i].y * xh[i].y; dg[i].z += dyv[i].z * xh[i].z; dg[i].w += dyv[i].w * xh[i].w;
        db[i].x += dyv[i].x; db[i].y += dyv[i].y; db[i].z += dyv[i].z; db[i].w += dyv[i].w;
      } else {
        xh[i] = dyv[i] = rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float c1 = wave_sum_dpp(s1) / (float)D;
    const float c2 = wave_sum_dpp(s2) / (float)D;
    float4 o[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      o[i].x = fmaf(rv[i].x, rmul, rs * (dyv[i].x * g[i].x - c1 - xh[i].x * c2));
      o[i].y = fmaf(rv[i].y, rmul, rs * (dyv[i].y * g[i].y - c1 - xh[i].y * c2));
      o[i].z = fmaf(rv[i].z, rmul, rs * (dyv[i].z * g[i].z - c1 - xh[i].z * c2));
      o[i].w = fmaf(rv[i].w, rmul, rs * (dyv[i].w * g[i].w - c1 - xh[i].w * c2));
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < D4) stg_f4<LN_NT_BWD_ST>(reinterpret_cast<float4*>(dx + row * D) + c, o[i]);
      // the bf16 copy and the column sums feed the branch Linear, whose output was scaled per sample (drop-path)
      o[i].x *= sc; o[i].y *= sc; o[i].z *= sc; o[i].w *= sc;
      if (FULL || c < D4) { cs[i].x += o[i].x; cs[i].y += o[i].y; cs[i].z += o[i].z; cs[i].w += o[i].w; }
    }
    if (dxb) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (FULL || c < D4) {
          uint2 p;
          p.x = pack_op16x2(o[i].x, o[i].y);
          p.y = pack_op16x2(o[i].z, o[i].w);
          stg_u2<false>(reinterpret_cast<uint2*>(dxb + row * D) + c, p);
        }
      }
    }
  }
  // cross-wave combine
  float4* sm = reinterpret_cast<float4*>(smem);  // [wave][3][D4]
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      sm[(wave * 3 + 0) * D4 + c] = dg[i];
      sm[(wave * 3 + 1) * D4 + c] = db[i];
      sm[(wave * 3 + 2) * D4 + c] = cs[i];
    }
  }
  __syncthreads();
  const int nq = want_colsum ? 3 : 2;
  for (int idx = threadIdx.x; idx < nq * D4; idx += 256) {
    const int q = idx / D4, c = idx - q * D4;
    float4 a = sm[(0 * 3 + q) * D4 + c];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 b = sm[(w * 3 + q) * D4 + c];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    // partial layout: [q][grid][D] so each quantity can be reduced with reduce_partials_kernel
    reinterpret_cast<float4*>(partial + ((int64_t)q * gridDim.x + blockIdx.x) * D)[c] = a;
  }
}

static inline int ln_bwd_blocks(int64_t rows) {
  int64_t b = (rows + 31) / 32;  // >= 32 rows per block
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, float* mean, float* rstd,
                      int64_t rows, int D, float eps, tad_stream_t stream) {
  TAD_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  TAD_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAX_V, "layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D,
              64 * 4 * LN_MAX_V);
  TAD_REQUIRE(y_dtype == TAD_F32 || y_dtype == TAD_OP16, "layernorm_fwd: bad y_dtype %d", y_dtype);
  const int nv = (D / 4 + 63) / 64;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LN_FWD(NV)                                                                                                          \
  if (y_dtype == TAD_OP16)                                                                                                  \
    hipLaunchKernelGGL((layernorm_fwd_kernel<NV, true>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps); \
  else                                                                                                                      \
    hipLaunchKernelGGL((layernorm_fwd_kernel<NV, false>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps);
  switch (nv) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    case 5: LN_FWD(5); break;
    case 6: LN_FWD(6); break;
    default: LN_FWD(8); break;
  }
#undef LN_FWD
  return check_launch("layernorm_fwd");
}

#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
size_t tad_layernorm_bwd_workspace_bytes(int64_t rows, int D) { return (size_t)3 * ln_bwd_blocks(rows) * (size_t)D * sizeof(float); }
#endif

int tad_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* dres, float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, float* colsum_dx,
                      const float* rowscale, int rows_per_scale, int accumulate, void* ws, size_t ws_bytes, int64_t rows, int D,
                      tad_stream_t stream) {
  TAD_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && ws, "layernorm_bwd: null pointer");
  TAD_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * LN_MAX_V, "layernorm_bwd: unsupported D=%d", D);
  TAD_REQUIRE(dy_dtype == TAD_F32 || dy_dtype == TAD_OP16, "layernorm_bwd: bad dy_dtype %d", dy_dtype);
  TAD_REQUIRE(!rowscale || rows_per_scale > 0, "layernorm_bwd: rows_per_scale must be positive");
  const int blocks = ln_bwd_blocks(rows);
  if (ws_bytes < (size_t)3 * blocks * D * sizeof(float)) { set_error("layernorm_bwd: workspace too small"); return TAD_ENOSPACE; }
  const int rows_per_block = (int)((rows + blocks - 1) / blocks);
  const int nv = (D / 4 + 63) / 64;
  const size_t smem = (size_t)4 * 3 * D * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)ws;
  const int want_cs = colsum_dx != nullptr;
  const bool full = (D == 256 * nv);
#define LN_BWD_(NV, BF, FULL_)                                                                                                  \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NV, BF, FULL_>), dim3(blocks), dim3(256), smem, st, dy, x, gamma, mean, rstd, dres, dx, \
                     dx_bf16, partial, rows, D, rows_per_block, want_cs, rowscale, rows_per_scale)
#define LN_BWD(NV)                                                                                                             \
  if (dy_dtype == TAD_OP16) { if (full) LN_BWD_(NV, true, true); else LN_BWD_(NV, true, false); }                               \
  else { if (full) LN_BWD_(NV, false, true); else LN_BWD_(NV, false, false); }
  switch (nv) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    case 5: LN_BWD(5); break;
    case 6: LN_BWD(6); break;
    default: LN_BWD(8); break;
  }
#undef LN_BWD
#undef LN_BWD_
  int rc = check_launch("layernorm_bwd");
  if (rc) return rc;
  rc = launch_reduce_cols(partial, dgamma, dbeta, colsum_dx, colsum_dx ? 3 : 2, blocks, D, accumulate ? 1 : 0, st);
  return rc;
}

}  // extern "C"
