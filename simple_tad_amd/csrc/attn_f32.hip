// f32 attention on the matrix pipe: exact-f32 MFMA (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bitwise an fmaf chain; 64 FLOP /
// clk / SIMD = 157 TFLOP/s chip-wide, MI355X_MICROARCH.md) for packed f32 qkv [B,N,3,H,HD], HD = 64 or 80.
//
// Used by the precise mode (the <= 1e-3-per-slice parity gate: split-operand Linears + this) and by head dims without a 16-bit MFMA
// kernel (80: the "huge" configurations, modeling_finetune.py:390-398 / modeling_pretrain.py:364-386).  Round 2 ran these on plain
// VALU FMAs (403 of the precise step's 515 ms); the structure below is the 16-bit kernels' (csrc/attn_fwd.hip, attn_bwd.hip) with
// one f32 VGPR per operand:
//   forward / dQ : a wave owns 32 query rows, the QUERY sits on the MFMA lane (S^T = K Q^T), so softmax statistics are lane-local + one
//                  v_permlane32_swap, and the S^T / dS^T accumulator registers are directly the B operand of O^T += V^T P^T /
//                  dQ^T += K^T dS^T (accumulator row r of lane half h <-> key (r&3) + 8(r>>2) + 4h = the k index both operands use);
//   dK / dV      : a wave owns 32 keys (K / V rows pinned in registers), the KEY sits on the lane (S = Q K^T, dP = dO V^T), P / dS
//                  accumulators feed dV^T += dO^T P and dK^T += Q^T dS the same way.
// The reduction index of the first products is mapped so that a lane reads CONTIGUOUS head-dim elements (lane half h covers
// d in [h HD/2, (h+1) HD/2)): row operands come from the LDS as ds_read_b128 (conflict-free with rows padded to HD_PAD + 4 floats),
// column operands as ds_read_b32 with the lanes along d.  K/V (Q/dO) tiles of 32 rows are staged through registers (f32 rows cannot
// take the 16-byte LDS-DMA into a padded image); 2-3 workgroups per CU hide the staging.  Two kernels in backward, no atomics, as in
// the 16-bit path: dQ recomputes S and dP.
#include "common.h"

TAD_NAMESPACE_BEGIN

constexpr float F32_LOG2E = 1.44269504088896340736f;
constexpr float F32_LN2 = 0.69314718055994530942f;

template <int HD>
struct AttnF32 {
  static constexpr int HALF = HD / 2;              // head-dim elements per lane half in the first products
  static constexpr int NDT = (HD + 31) / 32;       // 32-row d tiles of the second products (80 -> 3, the last one half empty)
  static constexpr int STRIDE = NDT * 32 + 4;      // floats per LDS row: d tiles never read past the row, +4 keeps b128 row reads conflict-free
  static_assert(HD % 8 == 0 && HD <= 96, "head dim");
};

__device__ __forceinline__ float swap_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// stage 32 rows x HD floats (row `first + r` of a [rows][row_stride] f32 tensor, clamped to `last`) into tile[32][STRIDE], times mul
template <int HD>
__device__ __forceinline__ void stage_rows(float* tile, const float* src, int64_t row_stride, int first, int last, float mul, int tid) {
  constexpr int V4 = HD / 4;  // float4 per row
  for (int i = tid; i < 32 * V4; i += 256) {
    const int r = i / V4, c = (i - r * V4) * 4;
    const int row = min(first + r, last);
    float4 v = *reinterpret_cast<const float4*>(src + (int64_t)row * row_stride + c);
    v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
    *reinterpret_cast<float4*>(tile + r * AttnF32<HD>::STRIDE + c) = v;
  }
}

// acc += rowsT * colB : acc[i][j] += sum_k tile[i = lane&31][h HALF + k'] * b[k'] over the lane's half of the head dim
template <int HD>
__device__ __forceinline__ void mfma_rows(f32x16& acc, const float* tile, const float (&b)[AttnF32<HD>::HALF], int ql, int h) {
  constexpr int HALF = AttnF32<HD>::HALF;
  const float* rp = tile + ql * AttnF32<HD>::STRIDE + h * HALF;
#pragma unroll
  for (int s4 = 0; s4 < HALF / 4; ++s4) {
    const float4 a = *reinterpret_cast<const float4*>(rp + 4 * s4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * s4 + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * s4 + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * s4 + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * s4 + 3], acc, 0, 0, 0);
  }
}

// out[dt] += tile^T x : out[dt][d i][j] += sum_rows tile[row(s, h)][32 dt + i] * x[s]   (x = an accumulator tile used as the B operand)
template <int HD>
__device__ __forceinline__ void mfma_cols(f32x16 (&out)[AttnF32<HD>::NDT], const float* tile, const f32x16& x, int ql, int h) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float* rp = tile + acc_row(s, h) * AttnF32<HD>::STRIDE + ql;
#pragma unroll
    for (int dt = 0; dt < AttnF32<HD>::NDT; ++dt) out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(rp[32 * dt], x[s], out[dt], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------------------------------- forward
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                              int N, int H, float scale, const Drop drop) {
  using C = AttnF32<HD>;
  __shared__ __attribute__((aligned(16))) float Kt[32 * C::STRIDE], Vt[32 * C::STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = min(q0 + ql, N - 1);
  float q[C::HALF];
#pragma unroll
  for (int s = 0; s < C::HALF; ++s) q[s] = base[(int64_t)qrow * tok + h * C::HALF + s] * (scale * F32_LOG2E);  // scores in log2 units
  f32x16 o[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  for (int kv0 = 0; kv0 < N; kv0 += 32) {
    __syncthreads();
    stage_rows<HD>(Kt, base + (int64_t)H * HD, tok, kv0, N - 1, 1.f, tid);
    stage_rows<HD>(Vt, base + (int64_t)2 * H * HD, tok, kv0, N - 1, 1.f, tid);
    __syncthreads();
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    mfma_rows<HD>(s, Kt, q, ql, h);
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (kv0 + acc_row(r, h) >= N) s[r] = -1e30f;
      mx = fmaxf(mx, s[r]);
    }
    mx = swap_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = fast_exp2(s[r] - m_new);
      ps += s[r];
    }
    l_run = l_run * alpha + swap_sum(ps);   // the normaliser is the sum of the UN-dropped probabilities
    m_run = m_new;
    if (DROP) {
      const uint32_t row = (uint32_t)((b * H + head) * N + qrow);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = drop_keep(drop, row, (uint32_t)(kv0 + acc_row(r, h))) ? s[r] * drop.inv_keep : 0.f;
    }
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    mfma_cols<HD>(o, Vt, s, ql, h);
  }
  if (q0 + ql < N) {
    const float inv = 1.f / l_run;
    float* op = out + (((int64_t)b * N + qrow) * H + head) * HD;
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * dt + 8 * r4 + 4 * h;
        if (d < HD) *reinterpret_cast<float4*>(op + d) = make_float4(o[dt][4 * r4] * inv, o[dt][4 * r4 + 1] * inv, o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv);
      }
    if (h == 0 && lse) lse[((int64_t)b * H + head) * N + qrow] = (m_run + __log2f(l_run)) * F32_LN2;
  }
}

// ------------------------------------------------------------------------------------------------------------------------- dQ (+ delta)
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                                 const float* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, float* __restrict__ dqkv, int N, int H, float scale,
                                                                 const Drop drop) {
  using C = AttnF32<HD>;
  __shared__ __attribute__((aligned(16))) float Kt[32 * C::STRIDE], Vt[32 * C::STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = min(q0 + ql, N - 1);
  const int64_t orow = (((int64_t)b * N + qrow) * H + head) * HD + h * C::HALF;
  float q[C::HALF], g[C::HALF];
  float part = 0.f;
#pragma unroll
  for (int s = 0; s < C::HALF; ++s) {
    q[s] = base[(int64_t)qrow * tok + h * C::HALF + s] * (scale * F32_LOG2E);
    g[s] = dout[orow + s];
    part = fmaf(g[s], out[orow + s], part);
  }
  const float dlt = swap_sum(part);  // delta = rowsum(dO o O)
  const int64_t ridx = ((int64_t)b * H + head) * N + qrow;
  if (q0 + ql < N && h == 0) delta[ridx] = dlt;  // for the dK/dV kernel (launched behind this one on the same stream)
  const float lse2 = lse[ridx] * F32_LOG2E;
  f32x16 dq[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  for (int kv0 = 0; kv0 < N; kv0 += 32) {
    __syncthreads();
    stage_rows<HD>(Kt, base + (int64_t)H * HD, tok, kv0, N - 1, 1.f, tid);
    stage_rows<HD>(Vt, base + (int64_t)2 * H * HD, tok, kv0, N - 1, 1.f, tid);
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    mfma_rows<HD>(s, Kt, q, ql, h);
    mfma_rows<HD>(dp, Vt, g, ql, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (kv0 + acc_row(r, h) < N) ? fast_exp2(s[r] - lse2) : 0.f;
      float dpr = dp[r];  // d(loss) / d(dropped probability); through the dropout: x mask / (1 - p)
      if (DROP) dpr = drop_keep(drop, (uint32_t)((b * H + head) * N + qrow), (uint32_t)(kv0 + acc_row(r, h))) ? dpr * drop.inv_keep : 0.f;
      s[r] = p * (dpr - dlt);  // dS^T
    }
    mfma_cols<HD>(dq, Kt, s, ql, h);
  }
  if (q0 + ql < N) {
    float* op = dqkv + ((int64_t)b * N + qrow) * tok + head * HD;
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * dt + 8 * r4 + 4 * h;
        if (d < HD) *reinterpret_cast<float4*>(op + d) = make_float4(dq[dt][4 * r4] * scale, dq[dt][4 * r4 + 1] * scale, dq[dt][4 * r4 + 2] * scale, dq[dt][4 * r4 + 3] * scale);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------- dK, dV
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dqkv, int N, int H, float scale, const Drop drop) {
  using C = AttnF32<HD>;
  __shared__ __attribute__((aligned(16))) float Qt[32 * C::STRIDE], Gt[32 * C::STRIDE], Ls[32], Ds[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  const float* gbase = dout + ((int64_t)b * N * H + head) * HD;  // row q at + q * H * HD
  const int k0 = blockIdx.x * 128 + wave * 32;
  const int krow = min(k0 + kl, N - 1);
  float kf[C::HALF], vf[C::HALF];
#pragma unroll
  for (int s = 0; s < C::HALF; ++s) {
    kf[s] = base[(int64_t)krow * tok + (int64_t)H * HD + h * C::HALF + s];
    vf[s] = base[(int64_t)krow * tok + (int64_t)2 * H * HD + h * C::HALF + s];
  }
  f32x16 dk[C::NDT], dv[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
  const int64_t rbase = ((int64_t)b * H + head) * N;
  for (int qt0 = 0; qt0 < N; qt0 += 32) {
    __syncthreads();
    stage_rows<HD>(Qt, base, tok, qt0, N - 1, scale * F32_LOG2E, tid);   // Q in the score units of the forward (log2)
    stage_rows<HD>(Gt, gbase, (int64_t)H * HD, qt0, N - 1, 1.f, tid);
    if (tid < 32) {
      const int qi = qt0 + tid;
      Ls[tid] = qi < N ? lse[rbase + qi] * F32_LOG2E : 1e30f;  // rows past the sequence: P = exp2(S - 1e30) = 0
      Ds[tid] = qi < N ? delta[rbase + qi] : 0.f;
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    mfma_rows<HD>(s, Qt, kf, kl, h);    // S[query i][key j]
    mfma_rows<HD>(dp, Gt, vf, kl, h);   // dP[query i][key j]
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 l4 = *reinterpret_cast<const float4*>(Ls + 8 * r4 + 4 * h);
      const float4 d4 = *reinterpret_cast<const float4*>(Ds + 8 * r4 + 4 * h);
      const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = fast_exp2(s[4 * r4 + e] - la[e]);
        float pd = p, dpr = dp[4 * r4 + e];
        if (DROP) {
          const bool keep = drop_keep(drop, (uint32_t)((b * H + head) * N + min(qt0 + 8 * r4 + 4 * h + e, N - 1)), (uint32_t)krow);
          pd = keep ? p * drop.inv_keep : 0.f;
          dpr = keep ? dpr * drop.inv_keep : 0.f;
        }
        s[4 * r4 + e] = pd;                     // dropped P (what multiplied V in the forward)
        dp[4 * r4 + e] = p * (dpr - da[e]);     // dS
      }
    }
    mfma_cols<HD>(dv, Gt, s, kl, h);    // dV^T += dO^T P
    mfma_cols<HD>(dk, Qt, dp, kl, h);   // dK^T += (scale log2e Q)^T dS
  }
  if (k0 + kl < N) {
    float* okp = dqkv + ((int64_t)b * N + krow) * tok + (int64_t)H * HD + head * HD;
    float* ovp = okp + (int64_t)H * HD;
    const float ks = 1.f / F32_LOG2E;  // Qt carried scale * log2e: dK = scale * dS^T Q
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = 32 * dt + 8 * r4 + 4 * h;
        if (d < HD) {
          *reinterpret_cast<float4*>(okp + d) = make_float4(dk[dt][4 * r4] * ks, dk[dt][4 * r4 + 1] * ks, dk[dt][4 * r4 + 2] * ks, dk[dt][4 * r4 + 3] * ks);
          *reinterpret_cast<float4*>(ovp + d) = make_float4(dv[dt][4 * r4], dv[dt][4 * r4 + 1], dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]);
        }
      }
  }
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_attn_fwd_f32(const float* qkv, float* out, float* lse, int B, int N, int H, int d, float scale, float dropout_p, uint32_t seed,
                     tad_stream_t stream) {
  TAD_REQUIRE(qkv && out, "attn_fwd_f32: null pointer");
  TAD_REQUIRE(d == 64 || d == 80, "attn_fwd_f32: head_dim must be 64 or 80 (got %d)", d);
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535 && (int64_t)B * H * N < (1ll << 32), "attn_fwd_f32: bad shape");
  TAD_REQUIRE(scale > 0.f, "attn_fwd_f32: scale must be positive");
  Drop drop;
  TAD_REQUIRE(make_drop(dropout_p, seed, &drop), "attn_fwd_f32: dropout_p=%g outside [0, 1)", (double)dropout_p);
  const dim3 grid((N + 127) / 128, H, B);
#define FWD(HD_, DR_) hipLaunchKernelGGL((attn_fwd_f32_kernel<HD_, DR_>), grid, dim3(256), 0, (hipStream_t)stream, qkv, out, lse, N, H, scale, drop)
  if (d == 64) { if (dropout_p > 0.f) FWD(64, true); else FWD(64, false); }
  else { if (dropout_p > 0.f) FWD(80, true); else FWD(80, false); }
#undef FWD
  return check_launch("attn_fwd_f32");
}

int tad_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta, int B, int N, int H,
                     int d, float scale, float dropout_p, uint32_t seed, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attn_bwd_f32: null pointer");
  TAD_REQUIRE(d == 64 || d == 80, "attn_bwd_f32: head_dim must be 64 or 80 (got %d)", d);
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535 && (int64_t)B * H * N < (1ll << 32), "attn_bwd_f32: bad shape");
  TAD_REQUIRE(scale > 0.f, "attn_bwd_f32: scale must be positive");
  Drop drop;
  TAD_REQUIRE(make_drop(dropout_p, seed, &drop), "attn_bwd_f32: dropout_p=%g outside [0, 1)", (double)dropout_p);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((N + 127) / 128, H, B);
#define DQ(HD_, DR_) hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<HD_, DR_>), grid, dim3(256), 0, st, qkv, out, dout, lse, delta, dqkv, N, H, scale, drop)
#define DKV(HD_, DR_) hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<HD_, DR_>), grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, N, H, scale, drop)
  const bool dr = dropout_p > 0.f;
  if (d == 64) { if (dr) DQ(64, true); else DQ(64, false); } else { if (dr) DQ(80, true); else DQ(80, false); }
  int rc = check_launch("attn_bwd_dq_f32");
  if (rc) return rc;
  if (d == 64) { if (dr) DKV(64, true); else DKV(64, false); } else { if (dr) DKV(80, true); else DKV(80, false); }
#undef DQ
#undef DKV
  return check_launch("attn_bwd_dkv_f32");
}

}  // extern "C"
