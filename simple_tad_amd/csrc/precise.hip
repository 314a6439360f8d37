// "Precise" mode helpers: f32-accurate variants used for the parity gate (end-to-end <= 1e-3 against the fp32 reference),
// not for throughput.
//
//  * (the operand split x = hi + lo behind the precise Linear is tad_split_bf16x3 / tad_split_f16x3, elementwise.hip)
//  * attn_fwd_f32_kernel: plain f32 (VALU FMA) flash-style attention for packed f32 qkv, 64 query rows per workgroup.
#include "common.h"

TAD_NAMESPACE_BEGIN

// x [B,C,T,H,W] f32 -> cols [B*N, K] f32 (same token / k order as im2col_tubelets_kernel)
__global__ void im2col_tubelets_f32_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int C, int T, int H, int W,
                                           int tub, int p) {
  const int W4 = W >> 2;
  const int64_t total = (int64_t)B * C * T * H * W4;
  const int Hp = H / p, Wp = W / p, Tp = T / tub;
  const int K = C * tub * p * p;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i;
    const int w4 = (int)(r % W4); r /= W4;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C); r /= C;
    const int b = (int)r;
    const int w = w4 << 2;
    const int tp = t / tub, kt = t - tp * tub, hp = h / p, kh = h - hp * p, wp = w / p, kw = w - wp * p;
    const int64_t n = ((int64_t)(b * Tp + tp) * Hp + hp) * Wp + wp;
    const int k = ((c * tub + kt) * p + kh) * p + kw;
    *reinterpret_cast<float4*>(cols + n * K + k) = reinterpret_cast<const float4*>(x)[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// f32 attention, any head dim HD that is a multiple of 16 (64: the precise mode of the small / base / large models; 80: the "huge"
// configurations, modeling_finetune.py:390-398 / modeling_pretrain.py:364-386, which have no MFMA kernel).
// Forward.  Workgroup = 256 threads = 64 query rows of one (batch, head); thread (ty = tid>>4, tx = tid&15) owns query rows
// 4ty..4ty+3 and, per 64-key tile, keys 4tx..4tx+3 (scores) / head-dim columns CPT*tx .. CPT*tx+CPT-1 (output), CPT = HD / 16.
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                           int N, int H, float scale) {
  constexpr int CPT = HD / 16;
  __shared__ float Qs[64][HD + 1], Ks[64][HD + 1], Vs[64][HD + 1], Ps[64][65];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 64;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  for (int i = tid; i < 64 * HD; i += 256) {
    const int r = i / HD, c = i - r * HD;
    const int q = min(q0 + r, N - 1);
    Qs[r][c] = base[(int64_t)q * tok + c] * scale;
  }
  float m_run[4], l_run[4], o[4][CPT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m_run[i] = -1e30f; l_run[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) o[i][j] = 0.f;
  }
  for (int kv0 = 0; kv0 < N; kv0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * HD; i += 256) {
      const int r = i / HD, c = i - r * HD;
      const int key = min(kv0 + r, N - 1);
      Ks[r][c] = base[(int64_t)key * tok + (int64_t)H * HD + c];
      Vs[r][c] = base[(int64_t)key * tok + (int64_t)2 * H * HD + c];
    }
    __syncthreads();
    float s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
    for (int d = 0; d < HD; ++d) {
      float qa[4], kb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { qa[i] = Qs[4 * ty + i][d]; kb[i] = Ks[4 * tx + i][d]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[i][j] = fmaf(qa[i], kb[j], s[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kv0 + 4 * tx + j >= N) s[i][j] = -1e30f;
        mx = fmaxf(mx, s[i][j]);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));  // the 16 lanes tx = 0..15 share query rows
      const float m_new = fmaxf(m_run[i], mx);
      const float alpha = __expf(m_run[i] - m_new);
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = __expf(s[i][j] - m_new);
        ps += pv;
        Ps[4 * ty + i][4 * tx + j] = pv;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) ps += __shfl_xor(ps, off, 64);
      l_run[i] = l_run[i] * alpha + ps;
      m_run[i] = m_new;
#pragma unroll
      for (int j = 0; j < CPT; ++j) o[i][j] *= alpha;
    }
    __syncthreads();
    for (int k = 0; k < 64; ++k) {
      float pa[4], vb[CPT];
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[i] = Ps[4 * ty + i][k];
#pragma unroll
      for (int j = 0; j < CPT; ++j) vb[j] = Vs[k][CPT * tx + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CPT; ++j) o[i][j] = fmaf(pa[i], vb[j], o[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q0 + 4 * ty + i;
    if (q >= N) continue;
    const float inv = 1.f / l_run[i];
    float* op = out + (((int64_t)b * N + q) * H + head) * HD + CPT * tx;
#pragma unroll
    for (int j = 0; j < CPT; ++j) op[j] = o[i][j] * inv;
    if (tx == 0 && lse) lse[((int64_t)b * H + head) * N + q] = m_run[i] + __logf(l_run[i]);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// f32 attention backward (verification mode / head dims without an MFMA kernel): delta, dQ pass (per 64 query rows) and dK/dV pass
// (per 64 keys); same thread layout as the forward kernel.  No atomics.
template <int HD>
__global__ void attn_delta_f32_kernel(const float* __restrict__ o, const float* __restrict__ dout, float* __restrict__ delta, int B, int N,
                                      int H) {
  constexpr int CPT = HD / 16;
  const int64_t rows = (int64_t)B * N * H;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid >> 4;
  const int sub = (int)(gid & 15);
  float s = 0.f;
  if (row < rows) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) s = fmaf(o[row * HD + sub * CPT + j], dout[row * HD + sub * CPT + j], s);
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
  if (row < rows && sub == 0) {
    const int h = (int)(row % H);
    const int64_t bq = row / H;
    delta[((bq / N) * H + h) * N + (bq % N)] = s;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              float* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int CPT = HD / 16;
  __shared__ float Qs[64][HD + 1], Gs[64][HD + 1], Ks[64][HD + 1], Vs[64][HD + 1], Ds[64][65];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int head = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * 64;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  for (int i = tid; i < 64 * HD; i += 256) {
    const int r = i / HD, c = i - r * HD;
    const int q = min(q0 + r, N - 1);
    Qs[r][c] = base[(int64_t)q * tok + c] * scale;
    Gs[r][c] = dout[(((int64_t)b * N + q) * H + head) * HD + c];
  }
  float lq[4], dq_[4], acc[4][CPT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = min(q0 + 4 * ty + i, N - 1);
    lq[i] = lse[((int64_t)b * H + head) * N + q];
    dq_[i] = delta[((int64_t)b * H + head) * N + q];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[i][j] = 0.f;
  }
  for (int kv0 = 0; kv0 < N; kv0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * HD; i += 256) {
      const int r = i / HD, c = i - r * HD;
      const int key = min(kv0 + r, N - 1);
      Ks[r][c] = base[(int64_t)key * tok + (int64_t)H * HD + c];
      Vs[r][c] = base[(int64_t)key * tok + (int64_t)2 * H * HD + c];
    }
    __syncthreads();
    float s[4][4], dp[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[i][j] = 0.f; dp[i][j] = 0.f; }
    for (int d = 0; d < HD; ++d) {
      float qa[4], ga[4], kb[4], vb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { qa[i] = Qs[4 * ty + i][d]; ga[i] = Gs[4 * ty + i][d]; kb[i] = Ks[4 * tx + i][d]; vb[i] = Vs[4 * tx + i][d]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[i][j] = fmaf(qa[i], kb[j], s[i][j]); dp[i][j] = fmaf(ga[i], vb[j], dp[i][j]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = (kv0 + 4 * tx + j < N) ? __expf(s[i][j] - lq[i]) : 0.f;
        Ds[4 * ty + i][4 * tx + j] = pv * (dp[i][j] - dq_[i]);
      }
    __syncthreads();
    for (int k = 0; k < 64; ++k) {
      float da[4], kb[CPT];
#pragma unroll
      for (int i = 0; i < 4; ++i) da[i] = Ds[4 * ty + i][k];
#pragma unroll
      for (int j = 0; j < CPT; ++j) kb[j] = Ks[k][CPT * tx + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[i][j] = fmaf(da[i], kb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = q0 + 4 * ty + i;
    if (q >= N) continue;
    float* op = dqkv + ((int64_t)b * N + q) * tok + head * HD + CPT * tx;
#pragma unroll
    for (int j = 0; j < CPT; ++j) op[j] = acc[i][j] * scale;
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int CPT = HD / 16;
  __shared__ float Ks[64][HD + 1], Vs[64][HD + 1], Qs[64][HD + 1], Gs[64][HD + 1], Pt[64][65], Dt[64][65];
  __shared__ float Ls[64], Es[64];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int head = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * 64;
  const int64_t tok = (int64_t)3 * H * HD;
  const float* base = qkv + (int64_t)b * N * tok + head * HD;
  for (int i = tid; i < 64 * HD; i += 256) {
    const int r = i / HD, c = i - r * HD;
    const int key = min(k0 + r, N - 1);
    Ks[r][c] = base[(int64_t)key * tok + (int64_t)H * HD + c];
    Vs[r][c] = base[(int64_t)key * tok + (int64_t)2 * H * HD + c];
  }
  float dk[4][CPT], dv[4][CPT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CPT; ++j) { dk[i][j] = 0.f; dv[i][j] = 0.f; }
  for (int q0 = 0; q0 < N; q0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * HD; i += 256) {
      const int r = i / HD, c = i - r * HD;
      const int q = min(q0 + r, N - 1);
      Qs[r][c] = base[(int64_t)q * tok + c] * scale;
      Gs[r][c] = dout[(((int64_t)b * N + q) * H + head) * HD + c];
    }
    if (tid < 64) {
      const int q = min(q0 + tid, N - 1);
      Ls[tid] = lse[((int64_t)b * H + head) * N + q];
      Es[tid] = delta[((int64_t)b * H + head) * N + q];
    }
    __syncthreads();
    float s[4][4], dp[4][4];  // [key i][query j]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[i][j] = 0.f; dp[i][j] = 0.f; }
    for (int d = 0; d < HD; ++d) {
      float ka[4], va[4], qb[4], gb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ka[i] = Ks[4 * ty + i][d]; va[i] = Vs[4 * ty + i][d]; qb[i] = Qs[4 * tx + i][d]; gb[i] = Gs[4 * tx + i][d]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[i][j] = fmaf(ka[i], qb[j], s[i][j]); dp[i][j] = fmaf(va[i], gb[j], dp[i][j]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ql = 4 * tx + j;
        const float pv = (q0 + ql < N) ? __expf(s[i][j] - Ls[ql]) : 0.f;
        Pt[4 * ty + i][ql] = pv;
        Dt[4 * ty + i][ql] = pv * (dp[i][j] - Es[ql]);
      }
    __syncthreads();
    for (int q = 0; q < 64; ++q) {
      float pa[4], da[4], gb[CPT], qb[CPT];
#pragma unroll
      for (int i = 0; i < 4; ++i) { pa[i] = Pt[4 * ty + i][q]; da[i] = Dt[4 * ty + i][q]; }
#pragma unroll
      for (int j = 0; j < CPT; ++j) { gb[j] = Gs[q][CPT * tx + j]; qb[j] = Qs[q][CPT * tx + j]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < CPT; ++j) { dv[i][j] = fmaf(pa[i], gb[j], dv[i][j]); dk[i][j] = fmaf(da[i], qb[j], dk[i][j]); }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = k0 + 4 * ty + i;
    if (key >= N) continue;
    float* okp = dqkv + ((int64_t)b * N + key) * tok + (int64_t)H * HD + head * HD + CPT * tx;
    // Qs carried the softmax scale already: dK = dS^T (scale*Q)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      okp[j] = dk[i][j];
      okp[(int64_t)H * HD + j] = dv[i][j];
    }
  }
}

// elementwise erf-GELU forward / backward on f32 and f32 column sums (bias gradients) for the precise path
__global__ void gelu_f32_kernel(const float* __restrict__ h, float* __restrict__ a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = h[i];
    a[i] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  }
}
__global__ void gelu_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ h, float* __restrict__ dh, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = h[i];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    dh[i] = dy[i] * (cdf + x * pdf);
  }
}
__global__ void colsum_f32_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t M, int N) {
  // any N: one block per 64 columns; 4 waves over rows, f64 accumulation (verification path)
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (c < N)
    for (int64_t m = wave; m < M; m += 4) s += (double)a[m * N + c];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < N) out[c] = (float)(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}
// N % 4 == 0 (every Linear of the model): one 1024-thread block per 16 columns -- four lanes read one row's 64-byte segment as float4, a
// wave covers 16 rows per instruction, the 16 waves 256 rows per step, four row-steps in flight per lane -- so N / 16 = 48 .. 192 blocks
// stream the matrix instead of N / 64 = 12 .. 48 with one 4-byte load in flight per lane (5.1 ms per call at M = 50176, N = 3072: a third of
// the precise mode's step).  f64 accumulation per lane, then a fixed-order f64 tree over the 256 row lanes: deterministic.
__global__ __launch_bounds__(1024) void colsum_f32_wide_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t M, int N) {
  __shared__ double red[256][17];  // [row lane][column of the block] (+1: no bank conflicts in the tree)
  const int quad = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 16 + quad * 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < N) {
    const float* p = a + c;
    int64_t m = rl;
    for (; m + 768 < M; m += 1024) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + m * N);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (m + 256) * N);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (m + 512) * N);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (m + 768) * N);
      s0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      s1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
      s2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
      s3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
    }
    for (; m < M; m += 256) {
      const float4 v = *reinterpret_cast<const float4*>(p + m * N);
      s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
    }
  }
  red[rl][quad * 4 + 0] = s0; red[rl][quad * 4 + 1] = s1; red[rl][quad * 4 + 2] = s2; red[rl][quad * 4 + 3] = s3;
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {  // 256 row lanes x 16 columns -> 16 sums; thread (r, col) folds row r + half into row r
    const int col = threadIdx.x & 15;
    for (int r = threadIdx.x >> 4; r < half; r += 64) red[r][col] += red[r + half][col];
    __syncthreads();
  }
  if (threadIdx.x < 16 && blockIdx.x * 16 + (int)threadIdx.x < N) out[blockIdx.x * 16 + threadIdx.x] = (float)red[0][threadIdx.x];
}

static inline int grid_for(int64_t items) {
  int64_t g = (items + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

TAD_NAMESPACE_END

TAD_NAMESPACE_BEGIN
int launch_im2col_pairs_f32(const float* x, float* cols, int B, int C, int T, int H, int W, int tubelet, int patch, int ldk, hipStream_t st);
TAD_NAMESPACE_END
using namespace tad;

extern "C" {


int tad_im2col_tubelets_f32(const float* x, float* cols, int B, int C, int T, int H, int W, int tubelet, int patch, tad_stream_t stream) {
  TAD_REQUIRE(x && cols, "im2col_f32: null pointer");
  TAD_REQUIRE(B > 0 && C > 0 && tubelet > 0 && patch > 0 && T % tubelet == 0 && H % patch == 0 && W % patch == 0 && patch % 2 == 0,
              "im2col_f32: T/H/W must be multiples of tubelet/patch and patch even");
  if (patch % 8)  // same rule as the bf16 patch matrix: row stride tad_patch_embed_ldk(), zero padding (ViT-L/14)
    return launch_im2col_pairs_f32(x, cols, B, C, T, H, W, tubelet, patch, tad_patch_embed_ldk(C, tubelet, patch), (hipStream_t)stream);
  hipLaunchKernelGGL(im2col_tubelets_f32_kernel, dim3(grid_for((int64_t)B * C * T * H * (W / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     cols, B, C, T, H, W, tubelet, patch);
  return check_launch("im2col_f32");
}

int tad_attn_fwd_f32(const float* qkv, float* out, float* lse, int B, int N, int H, int d, float scale, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out, "attn_fwd_f32: null pointer");
  TAD_REQUIRE(d == 64 || d == 80, "attn_fwd_f32: head_dim must be 64 or 80 (got %d)", d);
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535 && scale > 0.f, "attn_fwd_f32: bad shape");
  const dim3 grid((N + 63) / 64, H, B);
  if (d == 64) hipLaunchKernelGGL(attn_fwd_f32_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, lse, N, H, scale);
  else hipLaunchKernelGGL(attn_fwd_f32_kernel<80>, grid, dim3(256), 0, (hipStream_t)stream, qkv, out, lse, N, H, scale);
  return check_launch("attn_fwd_f32");
}

int tad_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, float* delta, int B, int N, int H,
                     int d, float scale, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attn_bwd_f32: null pointer");
  TAD_REQUIRE((d == 64 || d == 80) && B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535 && scale > 0.f,
              "attn_bwd_f32: bad shape (head_dim 64 or 80)");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)B * N * H;
  const dim3 dgrid((unsigned)((rows * 16 + 255) / 256)), grid((N + 63) / 64, H, B);
#define BWD_F32(HD_)                                                                                                     \
  {                                                                                                                      \
    hipLaunchKernelGGL(attn_delta_f32_kernel<HD_>, dgrid, dim3(256), 0, st, out, dout, delta, B, N, H);                  \
    hipLaunchKernelGGL(attn_bwd_dq_f32_kernel<HD_>, grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, N, H, scale);   \
    hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel<HD_>, grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, N, H, scale);  \
  }
  if (d == 64) BWD_F32(64) else BWD_F32(80)
#undef BWD_F32
  return check_launch("attn_bwd_f32");
}

int tad_gelu_f32(const float* h, float* a, int64_t n, tad_stream_t stream) {
  TAD_REQUIRE(h && a && n > 0, "gelu_f32: bad args");
  hipLaunchKernelGGL(gelu_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, h, a, n);
  return check_launch("gelu_f32");
}

int tad_gelu_bwd_f32(const float* dy, const float* h, float* dh, int64_t n, tad_stream_t stream) {
  TAD_REQUIRE(dy && h && dh && n > 0, "gelu_bwd_f32: bad args");
  hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, h, dh, n);
  return check_launch("gelu_bwd_f32");
}

int tad_colsum_f32(const float* a, float* out, int64_t M, int N, tad_stream_t stream) {
  TAD_REQUIRE(a && out && M > 0 && N > 0, "colsum_f32: bad args");
  if (N % 4 == 0) hipLaunchKernelGGL(colsum_f32_wide_kernel, dim3((N + 15) / 16), dim3(1024), 0, (hipStream_t)stream, a, out, M, N);
  else hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, a, out, M, N);
  return check_launch("colsum_f32");
}

}  // extern "C"
