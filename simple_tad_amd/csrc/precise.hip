// "Precise" mode helpers: f32-accurate variants used for the parity gate (end-to-end <= 1e-3 against the fp32 reference),
// not for throughput.
//
//  * (the operand split x = hi + lo behind the precise Linear is tad_split_bf16x3 / tad_split_f16x3, elementwise.hip)
//  * (f32 attention: csrc/attn_f32.hip, exact-f32 MFMA)
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
