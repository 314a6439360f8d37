// MAE pre-training path: the pieces modeling_pretrain.py / engine_for_pretraining.py add around the shared Block
// stack -- visible-token gather, decoder-input assembly, patchify + per-patch normalised pixel targets, MSE.  All HBM-bound
// row movers / reductions; every row is D f32 with D % 4 == 0 (float4 per lane).
#include "common.h"

TAD_NAMESPACE_BEGIN

static inline int rows_grid(int64_t rows) {
  int64_t g = (rows + 3) / 4;  // 4 rows (waves) per 256-thread block
  return (int)(g > 0x7fffffff ? 0x7fffffff : g);
}

// out[r] = src[idx[r]]   (x[~mask].reshape(B,-1,C), modeling_pretrain.py:98; also its transpose-free inverse for backward)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t n_out, int D4) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_out; r += (int64_t)gridDim.x * 4) {
    const float4* s = reinterpret_cast<const float4*>(src) + (int64_t)idx[r] * D4;
    float4* o = reinterpret_cast<float4*>(out) + r * D4;
    for (int c = lane; c < D4; c += 64) o[c] = s[c];
  }
}

// out[idx[r]] = src[r]  (indices unique; rows of out that no index names are left as they are: the caller zero-fills)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           float* __restrict__ out, int64_t n_in, int D4) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_in; r += (int64_t)gridDim.x * 4) {
    const float4* s = reinterpret_cast<const float4*>(src) + r * D4;
    float4* o = reinterpret_cast<float4*>(out) + (int64_t)idx[r] * D4;
    for (int c = lane; c < D4; c += 64) o[c] = s[c];
  }
}

// Decoder input (modeling_pretrain.py:283-288): x_full[b] = cat(x_vis[b] + pos[vis], mask_token + pos[masked])
__global__ __launch_bounds__(256) void mae_assemble_kernel(const float* __restrict__ xv, const float* __restrict__ mask_token,
                                                           const float* __restrict__ pos, const int32_t* __restrict__ vis_idx,
                                                           const int32_t* __restrict__ mask_idx, float* __restrict__ out, int B, int Nv,
                                                           int Nm, int D4) {
  const int lane = threadIdx.x & 63;
  const int N = Nv + Nm;
  const int64_t rows = (int64_t)B * N;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const int b = (int)(r / N), j = (int)(r - (int64_t)b * N);
    const bool vis = j < Nv;
    const int tok = vis ? vis_idx[(int64_t)b * Nv + j] : mask_idx[(int64_t)b * Nm + (j - Nv)];
    const float4* a = vis ? reinterpret_cast<const float4*>(xv) + ((int64_t)b * Nv + j) * D4 : reinterpret_cast<const float4*>(mask_token);
    const float4* p = reinterpret_cast<const float4*>(pos) + (int64_t)tok * D4;
    float4* o = reinterpret_cast<float4*>(out) + r * D4;
    for (int c = lane; c < D4; c += 64) {
      const float4 u = a[c], v = p[c];
      o[c] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
  }
}

// Reconstruction target (engine_for_pretraining.py:51-66): un-normalise the clip, cut tubelet patches, and (normalize_target)
// standardise each (patch, channel) over its tub*p*p pixels with the UNBIASED variance: (v - mean) / (sqrt(var) + 1e-6).
// labels[b, j, pix*3 + c] for the j-th masked token of clip b, pix = (kt*p + kh)*p + kw  ('b n (p c)').
// One workgroup per masked token; wave w < 3 owns channel w (two-pass statistics in registers, tub*p*p <= 64*NPIX pixels).
template <int NPIX>
__global__ __launch_bounds__(256) void mae_target_kernel(const float* __restrict__ x, const int32_t* __restrict__ mask_idx,
                                                         float* __restrict__ labels, int Nm, int T, int H, int W, int tub, int p,
                                                         float m0, float m1, float m2, float s0, float s1, float s2, int normalize) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= 3) return;
  const int64_t row = blockIdx.x;  // b * Nm + j
  const int b = (int)(row / Nm);
  const int tok = mask_idx[row];
  const int Hp = H / p, Wp = W / p;
  const int tp = tok / (Hp * Wp), hw = tok - tp * Hp * Wp, hp = hw / Wp, wp = hw - hp * Wp;
  const int c = wave;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  const int npix = tub * p * p;
  const float* base = x + (((int64_t)b * 3 + c) * T + (int64_t)tp * tub) * H * W + (int64_t)hp * p * W + (int64_t)wp * p;
  float v[NPIX];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NPIX; ++i) {
    const int pix = lane + 64 * i;
    v[i] = 0.f;
    if (pix < npix) {
      const int kt = pix / (p * p), rem = pix - kt * p * p, kh = rem / p, kw = rem - kh * p;
      v[i] = base[((int64_t)kt * H + kh) * W + kw] * sd + mean;  // videos * std + mean
      sum += v[i];
    }
  }
  float mu = 0.f, inv = 1.f;
  if (normalize) {
    mu = wave_sum(sum) / (float)npix;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NPIX; ++i)
      if (lane + 64 * i < npix) sq += (v[i] - mu) * (v[i] - mu);
    const float var = wave_sum(sq) / (float)(npix - 1);
    inv = 1.f / (sqrtf(var) + 1e-6f);
  }
  float* out = labels + row * (int64_t)(npix * 3);
#pragma unroll
  for (int i = 0; i < NPIX; ++i) {
    const int pix = lane + 64 * i;
    if (pix < npix) out[pix * 3 + c] = normalize ? (v[i] - mu) * inv : v[i];
  }
}

// nn.MSELoss (mean): partial[blk] = sum (pred - target)^2 over the block's slice (fixed order); grad = 2 (pred - target) / n
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t n4,
                                                  float inv_n, float* __restrict__ partial, float* __restrict__ grad) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(pred)[i], b = reinterpret_cast<const float4*>(target)[i];
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    if (grad) reinterpret_cast<float4*>(grad)[i] = make_float4(2.f * inv_n * d.x, 2.f * inv_n * d.y, 2.f * inv_n * d.z, 2.f * inv_n * d.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_gather_rows_f32(const float* src, const int32_t* idx, float* out, int64_t n_out, int D, tad_stream_t stream) {
  TAD_REQUIRE(src && idx && out && n_out > 0 && D > 0 && D % 4 == 0, "gather_rows: bad args (D must be a multiple of 4)");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(n_out)), dim3(256), 0, (hipStream_t)stream, src, idx, out, n_out, D / 4);
  return check_launch("gather_rows");
}

int tad_scatter_rows_f32(const float* src, const int32_t* idx, float* out, int64_t n_in, int D, tad_stream_t stream) {
  TAD_REQUIRE(src && idx && out && n_in > 0 && D > 0 && D % 4 == 0, "scatter_rows: bad args (D must be a multiple of 4)");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(rows_grid(n_in)), dim3(256), 0, (hipStream_t)stream, src, idx, out, n_in, D / 4);
  return check_launch("scatter_rows");
}

int tad_mae_assemble(const float* x_vis, const float* mask_token, const float* pos, const int32_t* vis_idx, const int32_t* mask_idx,
                     float* out, int B, int n_vis, int n_mask, int D, tad_stream_t stream) {
  TAD_REQUIRE(x_vis && mask_token && pos && vis_idx && mask_idx && out, "mae_assemble: null pointer");
  TAD_REQUIRE(B > 0 && n_vis > 0 && n_mask > 0 && D > 0 && D % 4 == 0, "mae_assemble: bad shape");
  hipLaunchKernelGGL(mae_assemble_kernel, dim3(rows_grid((int64_t)B * (n_vis + n_mask))), dim3(256), 0, (hipStream_t)stream, x_vis,
                     mask_token, pos, vis_idx, mask_idx, out, B, n_vis, n_mask, D / 4);
  return check_launch("mae_assemble");
}

int tad_mae_target(const float* videos, const int32_t* mask_idx, float* labels, int B, int n_mask, int T, int H, int W, int tubelet,
                   int patch, const float* mean3, const float* std3, int normalize_target, tad_stream_t stream) {
  TAD_REQUIRE(videos && mask_idx && labels && mean3 && std3, "mae_target: null pointer");
  TAD_REQUIRE(B > 0 && n_mask > 0 && tubelet > 0 && patch > 0 && T % tubelet == 0 && H % patch == 0 && W % patch == 0,
              "mae_target: T/H/W must be multiples of tubelet/patch");
  const int npix = tubelet * patch * patch;
  TAD_REQUIRE(npix >= 2 && npix <= 64 * 16, "mae_target: tubelet*patch^2 = %d outside [2, 1024]", npix);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((int64_t)B * n_mask));
#define TGT(NP) hipLaunchKernelGGL((mae_target_kernel<NP>), grid, dim3(256), 0, st, videos, mask_idx, labels, n_mask, T, H, W, tubelet, \
                                   patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], normalize_target ? 1 : 0)
  if (npix <= 64 * 4) TGT(4);
  else if (npix <= 64 * 8) TGT(8);
  else TGT(16);
#undef TGT
  return check_launch("mae_target");
}

int tad_mse_loss_blocks(int64_t n) {
  const int64_t b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

int tad_mse_loss(const float* pred, const float* target, int64_t n, float* partials, float* grad, tad_stream_t stream) {
  TAD_REQUIRE(pred && target && partials && n > 0 && n % 4 == 0, "mse_loss: bad args (n must be a positive multiple of 4)");
  hipLaunchKernelGGL(mse_kernel, dim3(tad_mse_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, pred, target, n / 4, 1.0f / (float)n,
                     partials, grad);
  return check_launch("mse_loss");
}

}  // extern "C"
