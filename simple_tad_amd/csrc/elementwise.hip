// HBM-bound helper kernels: casts, tubelet im2col, mean-pool, column sums, sum of squares.
// All are streaming kernels: 16-byte vector accesses, grid capped at ~8 blocks/CU with grid-stride loops.
#include "common.h"

TAD_NAMESPACE_BEGIN

static inline int capped_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- cast f32 -> bf16
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    uint4 o;
    o.x = pack_op16x2(a.x, a.y);
    o.y = pack_op16x2(a.z, a.w);
    o.z = pack_op16x2(b.x, b.y);
    o.w = pack_op16x2(b.z, b.w);
    reinterpret_cast<uint4*>(dst)[i] = o;
  }
  // tail
  for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f32_to_op16(src[i]);
}

// ---------------------------------------------------------------- transpose + cast: src [R,C] f32 -> dst [C,R] bf16
// 64x64 tile through LDS (padded), coalesced on both sides.
__global__ void transpose_cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int R, int C) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * C + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * R + r] = f32_to_op16(tile[tx][i]);
  }
}

// ---------------------------------------------------------------- batched bf16 transpose (operand mirrors of all weights, one launch)
// One workgroup per 64x64 tile of some matrix of the batch; the tile list is precomputed on the host (flat layouts never change):
// table[t] = {src offset of the tile, dst offset of the tile, C (src row pitch), R (dst row pitch), valid rows, valid cols,
// 0, 0} in elements; every matrix has R % 8 == C % 8 == 0, so tiles are made of whole 16-byte chunks.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                                     const int4* __restrict__ table) {
  __shared__ uint16_t tile[64][66];
  const int4 t0 = table[2 * blockIdx.x], t1 = table[2 * blockIdx.x + 1];
  const int64_t so = (uint32_t)t0.x, d_o = (uint32_t)t0.y;
  const int C = t0.z, R = t0.w, rows = t1.x, cols = t1.y;
  const int r = threadIdx.x >> 2, ch = (threadIdx.x & 3) * 2;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = (ch + k) * 8;
    if (r < rows && c0 < cols) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + so + (int64_t)r * C + c0);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tile[r][c0 + 2 * e] = (uint16_t)(w[e] & 0xffffu);
        tile[r][c0 + 2 * e + 1] = (uint16_t)(w[e] >> 16);
      }
    }
  }
  __syncthreads();
  const int c = r;  // dst row = src column
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r0 = (ch + k) * 8;
    if (c < cols && r0 < rows) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[r0 + 2 * e][c] | ((uint32_t)tile[r0 + 2 * e + 1][c] << 16);
      *reinterpret_cast<uint4*>(dst + d_o + (int64_t)c * R + r0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// ---------------------------------------------------------------- input stage: uint8 frames -> normalised bf16 patch matrix
// frames [B,T,H,W,3] uint8 (decoder / cv2 layout) -> cols [B*N, 3*tub*p*p] bf16 with the reference's arithmetic
//   v = (float(u8) / 255 - mean[c]) / std[c]      (run_inference.py:22-32 prepare_image; dota.py:443-460 tensor_normalize)
// in f32 with IEEE divisions, then the same RNE cast to bf16 as the f32 path -- so the patch matrix is bit-identical to
// im2col_tubelets(normalise(frames)) while reading 4x fewer bytes and skipping the f32 clip entirely.  ``bgr``: the channel in
// memory is 2 - c (cv2 frames; the reference converts with cv2.cvtColor(BGR2RGB)).  ``t_offset``: frame t of the clip lives in
// slot (t + t_offset) % T of the buffer (sliding-window ring buffer of run_inference.py:86-93 without shifting the frames).
// One thread: 8 consecutive pixels of one row = 24 contiguous bytes in, three 16-byte chunks out (one per channel).
__global__ void im2col_tubelets_u8_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ cols, int B, int T, int H, int W,
                                          int tub, int p, float m0, float m1, float m2, float s0, float s1, float s2, int bgr,
                                          int t_offset) {
  const int W8 = W >> 3;
  const int64_t total = (int64_t)B * T * H * W8;
  const int Hp = H / p, Wp = W / p, Tp = T / tub;
  const int K = 3 * tub * p * p;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i;
    const int w8 = (int)(r % W8); r /= W8;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); r /= T;  // frame index inside the clip
    const int b = (int)r;
    int slot = t + t_offset;
    slot = slot >= T ? slot - T : slot;
    const uint8_t* src = frames + ((((int64_t)b * T + slot) * H + h) * W + (int64_t)w8 * 8) * 3;
    uint32_t raw[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) raw[q] = reinterpret_cast<const uint32_t*>(src)[q];  // (W*3*8) % 4 == 0: 4-byte aligned
    const int w = w8 * 8;
    const int tp = t / tub, kt = t - tp * tub, hp = h / p, kh = h - hp * p, wp = w / p, kw = w - wp * p;
    const int64_t n = ((int64_t)b * Tp + tp) * Hp * Wp + (int64_t)hp * Wp + wp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cm = bgr ? 2 - c : c;  // channel position in memory
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int byte = e * 3 + cm;
        const float u = (float)((raw[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
        v[e] = (u / 255.0f - mean[c]) / sd[c];
      }
      const int k = ((c * tub + kt) * p + kh) * p + kw;
      *reinterpret_cast<uint4*>(cols + n * K + k) =
          make_uint4(pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3]), pack_op16x2(v[4], v[5]), pack_op16x2(v[6], v[7]));
    }
  }
}

// The same for even patch sizes that are not multiples of 8 (/14): one thread = 2 pixels = 6 contiguous bytes in, three 4-byte pairs out
// (a pair never straddles a patch: p and w are even); rows have stride ldk = K rounded up to 64 and the thread that owns k = 0 of a
// token zeroes the padding columns, as im2col_tubelets_pairs_kernel does for the f32 clip.
__global__ void im2col_tubelets_u8_pairs_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ cols, int B, int T, int H, int W,
                                                int tub, int p, int ldk, float m0, float m1, float m2, float s0, float s1, float s2, int bgr,
                                                int t_offset) {
  const int W2 = W >> 1;
  const int64_t total = (int64_t)B * T * H * W2;
  const int Hp = H / p, Wp = W / p, Tp = T / tub;
  const int K = 3 * tub * p * p;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i;
    const int w2 = (int)(r % W2); r /= W2;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); r /= T;
    const int b = (int)r;
    int slot = t + t_offset;
    slot = slot >= T ? slot - T : slot;
    const uint8_t* src = frames + ((((int64_t)b * T + slot) * H + h) * W + (int64_t)w2 * 2) * 3;  // 6 bytes, 2-byte aligned
    const uint16_t* s2p = reinterpret_cast<const uint16_t*>(src);
    const uint32_t lo = (uint32_t)s2p[0] | ((uint32_t)s2p[1] << 16);  // bytes 0..3
    const uint32_t hi = (uint32_t)s2p[2];                             // bytes 4..5
    const int w = w2 * 2;
    const int tp = t / tub, kt = t - tp * tub, hp = h / p, kh = h - hp * p, wp = w / p, kw = w - wp * p;
    const int64_t n = ((int64_t)b * Tp + tp) * Hp * Wp + (int64_t)hp * Wp + wp;
    uint16_t* row = cols + n * ldk;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cm = bgr ? 2 - c : c;
      const int b0 = cm, b1 = 3 + cm;  // byte positions of the two pixels' channel
      const float u0 = (float)((lo >> (b0 * 8)) & 0xffu);
      const float u1 = (float)(((b1 < 4 ? lo >> (b1 * 8) : hi >> ((b1 - 4) * 8))) & 0xffu);
      const int k = ((c * tub + kt) * p + kh) * p + kw;
      *reinterpret_cast<uint32_t*>(row + k) = pack_op16x2((u0 / 255.0f - mean[c]) / sd[c], (u1 / 255.0f - mean[c]) / sd[c]);
    }
    if (kt == 0 && kh == 0 && kw == 0)  // (k = 0 of channel 0)
      for (int z = K; z < ldk; z += 2) *reinterpret_cast<uint32_t*>(row + z) = 0u;
  }
}

// ---------------------------------------------------------------- tubelet im2col
// x [B,C,T,H,W] f32 -> cols [B*N, K] bf16, token n = (t'*H' + h')*W' + w', k = ((c*tub+kt)*p+kh)*p+kw.
// One thread moves 8 consecutive w (32 B in, 16 B out); threads walk x in memory order -> coalesced reads.
__global__ void im2col_tubelets_kernel(const float* __restrict__ x, uint16_t* __restrict__ cols, int B, int C,
                                       int T, int H, int W, int tub, int p) {
  const int W8 = W >> 3;
  const int64_t total = (int64_t)B * C * T * H * W8;
  const int Hp = H / p, Wp = W / p, Tp = T / tub;
  const int K = C * tub * p * p;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i;
    const int w8 = (int)(r % W8); r /= W8;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C); r /= C;
    const int b = (int)r;
    const int w = w8 << 3;
    const float4 a0 = reinterpret_cast<const float4*>(x)[2 * i];
    const float4 a1 = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const int tp = t / tub, kt = t - tp * tub;
    const int hp = h / p, kh = h - hp * p;
    const int wp = w / p, kw = w - wp * p;
    const int64_t n = ((int64_t)(b * Tp + tp) * Hp + hp) * Wp + wp;
    const int k = ((c * tub + kt) * p + kh) * p + kw;
    uint4 o;
    o.x = pack_op16x2(a0.x, a0.y);
    o.y = pack_op16x2(a0.z, a0.w);
    o.z = pack_op16x2(a1.x, a1.y);
    o.w = pack_op16x2(a1.z, a1.w);
    *reinterpret_cast<uint4*>(cols + n * K + k) = o;
  }
}

// Patch sizes that are even but not a multiple of 8 (ViT-L/14: p = 14, K = 3*2*14*14 = 1176): one thread moves 2 consecutive w.  The
// patch matrix gets a row stride of ldk = K rounded up to the GEMM's K-tile (64); the thread that owns k = 0 of a token zeroes the
// ldk - K padding columns of its row, so the padded matrix times a zero-padded weight is the un-padded product exactly.
template <typename OutT>
__global__ void im2col_tubelets_pairs_kernel(const float* __restrict__ x, OutT* __restrict__ cols, int B, int C, int T, int H, int W,
                                             int tub, int p, int ldk) {
  const int W2 = W >> 1;
  const int64_t total = (int64_t)B * C * T * H * W2;
  const int Hp = H / p, Wp = W / p, Tp = T / tub;
  const int K = C * tub * p * p;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r = i;
    const int w2 = (int)(r % W2); r /= W2;
    const int h = (int)(r % H); r /= H;
    const int t = (int)(r % T); r /= T;
    const int c = (int)(r % C); r /= C;
    const int b = (int)r;
    const int w = w2 << 1;
    const float2 a = reinterpret_cast<const float2*>(x)[i];
    const int tp = t / tub, kt = t - tp * tub, hp = h / p, kh = h - hp * p, wp = w / p, kw = w - wp * p;
    const int64_t n = ((int64_t)(b * Tp + tp) * Hp + hp) * Wp + wp;
    const int k = ((c * tub + kt) * p + kh) * p + kw;
    OutT* row = cols + n * ldk;
    if constexpr (sizeof(OutT) == 2) *reinterpret_cast<uint32_t*>(row + k) = pack_op16x2(a.x, a.y);
    else *reinterpret_cast<float2*>(row + k) = a;
    if (k == 0) {
      for (int z = K; z < ldk; z += 2) {
        if constexpr (sizeof(OutT) == 2) *reinterpret_cast<uint32_t*>(row + z) = 0u;
        else *reinterpret_cast<float2*>(row + z) = make_float2(0.f, 0.f);
      }
    }
  }
}
template __global__ void im2col_tubelets_pairs_kernel<uint16_t>(const float*, uint16_t*, int, int, int, int, int, int, int, int);
template __global__ void im2col_tubelets_pairs_kernel<float>(const float*, float*, int, int, int, int, int, int, int, int);

// ---------------------------------------------------------------- mean-pool
// partial[b][s][D]: block (x = column chunk of 256, y = split s, z = b); 4 waves stride over the rows of the split.
__global__ void meanpool_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int N, int D) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col4 = blockIdx.x * 64 + lane;  // float4 column index
  const int s = blockIdx.y, b = blockIdx.z;
  const int D4 = D >> 2;
  const int rows_per = (N + TAD_POOL_SPLIT - 1) / TAD_POOL_SPLIT;
  const int n0 = s * rows_per, n1 = min(N, n0 + rows_per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col4 < D4) {
    const float4* base = reinterpret_cast<const float4*>(x) + (int64_t)b * N * D4 + col4;
    for (int n = n0 + wave; n < n1; n += 4) {
      const float4 v = base[(int64_t)n * D4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col4 < D4) {
    float4 t = red[0][lane];
    for (int w = 1; w < 4; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
    reinterpret_cast<float4*>(partial)[((int64_t)b * TAD_POOL_SPLIT + s) * D4 + col4] = t;
  }
}
__global__ void meanpool_final_kernel(const float* __restrict__ partial, float* __restrict__ y, int B, int N, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, j = i - b * D;
  float s = 0.f;
  for (int k = 0; k < TAD_POOL_SPLIT; ++k) s += partial[((int64_t)b * TAD_POOL_SPLIT + k) * D + j];
  y[i] = s / (float)N;
}
__global__ void meanpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, uint16_t* __restrict__ dxb,
                                    int B, int N, int D) {
  const int D4 = D >> 2;
  const int64_t total = (int64_t)B * N * D4;
  const float inv = 1.f / (float)N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % D4);
    const int b = (int)(i / ((int64_t)N * D4));
    float4 v = reinterpret_cast<const float4*>(dy)[(int64_t)b * D4 + c4];
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    if (dx) reinterpret_cast<float4*>(dx)[i] = v;
    if (dxb) {
      uint2 o;
      o.x = pack_op16x2(v.x, v.y);
      o.y = pack_op16x2(v.z, v.w);
      reinterpret_cast<uint2*>(dxb)[i] = o;
    }
  }
}

// ---------------------------------------------------------------- column sums of a bf16 matrix
// stage 1: block (x: chunk of 512 columns, y: row split) -> partial [splits][N]; each thread owns 8 columns (16 B loads)
// and blockDim.y... here: 256 threads = 64 lanes (8 cols each = 512 cols) x 4 waves over rows.
__global__ void colsum_partial_kernel(const uint16_t* __restrict__ a, float* __restrict__ partial, int64_t M, int N,
                                      int rows_per) {
  __shared__ float red[4][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 8;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
    for (int64_t m = m0 + wave; m < m1; m += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(a + m * N + c0);
      acc[0] += op16_lo_f32(v.x); acc[1] += op16_hi_f32(v.x);
      acc[2] += op16_lo_f32(v.y); acc[3] += op16_hi_f32(v.y);
      acc[4] += op16_lo_f32(v.z); acc[5] += op16_hi_f32(v.z);
      acc[6] += op16_lo_f32(v.w); acc[7] += op16_hi_f32(v.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane][j] = acc[j];
  __syncthreads();
  if (wave == 0 && c0 < N) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      partial[(int64_t)blockIdx.y * N + c0 + j] = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
  }
}
// f32 variant over a row window of a [B, R, N] tensor: sums rows r in [r0, r0 + rc) of every batch entry (the mask-token gradient of the
// MAE decoder input: the masked positions of every clip, modeling_pretrain.py:283-287) -- no contiguous copy of the window needed.
// 256 threads = 64 lanes (4 columns each = 256 columns) x 4 waves over rows; blockIdx.y = row split over the B * rc window rows.
__global__ void colsum_window_f32_kernel(const float* __restrict__ a, float* __restrict__ partial, int B, int R, int N, int r0, int rc,
                                         int rows_per) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + lane) * 4;
  const int64_t total = (int64_t)B * rc;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per, m1 = min(total, m0 + rows_per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < N) {
    for (int64_t m = m0 + wave; m < m1; m += 4) {
      const int64_t b = m / rc, r = m - b * rc;
      const float4 v = *reinterpret_cast<const float4*>(a + ((b * R + r0 + r) * (int64_t)N + c0));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c0 < N) {
    float4 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
    *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.y * N + c0) = t;
  }
}
// stage 2 (also used for LayerNorm dgamma/dbeta and split-K slabs): out[j] (+)= sum_s partial[s][j]
__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int splits,
                                       int64_t n, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 s = reinterpret_cast<const float4*>(partial)[i];
    for (int k = 1; k < splits; ++k) {
      const float4 v = reinterpret_cast<const float4*>(partial + (int64_t)k * n)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
      const float4 o = reinterpret_cast<float4*>(out)[i];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(int64_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
  }
}


// Same reduction for the "many partial rows, few columns" case (LayerNorm dgamma/dbeta over ~1000 blocks, column sums):
// one 1024-thread block per 256 columns; the 16 waves stride over the partial rows, 4 loads in flight each, then combine
// through LDS.  blockIdx.y selects one of up to 3 (partial, out) pairs so one launch serves all: pair q starts at
// partial + q * q_stride and its partial rows are row_stride floats apart ([q][splits][n]: q_stride = splits * n, row_stride = n;
// column ranges of one [splits][N] array: q_stride = distance between the ranges, row_stride = N).
// Block = 16 float4 columns x 64 row lanes (lane = 16 * row-sub + column inside a wave: 256 contiguous bytes per partial row and
// wave quarter): a [1024][3][768] LayerNorm partial is spread over 36 blocks instead of 9 (round 1: 64 columns per block, 12 us per launch
// with 9 of 256 CUs pulling 9.4 MB).
__global__ __launch_bounds__(1024) void reduce_cols_kernel(const float* __restrict__ partial, float* out0, float* out1, float* out2,
                                                           int splits, int n, int accumulate, int64_t q_stride, int row_stride) {
  __shared__ float4 red[64][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, rl = wave * 4 + (lane >> 4);  // row lane 0..63
  const int q = blockIdx.y;
  float* out = q == 0 ? out0 : (q == 1 ? out1 : out2);
  const float* base = partial + (int64_t)q * q_stride;
  const int c4 = blockIdx.x * 16 + col;
  const int n4 = n >> 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
    int s = rl;
    for (; s + 192 < splits; s += 256) {
      const float4 a = reinterpret_cast<const float4*>(base + (int64_t)s * row_stride)[c4];
      const float4 b = reinterpret_cast<const float4*>(base + (int64_t)(s + 64) * row_stride)[c4];
      const float4 c = reinterpret_cast<const float4*>(base + (int64_t)(s + 128) * row_stride)[c4];
      const float4 d = reinterpret_cast<const float4*>(base + (int64_t)(s + 192) * row_stride)[c4];
      acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
      acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; s < splits; s += 64) {
      const float4 a = reinterpret_cast<const float4*>(base + (int64_t)s * row_stride)[c4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
  }
  red[rl][col] = acc;
  __syncthreads();
  // 64 -> 16 -> 1 per column in a fixed order (deterministic); every barrier is reached by all 16 waves
  const int fc = threadIdx.x & 15, fg = (threadIdx.x >> 4) & 15;  // thread < 256: folds rows 4 fg .. 4 fg + 3 of column fc
  float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < 256) {
    t4 = red[4 * fg][fc];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const float4 v = red[4 * fg + w][fc]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
  }
  __syncthreads();
  if (threadIdx.x < 256) red[fg][fc] = t4;
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + (int)threadIdx.x < n4) {
    const int c = threadIdx.x;
    float4 t = red[0][c];
#pragma unroll
    for (int w = 1; w < 16; ++w) { const float4 v = red[w][c]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    float* o = out + (int64_t)(blockIdx.x * 16 + c) * 4;
    if (accumulate) {
      const float4 p = *reinterpret_cast<float4*>(o);
      t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
    }
    *reinterpret_cast<float4*>(o) = t;
  }
}

// ---------------------------------------------------------------- y = bf16(rowscale * gamma * x)
__global__ void scale_cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ gamma,
                                  const float* __restrict__ rowscale, int rows_per_scale, int64_t M, int N) {
  const int N4 = N >> 2;
  const int64_t total = M * N4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t m = i / N4;
    const int c4 = (int)(i - m * N4);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float s = rowscale ? rowscale[m / rows_per_scale] : 1.f;
    if (gamma) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
      v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    }
    uint2 o;
    o.x = pack_op16x2(v.x * s, v.y * s);
    o.y = pack_op16x2(v.z * s, v.w * s);
    reinterpret_cast<uint2*>(y)[i] = o;
  }
}

// ---------------------------------------------------------------- sum of squares (deterministic: no atomics)
// Pass 1: every block leaves the sum of its grid-strided share in partial[block]; pass 2: one block adds the partials in a fixed order
// and accumulates into *out.  (Round 2 added the block sums with atomicAdd: the gradient norm then differed in its last bits from run
// to run, and with it the clipping coefficient and every parameter after the first clipped step.)
constexpr int SUMSQ_MAX_BLOCKS = 1024;
// 16 bytes per lane and load, four independent partial sums per lane, four loads in flight (round 5: the scalar one-accumulator loop read the
// 345 MB gradient buffer of ViT-B at 2.3 TB/s -- 148 us of every `half`-mode step).  `head` elements in front of the first 16-byte boundary
// and the n % 4 elements behind the last whole float4 go to block 0's lanes one by one; the order of additions is fixed by (n, alignment, grid).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t head = min<int64_t>(n, (int64_t)((16 - ((uintptr_t)x & 15)) & 15) / 4);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + head);
  const int64_t n4 = (n - head) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    s1 += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
    s2 += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    s3 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  for (; i < n4; i += stride) {
    const float4 a = x4[i];
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  if (blockIdx.x == 0) {
    const int64_t tail0 = head + (n4 << 2);
    if ((int64_t)threadIdx.x < head) s1 += x[threadIdx.x] * x[threadIdx.x];
    if (tail0 + threadIdx.x < n) s2 += x[tail0 + threadIdx.x] * x[tail0 + threadIdx.x];
  }
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// coef != null: instead of accumulating into *out, write out[0] = sqrt(sum) * inv_scale (the gradient norm with the loss scale removed),
// out[1] = inv_scale * min(1, max_norm / (out[0] + 1e-6)) (max_norm <= 0: inv_scale) or 0 when the norm is not finite, out[2] = 1 when it
// is not finite (tad_grad_norm_coef)
__global__ void sumsq_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out, int coef, float inv_scale,
                                    float max_norm) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    if (!coef) {
      *out += tot;
    } else {
      const float norm = sqrtf(tot) * inv_scale;
      const bool bad = !(fabsf(norm) < INFINITY);  // inf or NaN
      float c = inv_scale;
      if (max_norm > 0.f) c *= fminf(max_norm / (norm + 1e-6f), 1.f);
      out[0] = norm;
      out[1] = bad ? 0.f : c;
      out[2] = bad ? 1.f : 0.f;
    }
  }
}

// host-callable launcher shared with other translation units
// up to three outputs reduced from partial[q][splits][n] in one launch (n % 4 == 0)
int launch_reduce_cols(const float* partial, float* out0, float* out1, float* out2, int nq, int splits, int n, int accumulate,
                       hipStream_t st) {
  hipLaunchKernelGGL(reduce_cols_kernel, dim3((n / 4 + 15) / 16, nq), dim3(1024), 0, st, partial, out0, out1, out2, splits, n, accumulate,
                     (int64_t)splits * n, n);
  return check_launch("reduce_cols");
}

// two column ranges [c0, c0 + n) and [c1, c1 + n) of partial[splits][N] reduced into out0 / out1 in one launch (n, c0, c1, N % 4 == 0)
int launch_reduce_col_ranges(const float* partial, int N, int splits, int c0, float* out0, int c1, float* out1, int n, int accumulate,
                             hipStream_t st) {
  hipLaunchKernelGGL(reduce_cols_kernel, dim3((n / 4 + 15) / 16, 2), dim3(1024), 0, st, partial + c0, out0, out1, nullptr, splits, n, accumulate,
                     (int64_t)(c1 - c0), N);
  return check_launch("reduce_col_ranges");
}

// The two reductions that follow a dW GEMM in ONE launch: the split-K slabs (as reduce_partials_kernel) and, in extra blocks at the
// start of the grid, the bias column sums partial2[rows2][n2] -> out2a (columns [0, n2a)) and, if out2b, columns [c2b, c2b + n2a) ->
// out2b (the qkv Linear: q_bias / v_bias ranges), else all n2 columns -> out2a.
// outB != null (the pair launch of two weight gradients, launch_gemm_tn_pair): elements [0, nA) of a slab go to out, [nA, n) to outB.
__global__ void reduce_dw_kernel(const float* __restrict__ partial, float* __restrict__ out, int splits, int64_t n, int accumulate,
                                 int slab_blocks, const float* __restrict__ partial2, int rows2, int n2, float* out2a, float* out2b, int n2a,
                                 int c2b, float* __restrict__ outB, int64_t nA) {
  // the (few, latency-bound) bias blocks come FIRST in the grid so that they run beside the slab blocks, not after them
  const int bias_blocks = (int)gridDim.x - slab_blocks;
  if ((int)blockIdx.x >= bias_blocks) {
    const int64_t stride = (int64_t)slab_blocks * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)((int)blockIdx.x - bias_blocks) * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 s = reinterpret_cast<const float4*>(partial)[i];
      for (int k = 1; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(partial + (int64_t)k * n)[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float4* const dst4 = (outB && 4 * i >= nA) ? reinterpret_cast<float4*>(outB) + (i - (nA >> 2)) : reinterpret_cast<float4*>(out) + i;
      if (accumulate) {
        const float4 o = *dst4;
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      *dst4 = s;
    }
    return;
  }
  if (!partial2) return;  // (pair launch without bias sums: no bias blocks are launched)
  // bias job: one float4 column group per thread
  const int c4 = (int)blockIdx.x * blockDim.x + threadIdx.x;
  const int groups = n2a >> 2;
  const int total = out2b ? 2 * groups : groups;
  if (c4 >= total) return;
  const bool second = c4 >= groups;
  const int col = second ? c2b + 4 * (c4 - groups) : 4 * c4;
  float* dst = (second ? out2b : out2a) + (second ? 4 * (c4 - groups) : 4 * c4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = 0;
  for (; r + 8 <= rows2; r += 8) {  // 8 loads in flight (the rows are read in a fixed order: deterministic sums)
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(partial2 + (int64_t)(r + k) * n2 + col);
#pragma unroll
    for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  for (; r < rows2; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(partial2 + (int64_t)r * n2 + col);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (accumulate) {
    const float4 o = *reinterpret_cast<float4*>(dst);
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *reinterpret_cast<float4*>(dst) = s;
}

// n % 4 == 0, n2 % 4 == 0; out2b != null: two ranges of n2a columns each (the second starting at column c2b), else n2a = n2
int launch_reduce_dw(const float* partial, float* out, int splits, int64_t n, int accumulate, const float* partial2, int rows2, int n2,
                     float* out2a, float* out2b, int n2a, int c2b, hipStream_t st) {
  const int slab_blocks = capped_grid((n + 3) / 4, 256);
  const int groups = (out2b ? 2 : 1) * (n2a / 4);
  const int bias_blocks = (groups + 255) / 256;
  hipLaunchKernelGGL(reduce_dw_kernel, dim3(slab_blocks + bias_blocks), dim3(256), 0, st, partial, out, splits, n, accumulate, slab_blocks,
                     partial2, rows2, n2, out2a, out2b, n2a, c2b, (float*)nullptr, n);
  return check_launch("reduce_dw");
}

// the same over slabs that hold TWO outputs: elements [0, nA) -> outA, [nA, n) -> outB (nA % 4 == 0); partial2 == null: no bias sums
int launch_reduce_dw_pair(const float* partial, float* outA, float* outB, int64_t nA, int splits, int64_t n, int accumulate, const float* partial2,
                          int rows2, int n2, float* out2a, float* out2b, int n2a, int c2b, hipStream_t st) {
  const int slab_blocks = capped_grid((n + 3) / 4, 256);
  const int groups = partial2 ? (out2b ? 2 : 1) * (n2a / 4) : 0;
  const int bias_blocks = (groups + 255) / 256;
  hipLaunchKernelGGL(reduce_dw_kernel, dim3(slab_blocks + bias_blocks), dim3(256), 0, st, partial, outA, splits, n, accumulate, slab_blocks,
                     partial2, rows2, n2, out2a, out2b, n2a, c2b, outB, nA);
  return check_launch("reduce_dw_pair");
}

int launch_reduce_partials(const float* partial, float* out, int splits, int64_t n, int accumulate, hipStream_t st) {
  if (n <= 16384 && splits >= 16 && (n & 3) == 0) return launch_reduce_cols(partial, out, nullptr, nullptr, 1, splits, (int)n, accumulate, st);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(capped_grid((n + 3) / 4, 256)), dim3(256), 0, st, partial, out, splits, n,
                     accumulate);
  return check_launch("reduce_partials");
}

TAD_NAMESPACE_END

TAD_NAMESPACE_BEGIN
// precise-mode (f32) patch matrix for even patch sizes that are not a multiple of 4; called from precise.hip
int launch_im2col_pairs_f32(const float* x, float* cols, int B, int C, int T, int H, int W, int tubelet, int patch, int ldk, hipStream_t st) {
  const int64_t pairs = (int64_t)B * C * T * H * (W / 2);
  hipLaunchKernelGGL(im2col_tubelets_pairs_kernel<float>, dim3(capped_grid(pairs, 256)), dim3(256), 0, st, x, cols, B, C, T, H, W, tubelet,
                     patch, ldk);
  return check_launch("im2col_f32");
}
// ---------------------------------------------------------------- operand split for the precise / split-operand Linears
// x = hi + lo with hi = op16(x), lo = op16(x - hi) (bf16: 16 significant bits, f16: 22).  A product of two f32 operands is recovered
// to ~2^-17 (bf16) relative from three MFMA products  A_hi*B_hi + A_hi*B_lo + A_lo*B_hi  -- and those three products are ONE GEMM
// over a 3x longer reduction dimension: [A_hi | A_hi | A_lo] . [B_hi | B_lo | B_hi]^T.  So the split Linear reuses the production
// MFMA GEMM kernels unchanged; only the operand preparation differs.
// out row r, for source row m = r % M (stack mode) or r (concat mode)
// concat (along K): out [M, 3K]; role A: [hi | hi | lo], role B: [hi | lo | hi]
// stack  (along M): out [3M, K]; role A: rows [hi ; hi ; lo], role B: rows [hi ; lo ; hi]
__global__ void split_bf16x3_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int64_t M, int K, int role_b, int stack) {
  const int K4 = K >> 2;
  const int64_t total = M * K4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t m = i / K4;
    const int c = (int)(i - m * K4) * 4;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = f32_to_op16(f[e]);
      lo[e] = f32_to_op16(f[e] - op16_to_f32(hi[e]));
    }
    const uint2 H = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
    const uint2 L = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    const uint2 s0 = H, s1 = role_b ? L : H, s2 = role_b ? H : L;
    if (stack) {
      *reinterpret_cast<uint2*>(out + (m)*K + c) = s0;
      *reinterpret_cast<uint2*>(out + (M + m) * K + c) = s1;
      *reinterpret_cast<uint2*>(out + (2 * M + m) * K + c) = s2;
    } else {
      uint16_t* o = out + m * 3 * K;
      *reinterpret_cast<uint2*>(o + c) = s0;
      *reinterpret_cast<uint2*>(o + K + c) = s1;
      *reinterpret_cast<uint2*>(o + 2 * K + c) = s2;
    }
  }
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, tad_stream_t stream) {
  TAD_REQUIRE(src && dst && n >= 0, "cast: null pointer");
  if (n == 0) return TAD_OK;
  TAD_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, "cast: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(capped_grid((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  return check_launch("cast_f32_bf16");
}

int tad_transpose_cast_f32_bf16(const float* src, uint16_t* dst, int R, int C, tad_stream_t stream) {
  TAD_REQUIRE(src && dst && R > 0 && C > 0, "transpose_cast: bad args");
  hipLaunchKernelGGL(transpose_cast_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, dst, R, C);
  return check_launch("transpose_cast");
}

#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
int tad_transpose_bf16_batched(const uint16_t* src, uint16_t* dst, const int32_t* table, int n_tiles, tad_stream_t stream) {
  TAD_REQUIRE(src && dst && table && n_tiles > 0, "transpose_bf16_batched: bad args");
  TAD_REQUIRE((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)table) & 15) == 0, "transpose_bf16_batched: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, src, dst,
                     reinterpret_cast<const int4*>(table));
  return check_launch("transpose_bf16_batched");
}

int tad_patch_embed_ldk(int C, int tubelet, int patch) {
  const int K = C * tubelet * patch * patch;
  return (K + 63) / 64 * 64;
}
#endif

int tad_im2col_tubelets(const float* x, uint16_t* cols, int B, int C, int T, int H, int W, int tubelet, int patch,
                        tad_stream_t stream) {
  TAD_REQUIRE(x && cols, "im2col: null pointer");
  TAD_REQUIRE(B > 0 && C > 0 && tubelet > 0 && patch > 0 && T % tubelet == 0 && H % patch == 0 && W % patch == 0,
              "im2col: T/H/W must be multiples of tubelet/patch (got T=%d H=%d W=%d tub=%d p=%d)", T, H, W, tubelet, patch);
  TAD_REQUIRE(patch % 2 == 0, "im2col: patch size must be even (got %d)", patch);
  if (patch % 8) {  // row stride tad_patch_embed_ldk(): see im2col_tubelets_pairs_kernel
    const int64_t pairs = (int64_t)B * C * T * H * (W / 2);
    hipLaunchKernelGGL(im2col_tubelets_pairs_kernel<uint16_t>, dim3(capped_grid(pairs, 256)), dim3(256), 0, (hipStream_t)stream, x, cols, B,
                       C, T, H, W, tubelet, patch, tad_patch_embed_ldk(C, tubelet, patch));
    return check_launch("im2col_tubelets");
  }
  const int64_t total = (int64_t)B * C * T * H * (W / 8);
  hipLaunchKernelGGL(im2col_tubelets_kernel, dim3(capped_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cols, B, C, T,
                     H, W, tubelet, patch);
  return check_launch("im2col_tubelets");
}

int tad_im2col_tubelets_u8(const uint8_t* frames, uint16_t* cols, int B, int T, int H, int W, int tubelet, int patch, const float* mean3,
                           const float* std3, int bgr, int t_offset, tad_stream_t stream) {
  TAD_REQUIRE(frames && cols && mean3 && std3, "im2col_u8: null pointer");
  TAD_REQUIRE(B > 0 && tubelet > 0 && patch > 0 && T % tubelet == 0 && H % patch == 0 && W % patch == 0,
              "im2col_u8: T/H/W must be multiples of tubelet/patch (got T=%d H=%d W=%d tub=%d p=%d)", T, H, W, tubelet, patch);
  TAD_REQUIRE(patch % 2 == 0, "im2col_u8: patch size must be even (got %d)", patch);
  TAD_REQUIRE(t_offset >= 0 && t_offset < T, "im2col_u8: t_offset=%d outside [0, %d)", t_offset, T);
  TAD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "im2col_u8: zero std");
  TAD_REQUIRE((((uintptr_t)frames) & 3) == 0 && (((uintptr_t)cols) & 15) == 0, "im2col_u8: misaligned buffers");
  if (patch % 8) {  // rows of stride tad_patch_embed_ldk, zero-padded (ViT-L/14: K = 1176 -> 1216)
    const int ldk = tad_patch_embed_ldk(3, tubelet, patch);
    const int64_t pairs = (int64_t)B * T * H * (W / 2);
    hipLaunchKernelGGL(im2col_tubelets_u8_pairs_kernel, dim3(capped_grid(pairs, 256)), dim3(256), 0, (hipStream_t)stream, frames, cols, B, T,
                       H, W, tubelet, patch, ldk, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], bgr ? 1 : 0, t_offset);
    return check_launch("im2col_tubelets_u8");
  }
  const int64_t total = (int64_t)B * T * H * (W / 8);
  hipLaunchKernelGGL(im2col_tubelets_u8_kernel, dim3(capped_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, frames, cols, B, T, H,
                     W, tubelet, patch, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], bgr ? 1 : 0, t_offset);
  return check_launch("im2col_tubelets_u8");
}

#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
int tad_meanpool_fwd(const float* x, float* y, float* ws, int B, int N, int D, tad_stream_t stream) {
  TAD_REQUIRE(x && y && ws && B > 0 && N > 0 && D > 0 && D % 4 == 0, "meanpool_fwd: bad args (D must be a multiple of 4)");
  hipLaunchKernelGGL(meanpool_partial_kernel, dim3((D / 4 + 63) / 64, TAD_POOL_SPLIT, B), dim3(256), 0, (hipStream_t)stream, x, ws, N, D);
  hipLaunchKernelGGL(meanpool_final_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, y, B, N, D);
  return check_launch("meanpool_fwd");
}
#endif

int tad_meanpool_bwd(const float* dy, float* dx, uint16_t* dx_bf16, int B, int N, int D, tad_stream_t stream) {
  TAD_REQUIRE(dy && (dx || dx_bf16) && B > 0 && N > 0 && D % 4 == 0, "meanpool_bwd: bad args");
  hipLaunchKernelGGL(meanpool_bwd_kernel, dim3(capped_grid((int64_t)B * N * D / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     dx_bf16, B, N, D);
  return check_launch("meanpool_bwd");
}

static inline int colsum_splits(int64_t M) {
  int64_t s = (M + 255) / 256;
  if (s > 128) s = 128;
  if (s < 1) s = 1;
  return (int)s;
}
#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
size_t tad_colsum_workspace_bytes(int64_t M, int N) { return (size_t)colsum_splits(M) * (size_t)N * sizeof(float); }
#endif

int tad_colsum_bf16(const uint16_t* a, float* out, int accumulate, void* ws, size_t ws_bytes, int64_t M, int N,
                    tad_stream_t stream) {
  TAD_REQUIRE(a && out && ws && M > 0 && N > 0 && N % 8 == 0, "colsum: bad args (N must be a multiple of 8)");
  const int splits = colsum_splits(M);
  if (ws_bytes < (size_t)splits * N * sizeof(float)) { set_error("colsum: workspace too small"); return TAD_ENOSPACE; }
  const int rows_per = (int)((M + splits - 1) / splits);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 511) / 512, splits), dim3(256), 0, (hipStream_t)stream, a, (float*)ws, M, N, rows_per);
  return launch_reduce_partials((const float*)ws, out, splits, N, accumulate, (hipStream_t)stream);
}

#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
int tad_colsum_window_f32(const float* a, float* out, int accumulate, void* ws, size_t ws_bytes, int B, int R, int N, int r0, int rc,
                          tad_stream_t stream) {
  TAD_REQUIRE(a && out && ws && B > 0 && R > 0 && N > 0 && N % 4 == 0, "colsum_window: bad args (N must be a multiple of 4)");
  TAD_REQUIRE(r0 >= 0 && rc > 0 && r0 + rc <= R, "colsum_window: rows [%d, %d) outside [0, %d)", r0, r0 + rc, R);
  const int64_t total = (int64_t)B * rc;
  const int splits = colsum_splits(total);
  if (ws_bytes < (size_t)splits * N * sizeof(float)) { set_error("colsum_window: workspace too small"); return TAD_ENOSPACE; }
  const int rows_per = (int)((total + splits - 1) / splits);
  hipLaunchKernelGGL(colsum_window_f32_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, (hipStream_t)stream, a, (float*)ws, B, R, N, r0,
                     rc, rows_per);
  int rc_ = check_launch("colsum_window");
  if (rc_) return rc_;
  return launch_reduce_partials((const float*)ws, out, splits, N, accumulate, (hipStream_t)stream);
}
#endif

int tad_scale_cast_bf16(const float* x, uint16_t* y, const float* gamma, const float* rowscale, int rows_per_scale, int64_t M,
                        int N, tad_stream_t stream) {
  TAD_REQUIRE(x && y && M > 0 && N > 0 && N % 4 == 0, "scale_cast: bad args");
  TAD_REQUIRE(!rowscale || rows_per_scale > 0, "scale_cast: rows_per_scale must be > 0");
  hipLaunchKernelGGL(scale_cast_kernel, dim3(capped_grid(M * N / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, gamma, rowscale,
                     rows_per_scale, M, N);
  return check_launch("scale_cast");
}

#ifndef TAD_OPND_F16  // format-independent: one copy for the library (bf16 pass)
size_t tad_sumsq_workspace_bytes(void) { return (size_t)SUMSQ_MAX_BLOCKS * sizeof(float); }

int tad_sumsq_f32(const float* x, int64_t n, float* out, void* ws, size_t ws_bytes, tad_stream_t stream) {
  TAD_REQUIRE(x && out && ws && n >= 0, "sumsq: bad args");
  TAD_REQUIRE(ws_bytes >= tad_sumsq_workspace_bytes(), "sumsq: workspace too small");
  if (n == 0) return TAD_OK;
  int blocks = capped_grid(n, 256 * 16);
  if (blocks > SUMSQ_MAX_BLOCKS) blocks = SUMSQ_MAX_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, (float*)ws);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, blocks, out, 0, 1.f, 0.f);
  return check_launch("sumsq");
}

int tad_grad_norm_coef(const float* x, int64_t n, float inv_scale, float max_norm, float* out3, void* ws, size_t ws_bytes, tad_stream_t stream) {
  TAD_REQUIRE(x && out3 && ws && n > 0, "grad_norm_coef: bad args");
  TAD_REQUIRE(ws_bytes >= tad_sumsq_workspace_bytes(), "grad_norm_coef: workspace too small");
  int blocks = capped_grid(n, 256 * 16);
  if (blocks > SUMSQ_MAX_BLOCKS) blocks = SUMSQ_MAX_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, (float*)ws);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, blocks, out3, 1, inv_scale, max_norm);
  return check_launch("grad_norm_coef");
}
#endif

int tad_split_bf16x3(const float* x, uint16_t* out, int64_t M, int K, int role_b, int stack, tad_stream_t stream) {
  TAD_REQUIRE(x && out && M > 0 && K > 0 && K % 4 == 0, "split_bf16x3: bad args (K must be a multiple of 4)");
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(capped_grid(M * (K / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, out, M, K, role_b, stack);
  return check_launch("split_bf16x3");
}

}  // extern "C"
