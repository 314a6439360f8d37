// Fused space-time attention backward (recompute, no N x N tensor), head_dim 64, gfx950.
//
// Two kernels, both deterministic (no atomics):
//   1. attn_bwd_dq_kernel: one workgroup = 128 query rows, loops over 64-key tiles (structure of the forward kernel).  Its
//        prologue computes delta[b,h,q] = sum_d dO[q,d] * O[q,d] for its own rows (the dO fragments are in registers anyway) and
//        publishes -delta and -lse/scale for kernel 2 -- a separate delta pass was one more launch and one more read of dO per layer.
//        S^T = K Q^T, dP^T = V dO^T (query on the lane -> lse/delta are per-lane scalars),
//        dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T (dS^T accumulator registers are the MFMA B operand).
//   2. attn_bwd_dkv_kernel: one workgroup = 128 keys (32 per wave, K/V fragments pinned in registers), loops over
//        64-row query tiles: S = Q K^T and dP = dO V^T with the key on the lane and (-lse/scale, -delta) -- staged by LDS-DMA next
//        to the Q / dO tiles -- preloaded as the initial accumulators, then dV^T += dO^T P and dK^T += Q^T dS with P / dS taken
//        straight from the accumulator registers.
// Summing dQ across key blocks would need ~1 GB of f32 atomics per layer at N=1568 (0.8 ms at the chip's 1.3 TB/s
// atomic rate, more than the whole MFMA work), so dQ gets its own pass that recomputes S and dP (7 instead of 5
// MFMA products, but no cross-workgroup reduction and bitwise-reproducible results).
//
// Tiles that are read both by rows (ds_read_b128) and transposed (ds_read_b64_tr_b16) use one LDS image with a
// swizzle that is conflict-free for both (found by tools/lds_bank_sim.py).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.h"

TAD_NAMESPACE_BEGIN

constexpr int BHD = 64;
constexpr float LOG2E = 1.44269504088896340736f;

// 16-byte chunk swizzle for 128-byte rows, conflict-free for row reads (32 consecutive rows, chunk 2ks+h) and for
// transposed reads (4 consecutive rows x 64 B)
__device__ __forceinline__ int sw_dual(int row) {
  return ((row >> 1) & 1) | (((row >> 2) & 1) << 1) | ((((row >> 1) ^ (row >> 3)) & 1) << 2);
}

typedef __attribute__((ext_vector_type(8))) short s16x8_t;

// transposed fragment: lane (x = lane&31 within a 32-wide column tile, h = lane>>5) gets, for j = 0..7, element
// tile[rbase + 8*(j>>2) + 4*h + (j&3)][col0 + x]   (the k-order of an accumulator tile used as the other operand)
__device__ __forceinline__ op16x8 tr_frag_dual(const char* tile, int rbase, int col0, int lane) {
  const int G = lane >> 4, li = lane & 15;
  const int r0 = rbase + 4 * (G >> 1) + (li >> 2), r1 = r0 + 8;
  const int col = col0 + 16 * (G & 1) + 4 * (li & 3);
  const int ch = col >> 3, sub = (col & 7) * 2;
  const char* a0 = tile + r0 * 128 + ((ch ^ sw_dual(r0)) << 4) + sub;
  const char* a1 = tile + r1 * 128 + ((ch ^ sw_dual(r1)) << 4) + sub;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a1));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(op16x8, v);
}
// The same fragment through the inline-asm reads of common.h (the builtin form above makes the compiler drain the LDS-DMA of the
// next tile in front of it): tr_dual_addr gives the lane's two LDS addresses for rbase = 0 inside a tile at byte address `tile`
// (the second read's swizzle differs from the first's, so it has its own base); rbase (a multiple of 16 rows: the swizzle only
// looks at row bits 1..3) and the tile's position go into the instruction's immediate.
__device__ __forceinline__ void tr_dual_addr(uint32_t tile, int col0, int lane, uint32_t (&a)[2]) {
  const int G = lane >> 4, li = lane & 15;
  const int col = col0 + 16 * (G & 1) + 4 * (li & 3);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 4 * (G >> 1) + (li >> 2) + 8 * h;
    a[h] = tile + (uint32_t)(row * 128 + (((col >> 3) ^ sw_dual(row)) << 4) + (col & 7) * 2);
  }
}
// row fragment: lane (row = lane&31, h = lane>>5) gets tile[row0 + row][16ks + 8h .. +7]
__device__ __forceinline__ op16x8 row_frag_dual(const char* tile, int row, int ks, int h5) {
  return *reinterpret_cast<const op16x8*>(tile + row * 128 + (((2 * ks + h5) ^ sw_dual(row)) << 4));
}

__device__ __forceinline__ op16x8 pack8(const f32x16& a, int s2) {
  // pairwise v_cvt_pk_bf16_f32 (one instruction per two elements)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = pack_op16x2(a[8 * s2 + 2 * j], a[8 * s2 + 2 * j + 1]);
  return __builtin_bit_cast(op16x8, r);
}

// ------------------------------------------------------------------------------------------------ dQ
// DMA_MODE: 0 = production (the LDS-DMA pieces of the NEXT tile are issued at the top of a tile, before its first matrix instruction; the end of
// the tile waits for all of it).  2 / 3 exist in ablation builds only (timing experiments, wrong results): 2 = no DMA inside the loop,
// 3 = dK/dV kernel without its transposed LDS reads.  The variants round 2 measured and dropped (pieces spread into the tile, a
// three-deep tile ring, 64 keys per wave at one wave per SIMD) are archived under experiments/r02_variants.
template <int DMA_MODE>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                          const uint16_t* __restrict__ out_lo,
                                                          const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, uint16_t* __restrict__ dqkv, int N, int H, int B,
                                                          float scale, int lse_log2) {
  constexpr int TILE_BYTES = 64 * 128;
  constexpr int NST = 2;  // K/V (Q/dO) tile ring depth
  __shared__ __attribute__((aligned(1024))) char lds[NST * 2 * TILE_BYTES];  // [buf][K|V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N + 127) / 128;  // 1-D XCD-aware grid: the blocks of one (batch, head) pair share an L2 (see attn_fwd.hip)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int head = (lin / nblk) % H, b = lin / nblk / H;
  const int q0 = (lin % nblk) * 128 + wave * 32;
  const int ql = lane & 31, h5 = lane >> 5;
  const int64_t tok = (int64_t)3 * H * BHD;
  const uint16_t* base = qkv + (int64_t)b * N * tok + head * BHD;
  const uint16_t* kbase = base + (int64_t)H * BHD;
  const uint16_t* vbase = base + (int64_t)2 * H * BHD;
  const float c = scale * LOG2E;

  int qrow = q0 + ql;
  const bool qvalid = qrow < N;
  const bool wave_live = q0 < N;  // wave-uniform
  if (!qvalid) qrow = N - 1;
  op16x8 qf[4], dof[4];
  {
    const uint16_t* qp = base + (int64_t)qrow * tok + 8 * h5;
    const uint16_t* dp = dout + (((int64_t)b * N + qrow) * H + head) * BHD + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *reinterpret_cast<const op16x8*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const op16x8*>(dp + 16 * ks);
    }
  }
  const float lse2 = lse[((int64_t)b * H + head) * N + qrow] * LOG2E;
  // delta = rowsum(dO o O): this lane holds half of its query row (the 8-element groups 2ks + h5), lane ^ 32 the other half
  float dlt;
  {
    // With out_lo (what the forward's 16-bit rounding of O dropped) delta is taken of the UNROUNDED output.  delta = sum_k P_k dP_k
    // must cancel against the dP the kernels recompute exactly from V and dO; a delta computed from the rounded O is off by
    // dO . (O - round(O)), an error in dS proportional to P that does not cancel -- for rows whose dP_k are nearly equal across the
    // keys (dS small against delta: near-uniform attention, a common component in V) it dominated dQ / dK and the q / k gradients
    // behind them (measured on ViT-B at the real shape: the worst q-row slice of a qkv weight gradient off by 12 % in bf16, 2 % in f16).
    const int64_t orow = (((int64_t)b * N + qrow) * H + head) * BHD + 8 * h5;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const op16x8 of = *reinterpret_cast<const op16x8*>(out + orow + 16 * ks);
      if (out_lo) {
        const op16x8 ol = *reinterpret_cast<const op16x8*>(out_lo + orow + 16 * ks);
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)of[e] + (float)ol[e], (float)dof[ks][e], part);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) part = fmaf((float)of[e], (float)dof[ks][e], part);
      }
    }
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
    dlt = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    // published for the dK/dV kernel as the initial values of its accumulators: -delta, and -lse/scale (so that
    // exp2((q.k - lse/scale) * scale*log2e) = exp(q.k*scale - lse))
    if (qvalid && h5 == 0) {
      const int64_t idx = ((int64_t)b * H + head) * N + qrow;
      delta[idx] = -dlt;
      // (attn_bwd_dkv_kernel adds it to q.k before the scale; the pipelined kernel adds -lse * log2(e) after it)
      delta[(int64_t)B * H * N + idx] = lse_log2 ? -lse2 : -lse[idx] / scale;
    }
  }

  // K/V tiles go global -> LDS by LDS-DMA (see attn_fwd.hip): 1-KiB piece = 8 keys x 128 B, wave w moves pieces w and w+4 of K and
  // of V; the swizzle is applied to the per-lane SOURCE chunk.  Reads past the tensor return zero; keys >= N are masked below.
  const uint32_t qkv_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)tok * 2u;
  const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  uint32_t dma_k[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int key = (wave + 4 * i) * 8 + (lane >> 3);
    dma_k[i] = (uint32_t)(((int64_t)b * N + key) * tok * 2) + (uint32_t)((head + H) * BHD * 2) + (uint32_t)(((lane & 7) ^ sw_dual(key)) << 4);
  }
  const uint32_t v_off = (uint32_t)(H * BHD * 2), key_step = (uint32_t)(tok * 2);
#define DMA_K_(buf, kv0)                                                                                                        \
  {                                                                                                                             \
    char* kl_ = lds + (buf) * 2 * TILE_BYTES;                                                                                   \
    const uint32_t adv_ = (uint32_t)(kv0) * key_step;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + (wave + 4 * i) * 1024), 16, dma_k[i] + adv_, 0, 0, 0);     \
  }
#define DMA_V_(buf, kv0)                                                                                                        \
  {                                                                                                                             \
    char* kl_ = lds + (buf) * 2 * TILE_BYTES;                                                                                   \
    const uint32_t adv_ = (uint32_t)(kv0) * key_step;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + TILE_BYTES + (wave + 4 * i) * 1024), 16, dma_k[i] + v_off + adv_, 0, 0, 0); \
  }
#define DMA_KV(buf, kv0) { DMA_K_(buf, kv0); DMA_V_(buf, kv0); }

  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

  const int nt = (N + 63) / 64;
  DMA_KV(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // the tile loop runs in pairs so that the LDS buffer index is a literal in each copy of the body: every LDS address is then a
  // lane constant + immediate instead of a handful of v_add / v_or per fragment read
  uint32_t ktr[2][2];  // K^T fragment addresses (tile 0 of buffer 0), [d tile][first / second read]
  tr_dual_addr(lds_addr(lds), 0, lane, ktr[0]);
  tr_dual_addr(lds_addr(lds), 32, lane, ktr[1]);
  auto dq_tile = [&](auto BUFC, int t) {
    constexpr int BUF = decltype(BUFC)::value;
    const int kv0 = t * 64;
    const bool more = t + 1 < nt;  // is there a tile to request during this one?
    constexpr int NBUF = BUF ^ 1;  // its buffer ...
    const int nkv0 = kv0 + 64;     // ... and first key
    if (more && DMA_MODE == 0) DMA_KV(NBUF, nkv0);
    const char* kl = lds + BUF * 2 * TILE_BYTES;
    const char* vl = kl + TILE_BYTES;
    // a wave whose 32 query rows all lie past the sequence (the last block of N = 1568 has one live wave of four) only helps
    // staging the tiles: its matrix / VALU slots go to the other waves on its SIMD
    if (wave_live)
    static_for<0, 2>([&](auto ktc) {
      constexpr int kt = decltype(ktc)::value;
      if (kv0 + 32 * kt >= N) return;  // a half tile past the sequence (N = 1568: the second half of the last tile) contributes nothing
      // K^T fragments for the dQ product, issued now and waited for after the exponentials (asm reads: see common.h)
      s16x4 tl[2][2], th[2][2];  // [s2][dt]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          tl[s2][dt] = s2 == 0 ? lds_tr16_b64<BUF * 2 * TILE_BYTES + (kt * 32) * 128>(ktr[dt][0])
                               : lds_tr16_b64<BUF * 2 * TILE_BYTES + (kt * 32 + 16) * 128>(ktr[dt][0]);
          th[s2][dt] = s2 == 0 ? lds_tr16_b64<BUF * 2 * TILE_BYTES + (kt * 32) * 128>(ktr[dt][1])
                               : lds_tr16_b64<BUF * 2 * TILE_BYTES + (kt * 32 + 16) * 128>(ktr[dt][1]);
        }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = -dlt; }
      const int key = kt * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = TAD_MFMA_32x32x16(row_frag_dual(kl, key, ks, h5), qf[ks], s);
        dp = TAD_MFMA_32x32x16(row_frag_dual(vl, key, ks, h5), dof[ks], dp);
      }
      // dS^T = P^T o (dP^T - delta); keys >= N contribute nothing
      if (kv0 + 64 > N) {  // ragged last tile only: keys >= N get P = 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kg = kv0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h5;
          if (kg >= N) s[r] = -1e30f;
        }
      }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(s[r] * c - lse2) * dp[r];
      lds_wait<4>(tl[0][0], th[0][0], tl[0][1], th[0][1]);
      {
        const op16x8 dsf = pack8(ds, 0);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = TAD_MFMA_32x32x16(join_tr(tl[0][dt], th[0][dt]), dsf, dq[dt]);
      }
      lds_wait<0>(tl[1][0], th[1][0], tl[1][1], th[1][1]);
      {
        const op16x8 dsf = pack8(ds, 1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[dt] = TAD_MFMA_32x32x16(join_tr(tl[1][dt], th[1][dt]), dsf, dq[dt]);
      }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int t = 0; t < nt; t += 2) {
    dq_tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) dq_tile(std::integral_constant<int, 1>{}, t + 1);
  }

  if (qvalid) {
    uint16_t* op = dqkv + ((int64_t)b * N + qrow) * tok + head * BHD;  // q slot (index 0 of the "3" axis)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = dt * 32 + 8 * r4 + 4 * h5;
        uint2 pk;
        pk.x = pack_op16x2(dq[dt][4 * r4 + 0] * scale, dq[dt][4 * r4 + 1] * scale);
        pk.y = pack_op16x2(dq[dt][4 * r4 + 2] * scale, dq[dt][4 * r4 + 3] * scale);
        *reinterpret_cast<uint2*>(op + d) = pk;
      }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int DMA_MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                           const float* __restrict__ rowc_g, uint16_t* __restrict__ dqkv, int N, int H, int B,
                                                           float scale, unsigned long long* stamps) {
  constexpr int TILE_BYTES = 64 * 128;
  constexpr int STAGE = 2 * TILE_BYTES + 512;  // Q tile, dO tile, 64 x (-lse/scale), 64 x (-delta)
  constexpr int NST = 2;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N + 127) / 128;  // 1-D XCD-aware grid (see attn_fwd.hip)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int head = (lin / nblk) % H, b = lin / nblk / H;
  const int key0 = (lin % nblk) * 128 + wave * 32;
  const int kl_ = lane & 31, h5 = lane >> 5;
  const int64_t tok = (int64_t)3 * H * BHD;
  const uint16_t* base = qkv + (int64_t)b * N * tok + head * BHD;
  const uint16_t* kbase = base + (int64_t)H * BHD;
  const uint16_t* vbase = base + (int64_t)2 * H * BHD;
  const uint16_t* dobase = dout + ((int64_t)b * N * H + head) * BHD;  // row q at + q*H*64
  const float c = scale * LOG2E;

  int krow = key0 + kl_;
  const bool kvalid = krow < N;
  const bool wave_live = key0 < N;  // wave-uniform
  if (!kvalid) krow = N - 1;
  op16x8 kfr[4], vfr[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kfr[ks] = *reinterpret_cast<const op16x8*>(kbase + (int64_t)krow * tok + 16 * ks + 8 * h5);
    vfr[ks] = *reinterpret_cast<const op16x8*>(vbase + (int64_t)krow * tok + 16 * ks + 8 * h5);
  }

  // Q / dO tiles go global -> LDS by LDS-DMA (1-KiB piece = 8 rows x 128 B, wave w moves pieces w and w+4 of each; swizzle on the
  // per-lane SOURCE chunk); rows past the tensor read as zero, rows >= N are neutralised through the row constants below.
  const uint32_t qkv_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)tok * 2u;
  const uint32_t do_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)(H * BHD) * 2u;
  const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const auto rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dout), 0, (int)do_bytes, 0x00020000);
  uint32_t dma_q[2], dma_do[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 4 * i) * 8 + (lane >> 3);
    const uint32_t ch = (uint32_t)(((lane & 7) ^ sw_dual(row)) << 4);
    dma_q[i] = (uint32_t)(((int64_t)b * N + row) * tok * 2) + (uint32_t)(head * BHD * 2) + ch;
    dma_do[i] = (uint32_t)((((int64_t)b * N + row) * H + head) * BHD * 2) + ch;
  }
  const uint32_t q_step = (uint32_t)(tok * 2), do_step = (uint32_t)(H * BHD * 2);
  // Row constants (initial accumulator values, written by the dQ kernel): rows [0, BHN) of `rowc` hold -delta, rows [BHN, 2 BHN)
  // hold -lse/scale.  They are staged by LDS-DMA as well (4 bytes per lane: wave 0 moves the 64 -lse/scale values of the tile,
  // wave 1 the 64 -delta values), so the tile loop holds no ordinary global load and no LDS store -- with either of them in the loop
  // the compiler drained the DMA of the next tile (s_waitcnt vmcnt(0)) right after issuing it.
  const int64_t bhn = (int64_t)B * H * N;
  const auto rs_rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowc_g), 0, (int)(2 * bhn * 4), 0x00020000);
  const uint32_t rc_off = (uint32_t)(((wave == 0 ? bhn : 0) + ((int64_t)b * H + head) * N + lane) * 4);
#define LOAD_Q_(buf, q0)                                                                                   \
  {                                                                                                        \
    char* ql_ = lds + (buf) * STAGE;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(ql_ + (wave + 4 * i) * 1024), 16, dma_q[i] + (uint32_t)(q0) * q_step, 0, 0, 0); \
  }
#define LOAD_DO_RC_(buf, q0)                                                                               \
  {                                                                                                        \
    char* ql_ = lds + (buf) * STAGE;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, LDS_PTR(ql_ + TILE_BYTES + (wave + 4 * i) * 1024), 16, dma_do[i] + (uint32_t)(q0) * do_step, 0, 0, 0); \
    if (wave < 2) /* wave-uniform */                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rc, LDS_PTR(ql_ + 2 * TILE_BYTES + wave * 256), 4, rc_off + (uint32_t)(q0) * 4u, 0, 0, 0); \
  }
#define LOAD_QDO(buf, q0) { LOAD_Q_(buf, q0); LOAD_DO_RC_(buf, q0); }

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

  // Every LDS read of the tile loop is inline asm (common.h): the compiler can then neither drain the DMA of the next tile in front
  // of a read nor serialise read -> wait -> MFMA one fragment at a time; the reads of a half tile are issued in three batches and
  // waited for where their consumers start.  Lane-constant addresses (stage 0); the stage offset is added per tile, the half
  // tile / fragment position is an instruction immediate.
  const uint32_t lds0 = lds_addr(lds);
  uint32_t qtr[2][2];  // transposed fragments of the Q tile, [d tile][first / second read]; dO tile: + TILE_BYTES
  tr_dual_addr(lds0, 0, lane, qtr[0]);
  tr_dual_addr(lds0, 32, lane, qtr[1]);
  uint32_t rfa[4];     // row fragments (row lane&31 of a half tile, chunk 2ks + h5) of the Q tile; dO tile: + TILE_BYTES
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rfa[ks] = lds0 + (uint32_t)(kl_ * 128 + (((2 * ks + h5) ^ sw_dual(kl_)) << 4));
  const uint32_t rca = lds0 + 2 * TILE_BYTES + 16 * h5;  // row constants: 4 floats at [8 r4 + 4 h5]

  const int nt = (N + 63) / 64;
  LOAD_QDO(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef TAD_GEMM_ABLATION  // diagnostic builds only (tad_attn_debug_stamps): shader clock / 100 MHz clock around the tile loop
  if (stamps && tid == 0) {
    stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
  }
#endif
  int cur = 0;  // ring slot of tile t
  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;  // is there a tile to request during this one?
    const int nbuf = cur ^ 1;      // its ring slot ...
    const int nq0 = (t + 1) * 64;  // ... and first query row
    if (more && DMA_MODE == 0) LOAD_QDO(nbuf, nq0);
    const uint32_t so = (uint32_t)(cur * STAGE);
    if (wave_live)  // (see the dQ kernel: waves whose 32 keys all lie past the sequence only stage tiles)
    static_for<0, 2>([&](auto qtc) {
      constexpr int qt = decltype(qtc)::value;
      constexpr int HT = qt * 32 * 128;  // byte offset of the half tile inside a tile
      if (t * 64 + 32 * qt >= N) return;  // half tile of query rows past the sequence: P = dS = 0 there anyway
      // batch 1: initial accumulators (per-row constants; accumulator register r <-> row (r&3) + 8*(r>>2) + 4*h5) and row fragments
      f32x4 si[4], di[4];
      op16x8 qa[4], da[4];
      static_for<0, 4>([&](auto r4c) {
        constexpr int r4 = decltype(r4c)::value;
        si[r4] = lds_read_b128<f32x4, (qt * 32 + 8 * r4) * 4>(rca + so);
        di[r4] = lds_read_b128<f32x4, 256 + (qt * 32 + 8 * r4) * 4>(rca + so);
      });
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qa[ks] = lds_read_b128<op16x8, HT>(rfa[ks] + so);
        da[ks] = lds_read_b128<op16x8, TILE_BYTES + HT>(rfa[ks] + so);
      }
      // batch 2 / 3: transposed fragments for the dV / dK products of rows 0..15 / 16..31 of the half tile
      s16x4 dol[2][2], doh[2][2], qtl[2][2], qth[2][2];  // [s2][dt]
#define TR_ISSUE(s2_)                                                                         \
  _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                          \
    dol[s2_][dt] = lds_tr16_b64<TILE_BYTES + HT + 16 * (s2_) * 128>(qtr[dt][0] + so);         \
    doh[s2_][dt] = lds_tr16_b64<TILE_BYTES + HT + 16 * (s2_) * 128>(qtr[dt][1] + so);         \
    qtl[s2_][dt] = lds_tr16_b64<HT + 16 * (s2_) * 128>(qtr[dt][0] + so);                      \
    qth[s2_][dt] = lds_tr16_b64<HT + 16 * (s2_) * 128>(qtr[dt][1] + so);                      \
  }
#define TR_MFMA(s2_, YOUNGER, pf_, dsf_)                                                                                       \
  lds_wait<YOUNGER>(dol[s2_][0], doh[s2_][0], qtl[s2_][0], qth[s2_][0], dol[s2_][1], doh[s2_][1], qtl[s2_][1], qth[s2_][1]);   \
  _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                                           \
    dv[dt] = TAD_MFMA_32x32x16(join_tr(dol[s2_][dt], doh[s2_][dt]), pf_, dv[dt]);               \
    dk[dt] = TAD_MFMA_32x32x16(join_tr(qtl[s2_][dt], qth[s2_][dt]), dsf_, dk[dt]);              \
  }
      if constexpr (DMA_MODE != 3) {
        TR_ISSUE(0);
        lds_wait<16>(si[0], si[1], si[2], si[3], di[0], di[1], di[2], di[3]);  // (the counter saturates at 15: this also covers the row fragments)
        lds_wait<8>(qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3]);
      } else {  // ablation (timing only): no transposed reads at all -- how much of the kernel is LDS read traffic?
        lds_wait<0>(si[0], si[1], si[2], si[3], di[0], di[1], di[2], di[3]);
        lds_wait<0>(qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3]);
      }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = si[r >> 2][r & 3]; dp[r] = di[r >> 2][r & 3]; }
      if (t * 64 + 32 * qt + 32 > N) {  // ragged half tile (N % 32 != 0): rows >= N get exp2(c*(s - 3e30)) = 0 and delta = 0
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 64 + 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * h5 >= N) { s[r] = -3.0e30f; dp[r] = 0.f; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = TAD_MFMA_32x32x16(qa[ks], kfr[ks], s);
        dp = TAD_MFMA_32x32x16(da[ks], vfr[ks], dp);
      }
      if constexpr (DMA_MODE != 3) { TR_ISSUE(1); }
      f32x16 pm, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pm[r] = fast_exp2(s[r] * c);
        ds[r] = pm[r] * dp[r];
      }
      if constexpr (DMA_MODE != 3) {
        {
          const op16x8 pf = pack8(pm, 0), dsf = pack8(ds, 0);
          TR_MFMA(0, 8, pf, dsf);
        }
        {
          const op16x8 pf = pack8(pm, 1), dsf = pack8(ds, 1);
          TR_MFMA(1, 0, pf, dsf);
        }
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const op16x8 pf = pack8(pm, s2), dsf = pack8(ds, s2);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = TAD_MFMA_32x32x16(vfr[2 * s2 + dt], pf, dv[dt]);
            dk[dt] = TAD_MFMA_32x32x16(kfr[2 * s2 + dt], dsf, dk[dt]);
          }
        }
      }
#undef TR_ISSUE
#undef TR_MFMA
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

#ifdef TAD_GEMM_ABLATION
  if (stamps && tid == 0) {
    stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
#endif
  if (kvalid) {
    uint16_t* okp = dqkv + ((int64_t)b * N + krow) * tok + (int64_t)H * BHD + head * BHD;
    uint16_t* ovp = okp + (int64_t)H * BHD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = dt * 32 + 8 * r4 + 4 * h5;
        uint2 pk, pv;
        pk.x = pack_op16x2(dk[dt][4 * r4 + 0] * scale, dk[dt][4 * r4 + 1] * scale);
        pk.y = pack_op16x2(dk[dt][4 * r4 + 2] * scale, dk[dt][4 * r4 + 3] * scale);
        pv.x = pack_op16x2(dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1]);
        pv.y = pack_op16x2(dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]);
        *reinterpret_cast<uint2*>(okp + d) = pk;
        *reinterpret_cast<uint2*>(ovp + d) = pv;
      }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV, software-pipelined
// One wave per SIMD (the whole 512-register file), 64 keys per wave as two 32-key halves h = 0, 1, one workgroup = 256 keys.  The work
// on a (32-row query slice j, key half h) block is cut into three stages
//   A(j,h): S' = Q_j K_h^T - lse/scale, dP' = dO_j V_h^T - delta        8 MFMAs (row constants are the C operand of the chain heads)
//   B(j,h): P = exp2(c S'), dS = P o dP', both rounded to 16 bits          16 v_exp + 16 v_pk_mul + 16 v_cvt_pk
//   C(j,h): dV_h^T += dO_j^T P, dK_h^T += Q_j^T dS                          8 MFMAs
// and the loop runs them in SLOTS of 16 MFMAs in which the three streams are independent of each other:
//   slot (j, 0):  C(j-1, 1), A(j, 1)   ||  B(j, 0)  ||  transposed fragments of slice j requested
//   slot (j, 1):  C(j, 0),   A(j+1, 0) ||  B(j, 1)  ||  row fragments + row constants of slice j+1 requested
// so every matrix instruction has 2 - 3 vector instructions and at most 4 LDS reads of OTHER blocks behind it (the source is written
// gap by gap with a scheduling barrier after each: hipcc otherwise puts a block's VALU work in front of its MFMAs, and the two waves
// per SIMD of attn_bwd_dkv_kernel only overlap them by accident), operands arrive a slot ahead of their use, and the Q / dO row and
// transposed fragments are read from the LDS once per 64 keys instead of once per 32.
// LDS: ring of three 64-row stages (Q tile, dO tile, row constants) filled by LDS-DMA two tiles ahead; one barrier per tile.
// NW waves per workgroup, 64 keys each: 4 (one workgroup per CU) or 2 (two per CU, one wave per SIMD all the same: at N = 1568 the last
// key block of a (clip, head) pair is 1/4 full instead of 1/8, 13 blocks of 128 keys do the work of 12.25 where 7 of 256 did that of 6.125)
template <bool RAGGED, int NW>  // RAGGED: N % 32 != 0, rows of the last slice past the sequence are neutralised (kept out of the other build's loop)
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv_pipe_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                const float* __restrict__ rowc_g, uint16_t* __restrict__ dqkv, int N, int H,
                                                                int B, float scale, unsigned long long* stamps) {
  constexpr int TILE_BYTES = 64 * 128;
  constexpr int STAGE = 2 * TILE_BYTES + 512;  // Q tile, dO tile, 64 x (-lse/scale), 64 x (-delta)
  constexpr int NST = 3;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int KB = 64 * NW;        // keys per workgroup
  constexpr int NP = 8 / NW;         // 1-KiB DMA pieces of a 64-row tile per wave
  const int nblk = (N + KB - 1) / KB;  // 1-D XCD-aware grid (see attn_fwd.hip)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int head = (lin / nblk) % H, b = lin / nblk / H;
  const int key0 = (lin % nblk) * KB + wave * 64;  // the wave's keys: key0 + 32 h + (lane & 31)
  const int kl_ = lane & 31, h5 = lane >> 5;
  const int64_t tok = (int64_t)3 * H * BHD;
  const uint16_t* base = qkv + (int64_t)b * N * tok + head * BHD;
  const uint16_t* kbase = base + (int64_t)H * BHD;
  const uint16_t* vbase = base + (int64_t)2 * H * BHD;
  const float c = scale * LOG2E;
  const bool wave_live = key0 < N;  // wave-uniform; a half past the sequence computes on a clamped key and is not stored

#ifdef TAD_PIPE_STAMPS
#define PSTAMP(i_) if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 64 + (i_)] = __builtin_amdgcn_s_memtime();
#else
#define PSTAMP(i_)
#endif
  PSTAMP(0);
  // staging exactly as in attn_bwd_dkv_kernel (1-KiB pieces, swizzle on the per-lane source chunk, row constants by 4-byte DMA)
  const uint32_t qkv_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)tok * 2u;
  const uint32_t do_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)(H * BHD) * 2u;
  const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const auto rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dout), 0, (int)do_bytes, 0x00020000);
  uint32_t dma_q[4], dma_do[4];  // (NP used; a template-sized array in a __global__ template makes hipcc drop the host stub)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (wave + NW * i) * 8 + (lane >> 3);
    const uint32_t ch = (uint32_t)(((lane & 7) ^ sw_dual(row)) << 4);
    dma_q[i] = (uint32_t)(((int64_t)b * N + row) * tok * 2) + (uint32_t)(head * BHD * 2) + ch;
    dma_do[i] = (uint32_t)((((int64_t)b * N + row) * H + head) * BHD * 2) + ch;
  }
  const uint32_t q_step = (uint32_t)(tok * 2), do_step = (uint32_t)(H * BHD * 2);
  const int64_t bhn = (int64_t)B * H * N;
  const auto rs_rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowc_g), 0, (int)(2 * bhn * 4), 0x00020000);
  const uint32_t rc_off = (uint32_t)(((wave == 0 ? bhn : 0) + ((int64_t)b * H + head) * N + lane) * 4);
#define STAGE_TILE(slot, q0)                                                                                                   \
  {                                                                                                                            \
    char* ql_ = lds + (slot) * STAGE;                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(ql_ + (wave + NW * i) * 1024), 16, dma_q[i] + (uint32_t)(q0) * q_step, 0, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, LDS_PTR(ql_ + TILE_BYTES + (wave + NW * i) * 1024), 16, dma_do[i] + (uint32_t)(q0) * do_step, 0, 0, 0); \
    }                                                                                                                          \
    if (wave < 2) /* wave-uniform */                                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rc, LDS_PTR(ql_ + 2 * TILE_BYTES + wave * 256), 4, rc_off + (uint32_t)(q0) * 4u, 0, 0, 0); \
  }

  f32x16 dk[2][2], dv[2][2];  // [key half][d tile]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[h][dt][r] = 0.f; dv[h][dt][r] = 0.f; }

  const uint32_t lds0 = lds_addr(lds);
  uint32_t qtr0[2][2];  // lane constants, stage 0: transposed fragments of the Q tile, [d tile][first / second read]; dO tile: + TILE_BYTES
  tr_dual_addr(lds0, 0, lane, qtr0[0]);
  tr_dual_addr(lds0, 32, lane, qtr0[1]);
  uint32_t rfa0[4];     // row fragments (row lane&31 of a slice, chunk 2ks + h5) of the Q tile; dO tile: + TILE_BYTES
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rfa0[ks] = lds0 + (uint32_t)(kl_ * 128 + (((2 * ks + h5) ^ sw_dual(kl_)) << 4));
  const uint32_t rca0 = lds0 + 2 * TILE_BYTES + 16 * h5;  // row constants: 4 floats at [8 r4 + 4 h5]

  const int nt = (N + 63) / 64, ns = (N + 31) / 32;
  STAGE_TILE(0, 0);
  if (nt > 1) STAGE_TILE(1, 64);
  // the wave's K / V fragments (behind the first tiles' DMA: one memory latency for both instead of two in a row)
  op16x8 kfr[2][4], vfr[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int krow = key0 + 32 * h + kl_;
    if (krow > N - 1) krow = N - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kfr[h][ks] = *reinterpret_cast<const op16x8*>(kbase + (int64_t)krow * tok + 16 * ks + 8 * h5);
      vfr[h][ks] = *reinterpret_cast<const op16x8*>(vbase + (int64_t)krow * tok + 16 * ks + 8 * h5);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)  // (matrix B operands only: they live in the accumulation registers, see "pipeline registers" below)
    asm volatile("" : "+a"(kfr[h][0]), "+a"(kfr[h][1]), "+a"(kfr[h][2]), "+a"(kfr[h][3]), "+a"(vfr[h][0]), "+a"(vfr[h][1]), "+a"(vfr[h][2]), "+a"(vfr[h][3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // pipeline registers.  Register files are assigned by hand (hipcc's allocator, left alone, shuttles hundreds of values per tile between
  // the two files): arch VGPRs for what VALU touches (S' / dP', row constants, packed P / dS) and for K / V; accumulation registers for
  // dK / dV and for the Q / dO fragments, which only matrix instructions read.  That takes the matrix instructions as inline asm,
  // so the compiler no longer pads their hazards: by construction every consumer of a matrix result (and every matrix consumer of a
  // VALU result) sits at least two matrix instructions behind its producer -- a dependent accumulator chain has one foreign MFMA
  // between its links, B reads what A finished half a slot earlier, C reads what B packed in the previous slot.
  f32x4 lc[4];               // row constants of the slice in its B stages: -lse * log2(e), element r at [r >> 2][r & 3]
  f32x4 ndq[4];              // -delta of the next slice on its way in ...
  f32x16 ndt;                // ... and joined: the C operand of the dO V^T chain heads (dP' = dO V^T - delta costs no vector instruction)
  op16x8 qa[4], da[4];       // (arch) row fragments of the slice whose A stage is next
  s16x4 tql[2][2], tqh[2][2], tdl[2][2], tdh[2][2];  // (arch) halves of the transposed fragments [s2][dt] on their way in (Q / dO)
  op16x8 fq[2][2], fd[2][2];  // the joined fragments of the slice whose C stages are next: the compiler may need a register copy to put
                              // two 64-bit halves side by side, and a VALU write must sit two wait states in front of a matrix
                              // instruction that reads it -- so the join happens a whole gap ahead of the first use (JOIN_TR)
  f32x16 sc[2], dpc[2];      // Q K^T, dO V^T of key half h (written by A, read by B)
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
  u32x4_ pw[2][2], dw[2][2];  // packed P / dS of key half h, [h][s2] (written by B, read by C)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) { pw[h][s2] = u32x4_{0u, 0u, 0u, 0u}; dw[h][s2] = u32x4_{0u, 0u, 0u, 0u}; }

// padding around each matrix instruction (wait states the compiler would have inserted had it seen an MFMA): TAD_PIPE_PRE / _POST =
// argument of an s_nop in front of / behind it, negative = none
#ifndef TAD_PIPE_PRE
#define TAD_PIPE_PRE -1
#endif
#ifndef TAD_PIPE_POST
#define TAD_PIPE_POST -1
#endif
#define PIPE_STR2_(x) #x
#define PIPE_STR_(x) PIPE_STR2_(x)
#if TAD_PIPE_PRE >= 0
#define MFMA_PRE_ "s_nop " PIPE_STR_(TAD_PIPE_PRE) "\n\t"
#else
#define MFMA_PRE_ ""
#endif
#if TAD_PIPE_POST >= 0
#define MFMA_POST_ "\n\ts_nop " PIPE_STR_(TAD_PIPE_POST)
#else
#define MFMA_POST_ ""
#endif
#ifdef TAD_OPND_F16
#define MFMA_ASM_ MFMA_PRE_ "v_mfma_f32_32x32x16_f16"
#else
#define MFMA_ASM_ MFMA_PRE_ "v_mfma_f32_32x32x16_bf16"
#endif
#ifdef TAD_PIPE_PRE_A
#define MFMA_PRE_A_ "s_nop " PIPE_STR_(TAD_PIPE_PRE_A) "\n\t"
#else
#define MFMA_PRE_A_ ""
#endif
#ifdef TAD_PIPE_PRE_C
#define MFMA_PRE_C_ "s_nop " PIPE_STR_(TAD_PIPE_PRE_C) "\n\t"
#else
#define MFMA_PRE_C_ ""
#endif
#define MFMA_HEAD_V(d_, a_, b_) asm volatile(MFMA_PRE_A_ MFMA_ASM_ " %0, %1, %2, 0" MFMA_POST_ : "=&v"(d_) : "v"(a_), "a"(b_))
#define MFMA_HEADC_V(d_, a_, b_, c_) asm volatile(MFMA_PRE_A_ MFMA_ASM_ " %0, %1, %2, %3" MFMA_POST_ : "=&v"(d_) : "v"(a_), "a"(b_), "v"(c_))
#define MFMA_ACC_V(d_, a_, b_) asm volatile(MFMA_PRE_A_ MFMA_ASM_ " %0, %1, %2, %0" MFMA_POST_ : "+v"(d_) : "v"(a_), "a"(b_))
#define MFMA_ACC_A(d_, a_, b_) asm volatile(MFMA_PRE_C_ MFMA_ASM_ " %0, %1, %2, %0" MFMA_POST_ : "+a"(d_) : "v"(a_), "v"(b_))

  // requests (stage offset so_, half-tile offset HT_ of the slice)
#define REQ_CONST(so_, HT_, r4_)                                                            \
  if constexpr ((TAD_PIPE_ABL & 2) == 0) {                                                  \
    lc[r4_] = lds_read_b128<f32x4, ((HT_) / 128 + 8 * (r4_)) * 4>(rca0 + (so_));            \
  }
#define REQ_ND(so_, HT_)                                                                    \
  if constexpr ((TAD_PIPE_ABL & 2) == 0) {                                                  \
    static_for<0, 4>([&](auto r4c) {                                                        \
      constexpr int r4 = decltype(r4c)::value;                                              \
      ndq[r4] = lds_read_b128<f32x4, 256 + ((HT_) / 128 + 8 * r4) * 4>(rca0 + (so_));       \
    });                                                                                     \
  }
  // -delta of the slice with first row row0_ has arrived: rows >= N of a ragged slice get dP' = 0 (their P is 0 through lc, but
  // 0 x Inf must not appear); joined into the 16-register C operand a whole gap ahead of the matrix instruction that reads it
#define GOT_ND(row0_)                                                                                                  \
  {                                                                                                                    \
    lds_wait<0>(ndq[0], ndq[1], ndq[2], ndq[3]);                                                                       \
    if (RAGGED && (row0_) + 32 > N) {                                                                                  \
      asm volatile("; ragged slice" ::: "memory");                                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                                   \
        if ((row0_) + (r & 3) + 8 * (r >> 2) + 4 * h5 >= N) ndq[r >> 2][r & 3] = 0.f;                                  \
    }                                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) ndt[r] = ndq[r >> 2][r & 3];                                        \
    asm volatile("" : "+v"(ndt));                                                                                      \
  }
#define REQ_ROWS2(so_, HT_, ks_)                                                            \
  if constexpr ((TAD_PIPE_ABL & 2) == 0) {                                                  \
    qa[ks_] = lds_read_b128<op16x8, (HT_)>(rfa0[ks_] + (so_));                          \
    da[ks_] = lds_read_b128<op16x8, TILE_BYTES + (HT_)>(rfa0[ks_] + (so_));             \
  }
#define REQ_TR_ONE(so_, HT_, s2_, dt_)                                                      \
  if constexpr ((TAD_PIPE_ABL & 2) == 0) {                                                  \
    tdl[s2_][dt_] = lds_tr16_b64<TILE_BYTES + (HT_) + 16 * (s2_) * 128>(qtr0[dt_][0] + (so_)); \
    tdh[s2_][dt_] = lds_tr16_b64<TILE_BYTES + (HT_) + 16 * (s2_) * 128>(qtr0[dt_][1] + (so_)); \
    tql[s2_][dt_] = lds_tr16_b64<(HT_) + 16 * (s2_) * 128>(qtr0[dt_][0] + (so_));       \
    tqh[s2_][dt_] = lds_tr16_b64<(HT_) + 16 * (s2_) * 128>(qtr0[dt_][1] + (so_));       \
  }
  // row constants [2 half_, 2 half_ + 1] have arrived (lc = -lse * log2(e), published in that form by the dQ kernel for this kernel);
  // rows >= N of a ragged slice (N % 32 != 0) get P = exp2(c s - 3e30) = 0 and dP' = 0
#define GOT_CONST(half_, row0_)                                                                                        \
  {                                                                                                                    \
    lds_wait<0>(lc[2 * (half_)], lc[2 * (half_) + 1]);                                                                 \
    if (RAGGED && (row0_) + 32 > N) {                                                                                  \
      asm volatile("; ragged slice" ::: "memory"); /* (keeps this a branch) */                                         \
      _Pragma("unroll") for (int r = 8 * (half_); r < 8 * (half_) + 8; ++r)                                            \
        if ((row0_) + (r & 3) + 8 * (r >> 2) + 4 * h5 >= N) lc[r >> 2][r & 3] = -3.0e30f;                              \
    }                                                                                                                  \
  }
#define WAIT_ROWS() lds_wait<0>(qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3])
#define WAIT_TR()                                                                                                        \
  {                                                                                                                      \
    lds_wait<0>(tdl[0][0], tdh[0][0], tql[0][0], tqh[0][0], tdl[0][1], tdh[0][1], tql[0][1], tqh[0][1]);             \
    lds_wait<0>(tdl[1][0], tdh[1][0], tql[1][0], tqh[1][0], tdl[1][1], tdh[1][1], tql[1][1], tqh[1][1]);             \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                                     \
      _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                                 \
        fd[s2][dt] = join_tr(tdl[s2][dt], tdh[s2][dt]);                                                                  \
        fq[s2][dt] = join_tr(tql[s2][dt], tqh[s2][dt]);                                                                  \
      }                                                                                                                  \
    asm volatile("" : "+v"(fd[0][0]), "+v"(fd[0][1]), "+v"(fd[1][0]), "+v"(fd[1][1]), "+v"(fq[0][0]), "+v"(fq[0][1]), "+v"(fq[1][0]), "+v"(fq[1][1])); \
  }

  // one gap = one matrix instruction + the vector work and LDS requests placed behind it
  // A stage, gap g in [0, 8): ks = g >> 1, Q K^T on even gaps, dO V^T on odd ones
#define GAP_A(hA_, g_)                                                                                                     \
  {                                                                                                                        \
    constexpr int ks = (g_) >> 1;                                                                                          \
    if constexpr (((g_) & 1) == 0) {                                                                                       \
      if constexpr (ks == 0) MFMA_HEAD_V(sc[hA_], qa[0], kfr[hA_][0]); else MFMA_ACC_V(sc[hA_], qa[ks], kfr[hA_][ks]);     \
    } else {                                                                                                               \
      if constexpr (ks == 0) MFMA_HEADC_V(dpc[hA_], da[0], vfr[hA_][0], ndt); else MFMA_ACC_V(dpc[hA_], da[ks], vfr[hA_][ks]); \
    }                                                                                                                      \
  }
  // C stage, gap g in [0, 8): s2 = g >> 2, dt = (g >> 1) & 1, dV on even gaps, dK on odd ones
#define GAP_C(hC_, g_)                                                                                                     \
  {                                                                                                                        \
    constexpr int s2 = (g_) >> 2, dt = ((g_) >> 1) & 1;                                                                    \
    if constexpr (((g_) & 1) == 0) {                                                                                       \
      MFMA_ACC_A(dv[hC_][dt], fd[s2][dt], pw[hC_][s2]);                                                                    \
    } else {                                                                                                               \
      MFMA_ACC_A(dk[hC_][dt], fq[s2][dt], dw[hC_][s2]);                                                                    \
    }                                                                                                                      \
  }
  // B stage.  A wave alone on its SIMD has nobody to cover the latency of a dependent vector instruction, so the seven operations of
  // an element pair p (F: t = c S + lc;  A: dP + nd;  E0, E1: P = exp2(t);  M: dS = P (dP + nd);  C0, C1: conversions of P and dS)
  // are spread over four gaps, every consumer at least one matrix instruction behind its producer:
  //   gap 2p-1: F(p), A(p)     gap 2p: E0(p), E1(p) [+ C1(p-1)]     gap 2p+1: M(p), C0(p) [+ F(p+1), A(p+1)]     gap 2p+2: C1(p)
  // Blocks follow each other slot by slot (B(j,0), B(j,1), B(j+1,0) ...), so the pipeline runs across slots: gap 15 starts pair 0 of the
  // NEXT block (key half 1 - h), gap 0 finishes pair 7 of the previous one.  (Each gap's work ends in an empty volatile asm naming its
  // results: volatile asms keep their source order, so the work can neither be collected in front of the slot -- where the
  // instruction selector put it -- nor slide behind the next MFMA.)
  f32x2 bT[2], bP[2], bD[2];  // per pair parity: t, P, dP + nd -> dS
  bD[1] = f32x2{0.f, 0.f};    // (the very first gap "finishes" a pair 7 that never was: packs zeros into the all-zero dS of C(-1, 1))
  bT[0] = bT[1] = bP[0] = bP[1] = bD[0] = f32x2{0.f, 0.f};
  // (single-issue f32 instructions through asm: hipcc packs adjacent scalar f32 adds / multiplies into v_pk_*_f32, and a packed f32
  // instruction beside matrix instructions costs 11 - 13 cycles instead of hiding in their shadow -- MI355X_MICROARCH.md, constants table)
#define S_FMA(r_, a_, b_, c_) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r_) : "v"(a_), "v"(b_), "v"(c_))
#define S_ADD(r_, a_, b_) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r_) : "v"(a_), "v"(b_))
#define S_MUL(r_, a_, b_) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r_) : "v"(a_), "v"(b_))
#define S_EXP(r_, a_) asm volatile("v_exp_f32 %0, %1" : "=v"(r_) : "v"(a_))
#ifdef TAD_OPND_F16
#define S_CVT(r_, a_, b_) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r_) : "v"(a_), "v"(b_))
#else
#define S_CVT(r_, a_, b_) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r_) : "v"(a_), "v"(b_))
#endif
#define B_FA(h_, p_)                                                                                                        \
  {                                                                                                                         \
    constexpr int r4 = (p_) >> 1, i0 = 2 * ((p_) & 1);                                                                      \
    float t0_, t1_;                                                                                                         \
    S_FMA(t0_, sc[h_][2 * (p_)], c, lc[r4][i0]);                                                                            \
    S_FMA(t1_, sc[h_][2 * (p_) + 1], c, lc[r4][i0 + 1]);                                                                    \
    bT[(p_) & 1] = f32x2{t0_, t1_};                                                                                         \
  }
#define B_E(p_)                                                                                                             \
  {                                                                                                                         \
    float e0_, e1_;                                                                                                         \
    S_EXP(e0_, bT[(p_) & 1][0]);                                                                                            \
    S_EXP(e1_, bT[(p_) & 1][1]);                                                                                            \
    bP[(p_) & 1] = f32x2{e0_, e1_};                                                                                         \
  }
#define B_MC0(h_, p_)                                                                                                       \
  {                                                                                                                         \
    {                                                                                                                       \
      float m0_, m1_;                                                                                                       \
      S_MUL(m0_, dpc[h_][2 * (p_)], bP[(p_) & 1][0]);                                                                       \
      S_MUL(m1_, dpc[h_][2 * (p_) + 1], bP[(p_) & 1][1]);                                                                   \
      bD[(p_) & 1] = f32x2{m0_, m1_};                                                                                       \
    }                                                                                                                       \
    uint32_t wp;                                                                                                            \
    S_CVT(wp, bP[(p_) & 1][0], bP[(p_) & 1][1]);                                                                            \
    pw[h_][(p_) >> 2][(p_) & 3] = wp;                                                                                       \
  }
#define B_C1(h_, p_)                                                                                                        \
  {                                                                                                                         \
    uint32_t wd;                                                                                                            \
    S_CVT(wd, bD[(p_) & 1][0], bD[(p_) & 1][1]);                                                                            \
    dw[h_][(p_) >> 2][(p_) & 3] = wd;                                                                                       \
  }
#define B_ANCHOR() asm volatile("" : "+v"(bT[0]), "+v"(bT[1]), "+v"(bP[0]), "+v"(bP[1]), "+v"(bD[0]), "+v"(bD[1]))
  // in-slot gap g_ of the block of key half hB_
#define GAP_B(hB_, g_)                                                                                                     \
  if constexpr ((TAD_PIPE_ABL & 1) == 0) {                                                                                 \
    constexpr int p = (g_) >> 1;                                                                                           \
    /* every vector instruction of the stage is an asm statement, volatile, in the order written: the compiler neither reorders   \
       them nor pads them (it counts an asm statement as zero wait states, so a VALU -> transcendental or transcendental -> VALU    \
       dependence across the MFMA asm got one or two s_nops each: 22 - 37 per slot).  What the hardware needs is met by placement:   \
       a consumer sits at least four instructions behind its producer. */                                                 \
    if constexpr (((g_) & 1) == 0) {                                                                                       \
      B_E(p);                                                                                                              \
      if constexpr (p > 0) B_C1(hB_, p - 1) else B_C1(1 - (hB_), 7);                                                       \
    } else {                                                                                                               \
      if constexpr (p < 7) B_FA(hB_, p + 1) else B_FA(1 - (hB_), 0);                                                       \
      B_MC0(hB_, p);                                                                                                       \
    }                                                                                                                      \
  }
#define GAP_END() __builtin_amdgcn_sched_barrier(0)
#ifndef TAD_PIPE_ABL
#define TAD_PIPE_ABL 0
#endif

  // (the gaps are spelled out by macro, not by static_for: asm operands inside the discarded branch of an `if constexpr` in a generic
  // lambda do not capture)
#define SLOT0_A(g_, row0_) { if constexpr ((g_) == 6) GOT_CONST(1, row0_); GAP_A(1, g_); GAP_B(0, g_); GAP_END(); }
#define SLOT0_C(g_, son_, HTN_, rown_)                                                                                     \
  {                                                                                                                        \
    if constexpr ((g_) == 7) GOT_ND(rown_);                                                                                \
    GAP_C(1, g_);                                                                                                          \
    GAP_B(0, (g_) + 8);                                                                                                    \
    if constexpr ((g_) < 4) REQ_ROWS2(son_, HTN_, (g_) & 3);                                                               \
    if constexpr ((g_) == 4) REQ_ND(son_, HTN_);                                                                           \
    GAP_END();                                                                                                             \
  }
#define SLOT1_A(g_, so_, HT_)                                                                                              \
  {                                                                                                                        \
    if constexpr ((g_) == 7) WAIT_TR();                                                                                    \
    GAP_A(0, g_);                                                                                                          \
    GAP_B(1, g_);                                                                                                          \
    if constexpr ((g_) < 4) REQ_TR_ONE(so_, HT_, ((g_) >> 1) & 1, (g_) & 1);                                               \
    GAP_END();                                                                                                             \
  }
#define SLOT1_C(g_, son_, HTN_, rown_)                                                                                     \
  {                                                                                                                        \
    if constexpr ((g_) == 6) GOT_CONST(0, rown_);                                                                          \
    GAP_C(0, g_);                                                                                                          \
    GAP_B(1, (g_) + 8);                                                                                                    \
    if constexpr ((g_) == 0) { REQ_CONST(son_, HTN_, 0); REQ_CONST(son_, HTN_, 1); }                                       \
    GAP_END();                                                                                                             \
  }
  // slot (j, 0): A(j, 1), C(j-1, 1) || B(j, 0).  Requests: row constants [2, 3] of slice j (stage so_, half tile HT_, first row row0_)
  // at gap 0, wanted at gap 7 (pair 4 starts there); row fragments of slice j+1 (stage son_, half tile HTN_) at gaps 8 - 11, wanted by
  // the next slot
#define SLOT0(so_, HT_, row0_, son_, HTN_)                                                                                 \
  {                                                                                                                        \
    REQ_CONST(so_, HT_, 2);                                                                                                \
    REQ_CONST(so_, HT_, 3);                                                                                                \
    GAP_END();                                                                                                             \
    SLOT0_A(0, row0_) SLOT0_A(1, row0_) SLOT0_A(2, row0_) SLOT0_A(3, row0_)                                                \
    SLOT0_A(4, row0_) SLOT0_A(5, row0_) SLOT0_A(6, row0_) SLOT0_A(7, row0_)                                                \
    SLOT0_C(0, son_, HTN_, (row0_) + 32) SLOT0_C(1, son_, HTN_, (row0_) + 32) SLOT0_C(2, son_, HTN_, (row0_) + 32)         \
    SLOT0_C(3, son_, HTN_, (row0_) + 32) SLOT0_C(4, son_, HTN_, (row0_) + 32) SLOT0_C(5, son_, HTN_, (row0_) + 32)         \
    SLOT0_C(6, son_, HTN_, (row0_) + 32) SLOT0_C(7, son_, HTN_, (row0_) + 32)                                              \
  }
  // slot (j, 1): A(j+1, 0), C(j, 0) || B(j, 1).  Requests: transposed fragments of slice j (so_, HT_) at gaps 0 - 3, wanted at gap 8;
  // row constants [0, 1] of slice j+1 (son_, HTN_, first row rown_) at gap 8, wanted at gap 15 (pair 0 of the next block starts there)
#define SLOT1(so_, HT_, son_, HTN_, rown_)                                                                                 \
  {                                                                                                                        \
    WAIT_ROWS();                                                                                                           \
    GAP_END();                                                                                                             \
    SLOT1_A(0, so_, HT_) SLOT1_A(1, so_, HT_) SLOT1_A(2, so_, HT_) SLOT1_A(3, so_, HT_)                                    \
    SLOT1_A(4, so_, HT_) SLOT1_A(5, so_, HT_) SLOT1_A(6, so_, HT_) SLOT1_A(7, so_, HT_)                                    \
    SLOT1_C(0, son_, HTN_, rown_) SLOT1_C(1, son_, HTN_, rown_) SLOT1_C(2, son_, HTN_, rown_) SLOT1_C(3, son_, HTN_, rown_) \
    SLOT1_C(4, son_, HTN_, rown_) SLOT1_C(5, son_, HTN_, rown_) SLOT1_C(6, son_, HTN_, rown_) SLOT1_C(7, son_, HTN_, rown_) \
  }

  if (wave_live) {
    // pipeline prologue: rows + row constants [0, 1] of slice 0 -> A(0, 0); transposed fragments of slice 0 stand in for "slice -1"
    // (C(-1, 1) adds P = dS = 0)
    REQ_ROWS2(0u, 0, 0) REQ_ROWS2(0u, 0, 1) REQ_ROWS2(0u, 0, 2) REQ_ROWS2(0u, 0, 3)
    REQ_TR_ONE(0u, 0, 0, 0) REQ_TR_ONE(0u, 0, 0, 1) REQ_TR_ONE(0u, 0, 1, 0) REQ_TR_ONE(0u, 0, 1, 1)
    WAIT_ROWS();
    WAIT_TR();
    REQ_CONST(0u, 0, 0);
    REQ_CONST(0u, 0, 1);
    REQ_ND(0u, 0);
    GOT_ND(0);
    asm volatile("s_nop 1" ::: "memory");
    GAP_A(0, 0) GAP_A(0, 1) GAP_A(0, 2) GAP_A(0, 3) GAP_A(0, 4) GAP_A(0, 5) GAP_A(0, 6) GAP_A(0, 7)
    GOT_CONST(0, 0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (A(0, 0) has landed)
    B_FA(0, 0);
    B_ANCHOR();
  }
  int slot = 0;  // ring slot of tile t
  const int nfull = ns >> 1;  // tiles with two slices; an odd last slice is peeled below (no join of pipeline registers inside the loop)
  PSTAMP(1);
  for (int t = 0; t < nfull; ++t) {
    if (t + 2 < nt) { const int s2_ = slot == 0 ? 2 : slot - 1; STAGE_TILE(s2_, (t + 2) * 64); }
    const int nslot = slot == 2 ? 0 : slot + 1;
    const uint32_t so = (uint32_t)(slot * STAGE), son = (uint32_t)(nslot * STAGE);
    if (wave_live) {
      SLOT0(so, 0, t * 64, so, 32 * 128);              // slice 2t;   rows of slice 2t+1 (this tile, second half)
      SLOT1(so, 0, so, 32 * 128, t * 64 + 32);         //             transposed fragments of 2t, constants of 2t+1
      SLOT0(so, 32 * 128, t * 64 + 32, son, 0);        // slice 2t+1; rows of slice 2t+2 (next tile; past the last slice: unused)
      SLOT1(so, 32 * 128, son, 0, t * 64 + 64);
    }
    if (t < 10) { PSTAMP(4 + 3 * t); }
    // (register copies the compiler places on the loop's back edge / exits must not read a matrix result that has not landed)
    asm volatile("s_nop 15\n\ts_waitcnt vmcnt(0)" ::"v"(sc[0]), "v"(dpc[0]) : "memory");
    if (t < 10) { PSTAMP(5 + 3 * t); }
    __syncthreads();
    if (t < 10) { PSTAMP(6 + 3 * t); }
    slot = nslot;
  }
  PSTAMP(2);
  if (wave_live) {
    asm volatile("s_nop 7" ::: "memory");
    if (ns & 1) {  // last slice alone in its tile (what it requests for "slice ns" is never used)
      const uint32_t so = (uint32_t)(slot * STAGE);
      SLOT0(so, 0, nfull * 64, so, 32 * 128);
      SLOT1(so, 0, so, 32 * 128, nfull * 64 + 32);
      // (its A stage worked on rows past the sequence; naming the result here keeps the compiler from handing those registers to
      // temporaries while the matrix instructions that write them are still in flight)
      asm volatile("s_nop 15\n\ts_nop 7" ::"v"(sc[0]), "v"(dpc[0]) : "memory");
    }
    // pipeline epilogue: the last pair of B(ns-1, 1), then C(ns-1, 1)
    B_C1(1, 7);
    asm volatile("s_nop 1" ::: "memory");
    GAP_C(1, 0) GAP_C(1, 1) GAP_C(1, 2) GAP_C(1, 3) GAP_C(1, 4) GAP_C(1, 5) GAP_C(1, 6) GAP_C(1, 7)
    // (the compiler does not know that the accumulators were written by matrix instructions: their results must have landed before
    // the stores below read them)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  }
#undef STAGE_TILE
#undef MFMA_ASM_
#undef MFMA_PRE_
#undef MFMA_POST_
#undef MFMA_HEAD_V
#undef MFMA_ACC_V
#undef MFMA_ACC_A
#undef REQ_CONST
#undef REQ_ROWS2
#undef REQ_TR_ONE
#undef GOT_CONST
#undef REQ_ND
#undef GOT_ND
#undef MFMA_HEADC_V
#undef WAIT_ROWS
#undef WAIT_TR
#undef GAP_C
#undef GAP_A
#undef GAP_B
#undef B_FA
#undef S_FMA
#undef S_ADD
#undef S_MUL
#undef S_EXP
#undef S_CVT
#undef B_E
#undef B_MC0
#undef B_C1
#undef B_ANCHOR
#undef GAP_END
#undef SLOT0
#undef SLOT1
#undef SLOT0_A
#undef SLOT0_C
#undef SLOT1_A
#undef SLOT1_C

  // dK / dV leave through the LDS (the ring is free now): an accumulator register holds 4 consecutive d of ONE key per lane, so direct
  // stores touch 64 rows per instruction, 8 bytes each (64 such instructions per wave; with one workgroup per CU nothing hides them).
  // Per wave: 64 rows x 144 B (128 + 16: row reads stay 16-byte aligned, the 8-byte writes of 32 consecutive keys spread over the
  // banks), written as 8-byte pieces, read back as 16-byte pieces of whole rows: 8 store instructions of 64 x 16 B per tensor.
  __syncthreads();  // (every wave is past its last read of the tile ring)
  {
    constexpr int ROWB = 144;
    char* ep = lds + wave * (64 * ROWB);
    static_for<0, 2>([&](auto tc) {
      constexpr int isv = decltype(tc)::value;  // 0: dK (x scale), 1: dV
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x16& acc = isv ? dv[h][dt] : dk[h][dt];
            const float m = isv ? 1.0f : scale;
            uint2 pk;
            pk.x = pack_op16x2(acc[4 * r4 + 0] * m, acc[4 * r4 + 1] * m);
            pk.y = pack_op16x2(acc[4 * r4 + 2] * m, acc[4 * r4 + 3] * m);
            *reinterpret_cast<uint2*>(ep + (32 * h + kl_) * ROWB + (dt * 32 + 8 * r4 + 4 * h5) * 2) = pk;
          }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 8 * i + (lane >> 3), ch = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(ep + row * ROWB + ch * 16);
        const int krow = key0 + row;
        if (krow < N && wave_live)
          *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + krow) * tok + (int64_t)(1 + isv) * H * BHD + head * BHD + ch * 8) = v;
      }
    });
  }
  PSTAMP(3);
#undef PSTAMP
}

TAD_NAMESPACE_END

using namespace tad;

// Process-wide state of the attention kernels: one copy for the library (defined by the bf16 pass, shared by the half pass).
namespace tad { namespace knobs {
#ifndef TAD_OPND_F16
unsigned long long* attn_stamps = nullptr;
int attn_dma_mode = getenv("TAD_ATTN_DMA_MODE") ? atoi(getenv("TAD_ATTN_DMA_MODE")) : 0;  // 2 / 3: timing-only ablations (ablation builds); shared with attn_fwd.hip
int attn_dkv_pipe = getenv("TAD_ATTN_DKV_PIPE") ? atoi(getenv("TAD_ATTN_DKV_PIPE")) : 0;  // 1: software-pipelined dK/dV kernel (N % 32 == 0)
#else
extern int attn_dkv_pipe;
extern unsigned long long* attn_stamps;
extern int attn_dma_mode;
#endif
}}  // namespace tad::knobs
using namespace tad::knobs;

#ifndef TAD_OPND_F16
extern "C" int tad_attn_tuning(const char* key, int value) {
  TAD_REQUIRE(key, "attn_tuning: null key");
  if (!strcmp(key, "dma_mode")) {
#ifdef TAD_GEMM_ABLATION
    TAD_REQUIRE(value == 0 || value == 2 || value == 3, "attn_tuning: dma_mode=%d not in {0, 2, 3}", value);
#else
    TAD_REQUIRE(value == 0, "attn_tuning: dma_mode=%d: only 0 outside ablation builds (2 / 3 are timing-only ablations)", value);
#endif
    attn_dma_mode = value;
    return TAD_OK;
  }
  if (!strcmp(key, "dkv_pipe")) {
    TAD_REQUIRE(value >= 0 && value <= 2, "attn_tuning: dkv_pipe=%d not in {0, 1, 2}", value);
    attn_dkv_pipe = value;
    return TAD_OK;
  }
  set_error("attn_tuning: unknown key '%s'", key);
  return TAD_EINVAL;
}

// Diagnostic (ablation builds only, like tad_linear_debug_stamps): while buf (device memory, 32 bytes per workgroup of the dK/dV grid)
// is set, workgroup w records {s_memrealtime, s_memtime} at the start and at the end of its tile loop in buf[4w .. 4w+3].
extern "C" int tad_attn_debug_stamps(void* buf) {
#if !defined(TAD_GEMM_ABLATION) && !defined(TAD_PIPE_STAMPS)
  if (buf) { set_error("attn_debug_stamps: needs an ablation build (TAD_BUILD_ABLATION=1 python -m simple_tad_amd.build --force)"); return TAD_EINVAL; }
#endif
  attn_stamps = (unsigned long long*)buf;
  return TAD_OK;
}

extern "C" size_t tad_attn_bwd_scratch_bytes(int B, int N, int H) {
  if (B <= 0 || N <= 0 || H <= 0) return 0;
  return (size_t)2 * B * H * N * sizeof(float);
}
#endif

extern "C" int tad_attn_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* out_lo, const uint16_t* dout, const float* lse,
                            uint16_t* dqkv, float* delta, int B, int N, int H, int d, float scale, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attn_bwd: null pointer");
  TAD_REQUIRE(d == BHD, "attn_bwd: head_dim must be 64 (got %d)", d);
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attn_bwd: bad shape");
  TAD_REQUIRE(scale > 0.f, "attn_bwd: scale must be positive");
  TAD_REQUIRE((int64_t)B * H * N * 8 < (1ll << 31), "attn_bwd: B*H*N too large for the row-constant descriptor");
  TAD_REQUIRE((int64_t)B * N * 3 * H * BHD * 2 < (1ll << 32), "attn_bwd: qkv exceeds the 4 GiB buffer descriptor (B=%d N=%d H=%d)", B, N, H);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(((N + 127) / 128) * H * B)), block(256);
  const int mode = attn_dma_mode;
#define LAUNCH_BWD(M_)                                                                                                       \
  {                                                                                                                          \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<M_>), grid, block, 0, st, qkv, out, out_lo, dout, lse, delta, dqkv, N, H, B, scale, 0);   \
    int rc = check_launch("attn_bwd_dq");                                                                                    \
    if (rc) return rc;                                                                                                       \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<M_>), grid, block, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps);  \
    return check_launch("attn_bwd_dkv");                                                                                     \
  }
#ifdef TAD_GEMM_ABLATION
  if (mode == 2) LAUNCH_BWD(2)
  if (mode == 3) LAUNCH_BWD(3)  // (dQ kernel: as mode 0; dK/dV kernel: no transposed LDS reads)
#endif
  (void)mode;
  if (attn_dkv_pipe) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<0>), grid, block, 0, st, qkv, out, out_lo, dout, lse, delta, dqkv, N, H, B, scale, 1);
    int rc = check_launch("attn_bwd_dq");
    if (rc) return rc;
    if (attn_dkv_pipe == 2) {  // two waves (128 keys) per workgroup
      const dim3 grid_p((unsigned)(((N + 127) / 128) * H * B)), block_p(128);
      if (N % 32) hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<true, 2>), grid_p, block_p, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps);
      else hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<false, 2>), grid_p, block_p, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps);
    } else {
      const dim3 grid_p((unsigned)(((N + 255) / 256) * H * B));
      if (N % 32) hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<true, 4>), grid_p, block, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps);
      else hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<false, 4>), grid_p, block, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps);
    }
    return check_launch("attn_bwd_dkv_pipe");
  }
  LAUNCH_BWD(0)
#undef LAUNCH_BWD
}
