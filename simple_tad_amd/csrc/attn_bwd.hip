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
// QS: the q third of qkv carries the factor scale * log2(e) (see attn_fwd.hip); without it the factor is applied to the f32 scores.
// DROP: attention dropout (see attn_fwd.hip): dP reaches the softmax backward as keep ? dP / (1 - p) : 0, delta = rowsum(dO o O) is that
// of the dropped forward (so it still equals sum_k P_k dP_k), dV sees the dropped P.
// HD: head dim 64, or 80 as 64 + 16 (see attn_fwd.hip): dims 64..79 of the K / V tiles travel in side images of 32-byte rows, the score
// and dP products get a fifth k-step, dQ a third (half-used) d tile.
// (Round 4 measured the row fragments of a half tile requested as one batch of asm reads, as in the forward: the backward pair 846 ->
// 866 us, with delta moved to the vector pipe to free the registers; not kept -- experiments/README.md.)
template <int HD, bool QS, bool DROP, int DMA_MODE>
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                          const uint16_t* __restrict__ out_lo,
                                                          const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, uint16_t* __restrict__ dqkv, int N, int H, int B,
                                                          float scale, const Drop drop) {
  static_assert(HD == 64 || HD == 80, "head dim");
  constexpr bool X = HD == 80;
  constexpr int NKS = HD / 16, NDT = X ? 3 : 2;
  constexpr bool SUBD = DROP;  // delta subtracted by vector instructions (else it is the initial dP accumulator)
  constexpr int TILE_BYTES = 64 * 128;
  constexpr int SIDE_BYTES = X ? 64 * 32 : 0;
  constexpr int BUF_BYTES = 2 * TILE_BYTES + 2 * SIDE_BYTES;  // [K main | V main | K side | V side]
  constexpr int NST = 2;  // K/V tile ring depth
  __shared__ __attribute__((aligned(1024))) char lds[NST * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N + 127) / 128;  // 1-D XCD-aware grid: the blocks of one (batch, head) pair share an L2 (see attn_fwd.hip)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int head = (lin / nblk) % H, b = lin / nblk / H;
  const int q0 = (lin % nblk) * 128 + wave * 32;
  const int ql = lane & 31, h5 = lane >> 5;
  const int64_t tok = (int64_t)3 * H * HD;
  const uint16_t* base = qkv + (int64_t)b * N * tok + head * HD;
  const float c = scale * LOG2E;

  int qrow = q0 + ql;
  const bool qvalid = qrow < N;
  const bool wave_live = q0 < N;  // wave-uniform
  if (!qvalid) qrow = N - 1;
  op16x8 qf[NKS], dof[NKS];
  {
    const uint16_t* qp = base + (int64_t)qrow * tok + 8 * h5;
    const uint16_t* dp = dout + (((int64_t)b * N + qrow) * H + head) * HD + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = *reinterpret_cast<const op16x8*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const op16x8*>(dp + 16 * ks);
    }
  }
  const float lse2 = lse[((int64_t)b * H + head) * N + qrow] * LOG2E;
  const uint32_t drop_row = (uint32_t)((b * H + head) * N + qrow);  // (DROP) the lane's row of the keep mask
  // delta = rowsum(dO o O): this lane holds half of its query row (the 8-element groups 2ks + h5), lane ^ 32 the other half
  float dlt;
  {
    // With out_lo (what the forward's 16-bit rounding of O dropped) delta is taken of the UNROUNDED output.  delta = sum_k P_k dP_k
    // must cancel against the dP the kernels recompute exactly from V and dO; a delta computed from the rounded O is off by
    // dO . (O - round(O)), an error in dS proportional to P that does not cancel -- for rows whose dP_k are nearly equal across the
    // keys (dS small against delta: near-uniform attention, a common component in V) it dominated dQ / dK and the q / k gradients
    // behind them (measured on ViT-B at the real shape: the worst q-row slice of a qkv weight gradient off by 12 % in bf16, 2 % in f16).
    const int64_t orow = (((int64_t)b * N + qrow) * H + head) * HD + 8 * h5;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
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
      delta[(int64_t)B * H * N + idx] = QS ? -lse2 : -lse[idx] / scale;
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
    dma_k[i] = (uint32_t)(((int64_t)b * N + key) * tok * 2) + (uint32_t)((head + H) * HD * 2) + (uint32_t)(((lane & 7) ^ sw_dual(key)) << 4);
  }
  const uint32_t v_off = (uint32_t)(H * HD * 2), key_step = (uint32_t)(tok * 2);
  // side images (HD 80): one piece per wave and tile = 32 keys x 32 B; waves 0 / 1 move the two halves of K's, waves 2 / 3 of V's
  const uint32_t dma_s = (uint32_t)(((int64_t)b * N + 32 * (wave & 1) + (lane >> 1)) * tok * 2) + (uint32_t)((head + (1 + (wave >> 1)) * H) * HD * 2) +
                         128u + (uint32_t)((lane & 1) << 4);
#define DMA_K_(buf, kv0)                                                                                                        \
  {                                                                                                                             \
    char* kl_ = lds + (buf) * BUF_BYTES;                                                                                        \
    const uint32_t adv_ = (uint32_t)(kv0) * key_step;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + (wave + 4 * i) * 1024), 16, dma_k[i] + adv_, 0, 0, 0);     \
  }
#define DMA_V_(buf, kv0)                                                                                                        \
  {                                                                                                                             \
    char* kl_ = lds + (buf) * BUF_BYTES;                                                                                        \
    const uint32_t adv_ = (uint32_t)(kv0) * key_step;                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + TILE_BYTES + (wave + 4 * i) * 1024), 16, dma_k[i] + v_off + adv_, 0, 0, 0); \
    if constexpr (X)                                                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + 2 * TILE_BYTES + (wave >> 1) * SIDE_BYTES + (wave & 1) * 1024), 16, \
                                               dma_s + adv_, 0, 0, 0);                                                          \
  }
#define DMA_KV(buf, kv0) { DMA_K_(buf, kv0); DMA_V_(buf, kv0); }

  f32x16 dq[NDT];  // (HD 80: of dq[2] only rows 0..15 = dims 64..79 mean something)
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
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
  // K side image, transposed reads (dims 64..79 as rows of the third d tile): key 4 (G >> 1) + (li >> 2) (+ 8), dims 4 (li & 3) .. +3;
  // the 16-lane groups with G & 1 = 1 would address dims 80..95 and repeat the other group's address (rows 16..31 are never stored)
  const uint32_t ktr_s = lds_addr(lds) + (uint32_t)(2 * TILE_BYTES + (4 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + 8 * (lane & 3));
  auto dq_tile = [&](auto BUFC, int t) {
    constexpr int BUF = decltype(BUFC)::value;
    const int kv0 = t * 64;
    const bool more = t + 1 < nt;  // is there a tile to request during this one?
    constexpr int NBUF = BUF ^ 1;  // its buffer ...
    const int nkv0 = kv0 + 64;     // ... and first key
    if (more && DMA_MODE == 0) DMA_KV(NBUF, nkv0);
    const char* kl = lds + BUF * BUF_BYTES;
    const char* vl = kl + TILE_BYTES;
    // a wave whose 32 query rows all lie past the sequence (the last block of N = 1568 has one live wave of four) only helps
    // staging the tiles: its matrix / VALU slots go to the other waves on its SIMD
    if (wave_live)
    static_for<0, 2>([&](auto ktc) {
      constexpr int kt = decltype(ktc)::value;
      if (kv0 + 32 * kt >= N) return;  // a half tile past the sequence (N = 1568: the second half of the last tile) contributes nothing
      // K^T fragments for the dQ product, [s2][dt], issued now and waited for after the exponentials (asm reads: see common.h)
      s16x4 tl[2][NDT], th[2][NDT];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          tl[s2][dt] = s2 == 0 ? lds_tr16_b64<BUF * BUF_BYTES + (kt * 32) * 128>(ktr[dt][0]) : lds_tr16_b64<BUF * BUF_BYTES + (kt * 32 + 16) * 128>(ktr[dt][0]);
          th[s2][dt] = s2 == 0 ? lds_tr16_b64<BUF * BUF_BYTES + (kt * 32) * 128>(ktr[dt][1]) : lds_tr16_b64<BUF * BUF_BYTES + (kt * 32 + 16) * 128>(ktr[dt][1]);
        }
        if constexpr (X) {
          tl[s2][NDT - 1] = s2 == 0 ? lds_tr16_b64<BUF * BUF_BYTES + (kt * 32) * 32>(ktr_s) : lds_tr16_b64<BUF * BUF_BYTES + (kt * 32 + 16) * 32>(ktr_s);
          th[s2][NDT - 1] = s2 == 0 ? lds_tr16_b64<BUF * BUF_BYTES + (kt * 32 + 8) * 32>(ktr_s) : lds_tr16_b64<BUF * BUF_BYTES + (kt * 32 + 24) * 32>(ktr_s);
        }
      }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = SUBD ? 0.f : -dlt; }
      const int key = kt * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = TAD_MFMA_32x32x16(row_frag_dual(kl, key, ks, h5), qf[ks], s);
        dp = TAD_MFMA_32x32x16(row_frag_dual(vl, key, ks, h5), dof[ks], dp);
      }
      if constexpr (X) {  // dims 64..79: the side images' rows
        s = TAD_MFMA_32x32x16(*reinterpret_cast<const op16x8*>(kl + 2 * TILE_BYTES + key * 32 + h5 * 16), qf[NKS - 1], s);
        dp = TAD_MFMA_32x32x16(*reinterpret_cast<const op16x8*>(kl + 2 * TILE_BYTES + SIDE_BYTES + key * 32 + h5 * 16), dof[NKS - 1], dp);
      }
      // dS^T = P^T o (dP^T - delta); keys >= N contribute nothing
      if (kv0 + 64 > N) {  // ragged last tile only: keys >= N get P = 0
        // (one lane value against 16 literals: see attn_bwd_dkv_kernel)
        int lim = N - (kv0 + kt * 32) - 4 * h5;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) >= lim) s[r] = -1e30f;
      }
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float dpe = dp[r];
        if (DROP) dpe = drop_keep(drop, drop_row, (uint32_t)(kv0 + kt * 32 + acc_row(r, h5))) ? dpe * drop.inv_keep : 0.f;
        ds[r] = fast_exp2(QS ? s[r] - lse2 : fmaf(s[r], c, -lse2)) * (SUBD ? dpe - dlt : dpe);  // (QS: scores in log2 units)
      }
      static_for<0, 2>([&](auto sc) {
        constexpr int s2 = decltype(sc)::value;
        if constexpr (X) lds_wait<(s2 == 0 ? 6 : 0)>(tl[s2][0], th[s2][0], tl[s2][1], th[s2][1], tl[s2][NDT - 1], th[s2][NDT - 1]);
        else lds_wait<(s2 == 0 ? 4 : 0)>(tl[s2][0], th[s2][0], tl[s2][1], th[s2][1]);
        const op16x8 dsf = pack8(ds, s2);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = TAD_MFMA_32x32x16(join_tr(tl[s2][dt], th[s2][dt]), dsf, dq[dt]);
      });
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int t = 0; t < nt; t += 2) {
    dq_tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) dq_tile(std::integral_constant<int, 1>{}, t + 1);
  }

  if (wave_live) {  // (the tile ring is free: the loop ended on a barrier) whole rows through the LDS, see store_rows_via_lds
    uint2 pk[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        pk[dt][r4].x = pack_op16x2(dq[dt][4 * r4 + 0] * scale, dq[dt][4 * r4 + 1] * scale);
        pk[dt][r4].y = pack_op16x2(dq[dt][4 * r4 + 2] * scale, dq[dt][4 * r4 + 3] * scale);
      }
    uint16_t* const qdst = dqkv + ((int64_t)b * N + q0) * tok + head * HD;  // q slot
    store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, pk, qdst, tok, N - q0, lane);
    if constexpr (X) {
      uint2 pk2[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        pk2[g].x = pack_op16x2(dq[NDT - 1][4 * g + 0] * scale, dq[NDT - 1][4 * g + 1] * scale);
        pk2[g].y = pack_op16x2(dq[NDT - 1][4 * g + 2] * scale, dq[NDT - 1][4 * g + 3] * scale);
      }
      store_side16(pk2, qdst + 64, tok, N - q0, lane);
    }
  }
}
#undef DMA_K_
#undef DMA_V_
#undef DMA_KV

// ------------------------------------------------------------------------------------------------ dK, dV
#ifndef TAD_DKV_STREAM
#define TAD_DKV_STREAM 0  // 1: head_dim 64 streams the tile after next into the ring in two batches (see STREAM in attn_bwd_dkv_kernel).  Measured in
                          // round 5 (tools/ab_attn.py, same results): backward pair 832.6 us without, 836.1 with -- the wait in front of the tile
                          // barrier is wave skew, not DMA latency (the scheme that gave the four-wave weight-gradient GEMM 6 %); off
#endif
#ifndef TAD_DKV_ROWC_SGPR
#define TAD_DKV_ROWC_SGPR 0  // (experiment) 1: the row constants (-lse, -delta: initial S / dP accumulators) of a half tile come from
                             // SCALAR registers (s_load of the 2 x 32 values, requested one half tile ahead) + a move and a lane-half select per
                             // accumulator register, instead of 8 of the 24 ds_read_b128 of a half tile.  Without dropout / head_dim 80.
#endif
#ifndef TAD_DKV_ABL
#define TAD_DKV_ABL 0  // timing experiments (experiments/README.md, round 4): 1 = a quarter of the row-constant LDS reads
#endif
template <int HD, bool QS, bool DROP, int DMA_MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                           const float* __restrict__ rowc_g, uint16_t* __restrict__ dqkv, int N, int H, int B,
                                                           float scale, unsigned long long* stamps, const Drop drop) {
  static_assert(HD == 64 || HD == 80, "head dim");
  constexpr bool X = HD == 80;   // dims 64..79 of the Q / dO tiles in side images of 32-byte rows (see attn_fwd.hip)
  constexpr int NKS = HD / 16, NDT = X ? 3 : 2;
  constexpr int BHD = HD;
  constexpr int TILE_BYTES = 64 * 128;
  constexpr int SIDE_BYTES = X ? 64 * 32 : 0;
  constexpr int SIDE_OFF = 2 * TILE_BYTES + 512;
  constexpr int STAGE = SIDE_OFF + 2 * SIDE_BYTES;  // Q tile, dO tile, 64 x (-lse/scale), 64 x (-delta) [, Q side, dO side]
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
  op16x8 kfr[NKS], vfr[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
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
  // side images (HD 80): one piece per wave and tile = 32 rows x 32 B; waves 0 / 1 move the two halves of Q's, waves 2 / 3 of dO's
  const int srow = 32 * (wave & 1) + (lane >> 1);
  const uint32_t dma_s = (wave < 2 ? (uint32_t)(((int64_t)b * N + srow) * tok * 2) + (uint32_t)(head * BHD * 2)
                                   : (uint32_t)((((int64_t)b * N + srow) * H + head) * BHD * 2)) + 128u + (uint32_t)((lane & 1) << 4);
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
    if constexpr (X) {                                                                                     \
      if (wave < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(ql_ + SIDE_OFF + (wave & 1) * 1024), 16, dma_s + (uint32_t)(q0) * q_step, 0, 0, 0); \
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, LDS_PTR(ql_ + SIDE_OFF + SIDE_BYTES + (wave & 1) * 1024), 16, dma_s + (uint32_t)(q0) * do_step, 0, 0, 0); \
    }                                                                                                      \
  }
#define LOAD_QDO(buf, q0) { LOAD_Q_(buf, q0); LOAD_DO_RC_(buf, q0); }
  // STREAM (head_dim 64): the Q / dO rows of a tile travel as TWO batches -- rows 0..31 (piece 0 of Q and of dO per wave) and rows 32..63
  // (piece 1 of each, plus the row constants of the whole tile) -- and the tile after next streams into a ring slot as each half of it is
  // released: rows 0..31 behind a barrier between the two half tiles, rows 32..63 behind the barrier at the end of the tile.  Every piece has
  // at least one and a half tiles to land, and the one wait of a tile is counted (vmcnt(2): the batch requested half a tile ago stays in
  // flight) instead of vmcnt(0) on a tile requested one tile ago.
#define LOAD_HALF0_(buf, q0)                                                                                                   \
  {                                                                                                                            \
    char* ql_ = lds + (buf) * STAGE;                                                                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(ql_ + wave * 1024), 16, dma_q[0] + (uint32_t)(q0) * q_step, 0, 0, 0);              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, LDS_PTR(ql_ + TILE_BYTES + wave * 1024), 16, dma_do[0] + (uint32_t)(q0) * do_step, 0, 0, 0); \
  }
#define LOAD_HALF1_(buf, q0)                                                                                                   \
  {                                                                                                                            \
    char* ql_ = lds + (buf) * STAGE;                                                                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(ql_ + (wave + 4) * 1024), 16, dma_q[1] + (uint32_t)(q0) * q_step, 0, 0, 0);              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_do, LDS_PTR(ql_ + TILE_BYTES + (wave + 4) * 1024), 16, dma_do[1] + (uint32_t)(q0) * do_step, 0, 0, 0); \
    if (wave < 2) /* wave-uniform */                                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rc, LDS_PTR(ql_ + 2 * TILE_BYTES + wave * 256), 4, rc_off + (uint32_t)(q0) * 4u, 0, 0, 0); \
  }

  f32x16 dk[NDT], dv[NDT];  // (HD 80: of dk[2] / dv[2] only rows 0..15 = dims 64..79 mean something)
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
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
  // side images: row fragment (row lane&31 of a half tile, dims 64 + 8 h5 .. +7) and transposed fragment (rows 4 (G >> 1) + (li >> 2)
  // (+ 8), dims 64 + 4 (li & 3) .. +3; the lane groups with G & 1 = 1 repeat the other group's address) of the Q side image; dO: + SIDE_BYTES
  const uint32_t rfa_s = lds0 + (uint32_t)(SIDE_OFF + kl_ * 32 + h5 * 16);
  const uint32_t qtr_s = lds0 + (uint32_t)(SIDE_OFF + (4 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + 8 * (lane & 3));

  const int nt = (N + 63) / 64;
  constexpr bool STREAM = TAD_DKV_STREAM && !X && DMA_MODE == 0;
  constexpr bool RC_SGPR = TAD_DKV_ROWC_SGPR && !X && !DROP && DMA_MODE == 0;
  float rc_s[32], rc_d[32];  // (RC_SGPR) -lse / -delta of the 32 rows of the coming half tile, wave-uniform
  const float* const rc_base = rowc_g + ((int64_t)b * H + head) * N;
  auto rc_load = [&](int row0) {  // row0 .. row0 + 31 (clamped by the caller: rows >= N are neutralised in the ragged branch)
    if constexpr (RC_SGPR) {
      const int r0 = __builtin_amdgcn_readfirstlane(row0);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        rc_s[i] = __builtin_nontemporal_load(rc_base + bhn + r0 + i);
        rc_d[i] = __builtin_nontemporal_load(rc_base + r0 + i);
      }
    }
  };
  rc_load(0);
  LOAD_QDO(0, 0);
  if (STREAM && 1 < nt) { LOAD_HALF0_(1, 64); LOAD_HALF1_(1, 64); }
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
  // the tile loop runs in pairs so that the ring slot is a literal in each copy of the body (as in the dQ kernel): every LDS address is
  // then a lane constant + immediate instead of ~25 v_add per half tile
  auto dkv_tile = [&](auto BUFC, int t) {
    constexpr int BUF = decltype(BUFC)::value;
    constexpr int SO = BUF * STAGE;   // byte offset of the ring slot, folded into the instruction immediates
    const bool more = t + 1 < nt;     // is there a tile to request during this one?
    constexpr int nbuf = BUF ^ 1;     // its ring slot ...
    const int nq0 = (t + 1) * 64;     // ... and first query row
    if (more && DMA_MODE == 0 && !STREAM) LOAD_QDO(nbuf, nq0);
    const bool more2 = t + 2 < nt;  // (STREAM) is there a tile after next to request into this slot?
    static_for<0, 2>([&](auto qtc) {
      constexpr int qt = decltype(qtc)::value;
      if constexpr (STREAM && qt == 1) {
        // every wave is done with rows 0..31 of this slot (their fragments were consumed by the first half tile's MFMAs): the tile
        // after next starts to stream into them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more2) LOAD_HALF0_(BUF, (t + 2) * 64);
      }
      if (!wave_live) return;  // (see the dQ kernel: waves whose 32 keys all lie past the sequence only stage tiles)
      constexpr int HT = qt * 32 * 128;  // byte offset of the half tile inside a tile
      if (t * 64 + 32 * qt >= N) return;  // half tile of query rows past the sequence: P = dS = 0 there anyway
      // batch 1: initial accumulators (per-row constants; accumulator register r <-> row (r&3) + 8*(r>>2) + 4*h5) and row fragments
      constexpr int HTS = qt * 32 * 32;  // byte offset of the half tile inside a side image
      constexpr int NTR = X ? 12 : 8;    // transposed reads per batch
      f32x4 si[4], di[4];
      op16x8 qa[NKS], da[NKS];
      if constexpr (!RC_SGPR)
      static_for<0, 4>([&](auto r4c) {
        constexpr int r4 = decltype(r4c)::value;
        if constexpr (!(TAD_DKV_ABL & 1) || r4 == 0) {
          si[r4] = lds_read_b128<f32x4, SO + (qt * 32 + 8 * r4) * 4>(rca);
          di[r4] = lds_read_b128<f32x4, SO + 256 + (qt * 32 + 8 * r4) * 4>(rca);
        } else {  // (timing experiment: one quarter of the row-constant reads, values wrong)
          si[r4] = si[0];
          di[r4] = di[0];
        }
      });
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qa[ks] = lds_read_b128<op16x8, SO + HT>(rfa[ks]);
        da[ks] = lds_read_b128<op16x8, SO + TILE_BYTES + HT>(rfa[ks]);
      }
      if constexpr (X) {
        qa[NKS - 1] = lds_read_b128<op16x8, SO + HTS>(rfa_s);
        da[NKS - 1] = lds_read_b128<op16x8, SO + SIDE_BYTES + HTS>(rfa_s);
      }
      // batch 2 / 3: transposed fragments for the dV / dK products of rows 0..15 / 16..31 of the half tile
      s16x4 dol[2][NDT], doh[2][NDT], qtl[2][NDT], qth[2][NDT];  // [s2][dt]
#define TR_ISSUE(s2_)                                                                         \
  _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                          \
    dol[s2_][dt] = lds_tr16_b64<SO + TILE_BYTES + HT + 16 * (s2_) * 128>(qtr[dt][0]);         \
    doh[s2_][dt] = lds_tr16_b64<SO + TILE_BYTES + HT + 16 * (s2_) * 128>(qtr[dt][1]);         \
    qtl[s2_][dt] = lds_tr16_b64<SO + HT + 16 * (s2_) * 128>(qtr[dt][0]);                      \
    qth[s2_][dt] = lds_tr16_b64<SO + HT + 16 * (s2_) * 128>(qtr[dt][1]);                      \
  }                                                                                           \
  if constexpr (X) {                                                                          \
    dol[s2_][NDT - 1] = lds_tr16_b64<SO + SIDE_BYTES + HTS + 16 * (s2_) * 32>(qtr_s);         \
    doh[s2_][NDT - 1] = lds_tr16_b64<SO + SIDE_BYTES + HTS + (16 * (s2_) + 8) * 32>(qtr_s);   \
    qtl[s2_][NDT - 1] = lds_tr16_b64<SO + HTS + 16 * (s2_) * 32>(qtr_s);                      \
    qth[s2_][NDT - 1] = lds_tr16_b64<SO + HTS + (16 * (s2_) + 8) * 32>(qtr_s);                \
  }
#define TR_MFMA(s2_, YOUNGER, pf_, dsf_)                                                                                       \
  lds_wait<YOUNGER>(dol[s2_][0], doh[s2_][0], qtl[s2_][0], qth[s2_][0], dol[s2_][1], doh[s2_][1], qtl[s2_][1], qth[s2_][1]);   \
  if constexpr (X) lds_wait<YOUNGER>(dol[s2_][NDT - 1], doh[s2_][NDT - 1], qtl[s2_][NDT - 1], qth[s2_][NDT - 1]);              \
  _Pragma("unroll") for (int dt = 0; dt < NDT; ++dt) {                                                                         \
    dv[dt] = TAD_MFMA_32x32x16(join_tr(dol[s2_][dt], doh[s2_][dt]), pf_, dv[dt]);               \
    dk[dt] = TAD_MFMA_32x32x16(join_tr(qtl[s2_][dt], qth[s2_][dt]), dsf_, dk[dt]);              \
  }
      if constexpr (DMA_MODE != 3 && !X) {
        TR_ISSUE(0);
        if constexpr (!RC_SGPR) lds_wait<16>(si[0], si[1], si[2], si[3], di[0], di[1], di[2], di[3]);  // (the counter saturates at 15: this also covers the row fragments)
        lds_wait<NTR>(qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3]);
      } else {  // ablation (timing only): no transposed reads at all -- how much of the kernel is LDS read traffic?
        lds_wait<0>(si[0], si[1], si[2], si[3], di[0], di[1], di[2], di[3]);
        lds_wait<0>(qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3]);
        if constexpr (X) lds_wait<0>(qa[NKS - 1], da[NKS - 1]);
      }
      f32x16 s, dp;
      if constexpr (RC_SGPR) {
        // accumulator register r of lane half h5 <-> row (r & 3) + 8 (r >> 2) + 4 h5 of the half tile: one scalar per lane half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[r] = h5 ? rc_s[(r & 3) + 8 * (r >> 2) + 4] : rc_s[(r & 3) + 8 * (r >> 2)];
          dp[r] = h5 ? rc_d[(r & 3) + 8 * (r >> 2) + 4] : rc_d[(r & 3) + 8 * (r >> 2)];
        }
        // request the constants of the next half tile: they land behind this half tile's matrix work
        const int nrow = min(t * 64 + 32 * qt + 32, N - 32);
        rc_load(nrow);
      } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = si[r >> 2][r & 3]; dp[r] = DROP ? 0.f : di[r >> 2][r & 3]; }
      }
      if (t * 64 + 32 * qt + 32 > N) {  // ragged half tile (N % 32 != 0): rows >= N get exp2(c*(s - 3e30)) = 0 and delta = 0
        // (one lane value against 16 literals: written as `row0 + literal + 4 h5 >= N` the compiler computed the 16 row indices in
        // front of this branch, i.e. in every half tile)
        int lim = N - (t * 64 + 32 * qt) - 4 * h5;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) >= lim) { s[r] = -3.0e30f; dp[r] = 0.f; if (DROP) di[r >> 2][r & 3] = 0.f; }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        s = TAD_MFMA_32x32x16(qa[ks], kfr[ks], s);
        dp = TAD_MFMA_32x32x16(da[ks], vfr[ks], dp);
      }
      // (head_dim 80 has no registers left to hold the row fragments and both batches of transposed fragments: there the
      // first batch is issued here, behind the S / dP products, and the second one behind the first batch's products)
      if constexpr (DMA_MODE != 3) { TR_ISSUE(X ? 0 : 1); }
      f32x16 pm, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pm[r] = QS ? fast_exp2(s[r]) : fast_exp2(s[r] * c);  // QS: q . k is in log2 units already
        if (DROP) {  // (the row constant -delta enters behind the mask; rows past the sequence have P = 0)
          const uint32_t row = (uint32_t)((b * H + head) * N + min(t * 64 + 32 * qt + acc_row(r, h5), N - 1));
          const bool keep = drop_keep(drop, row, (uint32_t)krow);
          ds[r] = pm[r] * ((keep ? dp[r] * drop.inv_keep : 0.f) + di[r >> 2][r & 3]);
          pm[r] = keep ? pm[r] * drop.inv_keep : 0.f;  // what the dV product sees
        } else
        ds[r] = pm[r] * dp[r];
      }
      if constexpr (DMA_MODE != 3) {
        {
          const op16x8 pf = pack8(pm, 0), dsf = pack8(ds, 0);
          TR_MFMA(0, (X ? 0 : NTR), pf, dsf);
        }
        if constexpr (X) { TR_ISSUE(1); }
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
    if constexpr (STREAM) {
      // the next tile has landed (its second batch was requested a whole tile ago; only the two pieces requested between the half tiles
      // of this one may still be in flight); every wave is done with this slot: its rows 32..63 and row constants take the tile after next
      if (more2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (more2) LOAD_HALF1_(BUF, (t + 2) * 64);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  for (int t = 0; t < nt; t += 2) {
    dkv_tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) dkv_tile(std::integral_constant<int, 1>{}, t + 1);
  }

#ifdef TAD_GEMM_ABLATION
  if (stamps && tid == 0) {
    stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
#endif
  if (wave_live) {  // (the tile ring is free: the loop ended on a barrier) whole rows through the LDS, see store_rows_via_lds
    uint16_t* okp = dqkv + ((int64_t)b * N + key0) * tok + (int64_t)H * BHD + head * BHD;
    uint2 pk[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        // dK = scale * dS^T Q; with QS the tile held Q' = Q * scale * log2(e)
        const float ks_ = QS ? 0.69314718055994530942f : scale;
        pk[dt][r4].x = pack_op16x2(dk[dt][4 * r4 + 0] * ks_, dk[dt][4 * r4 + 1] * ks_);
        pk[dt][r4].y = pack_op16x2(dk[dt][4 * r4 + 2] * ks_, dk[dt][4 * r4 + 3] * ks_);
      }
    store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, pk, okp, tok, N - key0, lane);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        pk[dt][r4].x = pack_op16x2(dv[dt][4 * r4 + 0], dv[dt][4 * r4 + 1]);
        pk[dt][r4].y = pack_op16x2(dv[dt][4 * r4 + 2], dv[dt][4 * r4 + 3]);
      }
    store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, pk, okp + (int64_t)H * BHD, tok, N - key0, lane);
    if constexpr (X) {  // dims 64..79: registers 0..7 of the third d tiles
      const float ks_ = QS ? 0.69314718055994530942f : scale;
      uint2 pk2[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        pk2[g].x = pack_op16x2(dk[NDT - 1][4 * g + 0] * ks_, dk[NDT - 1][4 * g + 1] * ks_);
        pk2[g].y = pack_op16x2(dk[NDT - 1][4 * g + 2] * ks_, dk[NDT - 1][4 * g + 3] * ks_);
      }
      store_side16(pk2, okp + 64, tok, N - key0, lane);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        pk2[g].x = pack_op16x2(dv[NDT - 1][4 * g + 0], dv[NDT - 1][4 * g + 1]);
        pk2[g].y = pack_op16x2(dv[NDT - 1][4 * g + 2], dv[NDT - 1][4 * g + 3]);
      }
      store_side16(pk2, okp + (int64_t)H * BHD + 64, tok, N - key0, lane);
    }
  }
}

TAD_NAMESPACE_END

using namespace tad;

// Process-wide state of the attention kernels: one copy for the library (defined by the bf16 pass, shared by the half pass).
namespace tad { namespace knobs {
#ifndef TAD_OPND_F16
unsigned long long* attn_stamps = nullptr;
int attn_dma_mode = getenv("TAD_ATTN_DMA_MODE") ? atoi(getenv("TAD_ATTN_DMA_MODE")) : 0;  // 2 / 3: timing-only ablations (ablation builds); shared with attn_fwd.hip
int attn_fwd_q64 = getenv("TAD_ATTN_FWD_Q64") ? atoi(getenv("TAD_ATTN_FWD_Q64")) : 0;  // 1: the forward with 64 query rows per wave (attn_fwd_q64_kernel; experiment, round 6)
#else
extern unsigned long long* attn_stamps;
extern int attn_dma_mode, attn_fwd_q64;
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
  if (!strcmp(key, "fwd_q64")) {
    TAD_REQUIRE(value == 0 || value == 1, "attn_tuning: fwd_q64=%d not in {0, 1}", value);
    attn_fwd_q64 = value;
    return TAD_OK;
  }
  set_error("attn_tuning: unknown key '%s'", key);
  return TAD_EINVAL;
}

// Diagnostic (ablation builds only, like tad_linear_debug_stamps): while buf (device memory, 32 bytes per workgroup of the dK/dV grid)
// is set, workgroup w records {s_memrealtime, s_memtime} at the start and at the end of its tile loop in buf[4w .. 4w+3].
extern "C" int tad_attn_debug_stamps(void* buf) {
#ifndef TAD_GEMM_ABLATION
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
                            uint16_t* dqkv, float* delta, int B, int N, int H, int d, float scale, int q_prescaled, float dropout_p,
                            uint32_t seed, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attn_bwd: null pointer");
  TAD_REQUIRE(d == 64 || d == 80, "attn_bwd: head_dim must be 64 or 80 (got %d)", d);
  const int BHD = d;
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attn_bwd: bad shape");
  TAD_REQUIRE(scale > 0.f, "attn_bwd: scale must be positive");
  Drop drop;
  TAD_REQUIRE(make_drop(dropout_p, seed, &drop), "attn_bwd: dropout_p=%g outside [0, 1)", (double)dropout_p);
  TAD_REQUIRE((int64_t)B * H * N * 8 < (1ll << 31), "attn_bwd: B*H*N too large for the row-constant descriptor");
  TAD_REQUIRE((int64_t)B * N * 3 * H * BHD * 2 < (1ll << 32), "attn_bwd: qkv exceeds the 4 GiB buffer descriptor (B=%d N=%d H=%d)", B, N, H);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(((N + 127) / 128) * H * B)), block(256);
  const int mode = attn_dma_mode;
#define LAUNCH_BWD__(Q_, D_, M_)                                                                                             \
  {                                                                                                                          \
    if (d == 64) hipLaunchKernelGGL((attn_bwd_dq_kernel<64, Q_, D_, M_>), grid, block, 0, st, qkv, out, out_lo, dout, lse, delta, dqkv, N, H, B, scale, drop);  \
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<80, Q_, D_, M_>), grid, block, 0, st, qkv, out, out_lo, dout, lse, delta, dqkv, N, H, B, scale, drop);  \
    int rc = check_launch("attn_bwd_dq");                                                                                    \
    if (rc) return rc;                                                                                                       \
    if (d == 64) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, Q_, D_, M_>), grid, block, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps, drop);  \
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<80, Q_, D_, M_>), grid, block, 0, st, qkv, dout, delta, dqkv, N, H, B, scale, attn_stamps, drop);  \
    return check_launch("attn_bwd_dkv");                                                                                     \
  }
#define LAUNCH_BWD_(Q_, M_) { if (dropout_p > 0.f) LAUNCH_BWD__(Q_, true, M_) else LAUNCH_BWD__(Q_, false, M_) }
#define LAUNCH_BWD(M_) { if (q_prescaled) LAUNCH_BWD_(true, M_) else LAUNCH_BWD_(false, M_) }
#ifdef TAD_GEMM_ABLATION
  if (mode == 2) LAUNCH_BWD(2)
  if (mode == 3) LAUNCH_BWD(3)  // (dQ kernel: as mode 0; dK/dV kernel: no transposed LDS reads)
#endif
  (void)mode;
  LAUNCH_BWD(0)
#undef LAUNCH_BWD
#undef LAUNCH_BWD_
#undef LAUNCH_BWD__
}
