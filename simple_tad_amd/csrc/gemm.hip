// bf16 MFMA GEMMs for the Linear layers of the Video-ViT path (gfx950, wave64).
//
//  gemm_nt : C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue)      forward Linear and input-gradient (with W^T)
//  gemm_tn : C[N,K] = P[Mr,N]^T * Q[Mr,K]  (split over Mr)        weight gradient
//
// Both stage 64-deep K-tiles global -> LDS with 16-byte LDS-DMA loads (buffer_load ... lds; out-of-range rows
// read as zero through the buffer descriptor's bounds check), double-buffered, one barrier per K-tile, and use
// v_mfma_f32_16x16x32_bf16.  The LDS images are XOR-swizzled on the 16-byte chunk index (the swizzle is applied
// to the per-lane *source* address because the DMA destination is lane-linear) so that every ds_read_b128 /
// ds_read_b64_tr_b16 below is bank-conflict free (tools/lds_bank_sim.py).
//
// gemm_nt computes C^T tiles (W rows feed the MFMA "A" operand) with a permuted assignment of W rows to fragment
// lanes, so that each lane ends up holding 4*NREP *consecutive* output columns of one output row: the epilogue
// (bias / GELU / residual / layer-scale / drop-path scale / GELU-backward) runs on registers and stores 16-byte
// vectors straight to HBM without an LDS round trip.
#include <stdlib.h>
#include "common.h"

namespace tad {

int launch_reduce_partials(const float* partial, float* out, int splits, int64_t n, int accumulate, hipStream_t st);

enum { EPI_PLAIN = 0, EPI_GELU = 1, EPI_RESIDUAL = 2, EPI_DGELU = 3 };

struct GemmNT {
  const uint16_t* A;  // [M,K]
  const uint16_t* B;  // [N,K]
  void* C;            // [M,N] f32 or bf16
  const float* bias;  // [N] or null
  const float* residual;   // [M or res_mod, N] f32 or null
  const float* gamma;      // [N] or null
  const float* rowscale;   // [ceil(M/rows_per_scale)] or null
  uint16_t* preact;        // [M,N] bf16 or null (EPI_GELU)
  const uint16_t* dgelu_h; // [M,N] bf16 (EPI_DGELU)
  int rows_per_scale;
  int res_mod;  // >0: residual row index = m % res_mod (pos_embed broadcast over the batch)
  int c_bf16;
  int epi;
  int M, N, K;
};

constexpr int BK = 64;             // K-tile depth (bf16 elements) -> 128-byte LDS rows
constexpr int ROW_BYTES = BK * 2;  // 128

// swizzle of the 16-byte chunk index (0..7) within a 128-byte LDS row
__device__ __forceinline__ int sw_nt(int row) { return ((row >> 1) & 7) ^ (((row >> 4) & 3) << 1); }

// one wave issues PIECES 1-KiB LDS-DMA pieces: piece i of wave w lands at tile + (i*NW + w)*1024; off[i] is the lane's byte
// offset into the buffer resource (already swizzled), `add` the per-tile advance
template <int PIECES, int NW>
__device__ __forceinline__ void stage_tile(const void* gbase, int gbytes, char* tile, const uint32_t* off, uint32_t add, int wave) {
  // descriptor over the whole operand: reads past the end return 0 (rows beyond M / N)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gbase), 0, gbytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < PIECES; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * NW + wave) * 1024), 16, off[i] + add, 0, 0, 0);
}

// wait until at most `stages_in_flight` later stages (LOADS DMA instructions each, per wave) are still outstanding
template <int LOADS>
__device__ __forceinline__ void wait_stage(int stages_in_flight) {
  static_assert(2 * LOADS <= 63, "vmcnt immediate is 6 bits");
  if (stages_in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
  else if (stages_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(const GemmNT p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int LOADS = BM / (8 * NW) + BN / (8 * NW);
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MREP = WTM / 16, NREP = WTN / 16;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row DMA pieces per wave");
  __shared__ __attribute__((aligned(1024))) char lds[STAGES * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

    const int a_bytes = (int)((int64_t)p.M * p.K * 2), b_bytes = (int)((int64_t)p.N * p.K * 2);

  // ---- DMA addressing: one wave-instruction fills 8 LDS rows (1 KiB); lane -> (row lane>>3, physical chunk lane&7)
  const int drow = lane >> 3, dchunk = lane & 7;
  uint32_t a_off[BM / (8 * NW)], b_off[BN / (8 * NW)];
#pragma unroll
  for (int i = 0; i < BM / (8 * NW); ++i) {
    const int row = (i * NW + wave) * 8 + drow;
    a_off[i] = (uint32_t)(m0 + row) * (uint32_t)(p.K * 2) + (uint32_t)((dchunk ^ sw_nt(row)) * 16);
  }
#pragma unroll
  for (int i = 0; i < BN / (8 * NW); ++i) {
    const int row = (i * NW + wave) * 8 + drow;
    b_off[i] = (uint32_t)(n0 + row) * (uint32_t)(p.K * 2) + (uint32_t)((dchunk ^ sw_nt(row)) * 16);
  }
#define STAGE_NT(buf, kt) \
  stage_tile<BM / (8 * NW), NW>(p.A, a_bytes, lds + (buf) * STAGE_BYTES, a_off, (uint32_t)(kt) * ROW_BYTES, wave); \
  stage_tile<BN / (8 * NW), NW>(p.B, b_bytes, lds + (buf) * STAGE_BYTES + A_BYTES, b_off, (uint32_t)(kt) * ROW_BYTES, wave)

  // ---- fragment addressing
  const int c = lane & 15, kq = lane >> 4;
  // A-operand rows (output rows m): 16 consecutive rows per m-rep
  uint32_t a_rd[MREP];
  int a_sw[MREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int row = wm * WTM + i * 16 + c;
    a_rd[i] = row * ROW_BYTES;
    a_sw[i] = sw_nt(row);
  }
  // B-operand rows (output cols n), permuted so that the 4 lanes (kq = 0..3) that share an output row write one contiguous
  // 64-byte segment per store instruction:
  //   f32 output : fragment j, lane c -> W row 16j + 4*(c>>2) + (c&3)            (lane kq holds cols 16j + 4kq .. +3 : 16 B)
  //   bf16 output: fragment j, lane c -> W row 32*(j>>1) + 8*(c>>2) + 4*(j&1) + (c&3)  (pair (j,j+1): cols 32(j>>1) + 8kq .. +7 : 16 B)
  static_assert(NREP % 2 == 0, "bf16 epilogue pairs n-fragments");
  uint32_t b_rd[NREP];
  int b_sw[NREP];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int rl = p.c_bf16 ? (32 * (j >> 1) + 8 * (c >> 2) + 4 * (j & 1) + (c & 3)) : (16 * j + 4 * (c >> 2) + (c & 3));
    const int row = wn * WTN + rl;
    b_rd[j] = row * ROW_BYTES;
    b_sw[j] = sw_nt(row);
  }

  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nk) { STAGE_NT(st, st); }
  int rd = 0, wr = STAGES - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once all but the younger stages' DMAs of this wave are done; the barrier then (a) publishes every
    // wave's part of tile kt and (b) proves all waves finished reading tile kt-1, whose buffer the next DMA overwrites
    wait_stage<LOADS>(min(STAGES - 2, nk - 1 - kt));
    block_barrier();
    const char* sa = lds + rd * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
    const bool more = kt + STAGES - 1 < nk;
    const int wr_now = wr, kt_next = kt + STAGES - 1;
    rd = (rd + 1 == STAGES) ? 0 : rd + 1;
    wr = (wr + 1 == STAGES) ? 0 : wr + 1;
#define KSTEP_NT(ks)                                                                                   \
  {                                                                                                    \
    bf16x8 af[MREP], bfr[NREP];                                                                        \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j)                                                   \
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_rd[j] + (((4 * (ks) + kq) ^ b_sw[j]) << 4));  \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i)                                                   \
        af[i] = *reinterpret_cast<const bf16x8*>(sa + a_rd[i] + (((4 * (ks) + kq) ^ a_sw[i]) << 4));   \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < NREP; ++j)                                               \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);    \
  }
    // Issuing a tile's LDS-DMA pieces blocks the issuing wave for ~100 cycles per piece.  The two waves that share a SIMD
    // (wave w and w + NW/2) therefore issue them at different times: the older half before its first k-step, the younger
    // half between its two k-steps, so the SIMD's matrix pipe always has one wave feeding it.
    const bool late = wave >= NW / 2;  // wave-uniform (scalar branches); the MFMA code is shared by both halves
    if (more && !late) { STAGE_NT(wr_now, kt_next); }
    KSTEP_NT(0);
    if (more && late) { STAGE_NT(wr_now, kt_next); }
    KSTEP_NT(1);
  }

  // ---- epilogue on registers.  Lane (c = lane&15, kq = lane>>4) holds output row m = m0 + wm*WTM + 16i + c and, per n-fragment j,
  // 4 consecutive columns (see the permutation above).  bf16 outputs are stored 16 bytes (two fragments) at a time.
  const int nwave = n0 + wn * WTN;
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int m = m0 + wm * WTM + i * 16 + c;
    if (m >= p.M) continue;
    const float rsc = p.rowscale ? p.rowscale[m / p.rows_per_scale] : 1.f;
    const int64_t rrow = p.res_mod > 0 ? (m % p.res_mod) : m;
    if (!p.c_bf16) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        const int n = nwave + 16 * j + 4 * kq;
        if (n >= p.N) continue;
        f32x4 v = acc[i][j];
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.epi == EPI_GELU) {
          if (p.preact) {
            uint2 h;
            h.x = pack_bf16x2(v[0], v[1]);
            h.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(p.preact + (int64_t)m * p.N + n) = h;
          }
          v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        } else if (p.epi == EPI_DGELU) {
          const uint2 h = *reinterpret_cast<const uint2*>(p.dgelu_h + (int64_t)m * p.N + n);
          v[0] *= gelu_erf_grad(__uint_as_float(h.x << 16));
          v[1] *= gelu_erf_grad(__uint_as_float(h.x & 0xffff0000u));
          v[2] *= gelu_erf_grad(__uint_as_float(h.y << 16));
          v[3] *= gelu_erf_grad(__uint_as_float(h.y & 0xffff0000u));
        } else if (p.epi == EPI_RESIDUAL) {
          if (p.gamma) {
            const float4 g = *reinterpret_cast<const float4*>(p.gamma + n);
            v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w;
          }
          if (p.rowscale) { v[0] *= rsc; v[1] *= rsc; v[2] *= rsc; v[3] *= rsc; }
          if (p.residual) {
            const float4 r = *reinterpret_cast<const float4*>(p.residual + rrow * p.N + n);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
          }
        }
        *reinterpret_cast<float4*>((float*)p.C + (int64_t)m * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
#pragma unroll
      for (int jp = 0; jp < NREP / 2; ++jp) {
        const int n = nwave + 32 * jp + 8 * kq;
        if (n >= p.N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * jp][e]; v[4 + e] = acc[i][2 * jp + 1][e]; }
        const bool full = (n + 8 <= p.N);  // N % 4 == 0: either 8 or 4 valid columns
        if (p.bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          if (full) {
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
        }
        const int64_t o = (int64_t)m * p.N + n;
        if (p.epi == EPI_GELU) {
          if (p.preact) {
            uint4 h;
            h.x = pack_bf16x2(v[0], v[1]); h.y = pack_bf16x2(v[2], v[3]); h.z = pack_bf16x2(v[4], v[5]); h.w = pack_bf16x2(v[6], v[7]);
            if (full) *reinterpret_cast<uint4*>(p.preact + o) = h;
            else *reinterpret_cast<uint2*>(p.preact + o) = make_uint2(h.x, h.y);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.epi == EPI_DGELU) {
          uint4 h = make_uint4(0, 0, 0, 0);
          if (full) h = *reinterpret_cast<const uint4*>(p.dgelu_h + o);
          else { const uint2 t = *reinterpret_cast<const uint2*>(p.dgelu_h + o); h.x = t.x; h.y = t.y; }
          const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= gelu_erf_grad(__uint_as_float(hw[e] << 16));
            v[2 * e + 1] *= gelu_erf_grad(__uint_as_float(hw[e] & 0xffff0000u));
          }
        } else if (p.epi == EPI_RESIDUAL) {
          if (p.gamma) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < 4 || full) v[e] *= p.gamma[n + e];
          }
          if (p.rowscale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rsc;
          }
          if (p.residual) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < 4 || full) v[e] += p.residual[rrow * p.N + n + e];
          }
        }
        uint4 ob;
        ob.x = pack_bf16x2(v[0], v[1]); ob.y = pack_bf16x2(v[2], v[3]); ob.z = pack_bf16x2(v[4], v[5]); ob.w = pack_bf16x2(v[6], v[7]);
        if (full) *reinterpret_cast<uint4*>((uint16_t*)p.C + o) = ob;
        else *reinterpret_cast<uint2*>((uint16_t*)p.C + o) = make_uint2(ob.x, ob.y);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// gemm_tn: slab[s][n][k] = sum_{m in split s} P[m][n] * Q[m][k]
struct GemmTN {
  const uint16_t* P;  // [Mr, N]  (dy)
  const uint16_t* Q;  // [Mr, K]  (x)
  float* slab;        // [splits][N][K]
  int Mr, N, K;
  int rows_per_split;  // multiple of 64
};

// swizzle of the 16-byte chunk index within a tile row (rows are >= 256 bytes); changes bits 1..3 only
__device__ __forceinline__ int sw_tn(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

// transposed 16x16x32 fragment from a [64 red rows][rowbytes] tile: lane (g = lane>>4, li = lane&15) supplies rows
// 32ks + 8g + (li>>2) (+4), columns col0 + 4*(li&3); receives column col0 + li, reduction rows 32ks + 8g + 0..7
__device__ __forceinline__ bf16x8 tr_frag_tn(const char* base, int rowbytes, int ks, int col0, int lane) {
  const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
  const int r0 = 32 * ks + 8 * g + lq, r1 = r0 + 4;
  const int col = col0 + 4 * lp;
  const int ch = col >> 3, sub = (col & 7) * 2;
  const char* a0 = base + r0 * rowbytes + ((ch ^ sw_tn(r0)) << 4) + sub;
  const char* a1 = base + r1 * rowbytes + ((ch ^ sw_tn(r1)) << 4) + sub;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(a1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tn_kernel(const GemmTN p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MREP = WTM / 16, NREP = WTN / 16;
  constexpr int PROW = BM * 2, QROW = BN * 2;  // bytes per LDS row
  constexpr int P_BYTES = BK * PROW, Q_BYTES = BK * QROW;
  constexpr int STAGE_BYTES = P_BYTES + Q_BYTES;
  constexpr int P_PIECES = P_BYTES / 1024 / NW, Q_PIECES = Q_BYTES / 1024 / NW;
  constexpr int P_LPR = PROW / 16, Q_LPR = QROW / 16;  // lanes (16-B chunks) per row
  static_assert(P_BYTES % (1024 * NW) == 0 && Q_BYTES % (1024 * NW) == 0, "tile must split into 1-KiB DMA pieces per wave");
  static_assert(P_LPR <= 64 && Q_LPR <= 64 && PROW >= 256 && QROW >= 256, "row length");
  constexpr int LOADS = P_PIECES + Q_PIECES;
  __shared__ __attribute__((aligned(1024))) char lds[STAGES * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // linear id = split * tiles + tile, remapped so that one XCD runs (mostly) one split: the workgroups that stream the same
  // rows of dy / x then share them through that XCD's L2 instead of each fetching them from HBM
  const int tiles_k = (p.K + BN - 1) / BN;
  const int tiles = tiles_k * ((p.N + BM - 1) / BM);
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / tiles;
  const int tile = lin - split * tiles;
  const int tn_ = tile / tiles_k, tk_ = tile - tn_ * tiles_k;
  const int n0 = tn_ * BM, k0 = tk_ * BN;
  const int mr0 = split * p.rows_per_split;
  const int nt = p.rows_per_split / BK;

  const int p_bytes = (int)((int64_t)p.Mr * p.N * 2), q_bytes = (int)((int64_t)p.Mr * p.K * 2);

  // DMA: piece = 1 KiB = (1024/PROW) rows; lane -> row lane / P_LPR, physical chunk lane % P_LPR
  uint32_t p_off[P_PIECES], q_off[Q_PIECES];
#pragma unroll
  for (int i = 0; i < P_PIECES; ++i) {
    const int piece = i * NW + wave;
    const int row = piece * (64 / P_LPR) + lane / P_LPR;
    const int chunk = (lane % P_LPR) ^ sw_tn(row);
    // columns beyond N only feed outputs that are never stored; clamp keeps the address inside the row
    int col = n0 + chunk * 8;
    if (col > p.N - 8) col = p.N - 8;
    p_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(p.N * 2) + (uint32_t)(col * 2);
  }
#pragma unroll
  for (int i = 0; i < Q_PIECES; ++i) {
    const int piece = i * NW + wave;
    const int row = piece * (64 / Q_LPR) + lane / Q_LPR;
    const int chunk = (lane % Q_LPR) ^ sw_tn(row);
    int col = k0 + chunk * 8;
    if (col > p.K - 8) col = p.K - 8;
    q_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(p.K * 2) + (uint32_t)(col * 2);
  }
#define STAGE_TN(buf, t) \
  stage_tile<P_PIECES, NW>(p.P, p_bytes, lds + (buf) * STAGE_BYTES, p_off, (uint32_t)(t) * BK * (uint32_t)(p.N * 2), wave); \
  stage_tile<Q_PIECES, NW>(p.Q, q_bytes, lds + (buf) * STAGE_BYTES + P_BYTES, q_off, (uint32_t)(t) * BK * (uint32_t)(p.K * 2), wave)

  // transposed fragment reads: 16-lane group g = lane>>4 covers reduction rows 8g..8g+7 of a 32-deep k-step;
  // lane i = lane&15 of the group supplies row (i>>2) (+4 for the second read), columns c0 + 4*(i&3) .. +3
  const int g = lane >> 4, li = lane & 15;
  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};


#pragma unroll
  for (int st = 0; st < STAGES - 1; ++st)
    if (st < nt) { STAGE_TN(st, st); }
  int rd = 0, wr = STAGES - 1;
  for (int t = 0; t < nt; ++t) {
    wait_stage<LOADS>(min(STAGES - 2, nt - 1 - t));
    block_barrier();
    const char* sp = lds + rd * STAGE_BYTES;
    const char* sq = sp + P_BYTES;
    const bool more = t + STAGES - 1 < nt;
    const int wr_now = wr, t_next = t + STAGES - 1;
    rd = (rd + 1 == STAGES) ? 0 : rd + 1;
    wr = (wr + 1 == STAGES) ? 0 : wr + 1;
#define KSTEP_TN(ks)                                                                                        \
  {                                                                                                         \
    bf16x8 pf[MREP], qf[NREP];                                                                              \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j) qf[j] = tr_frag_tn(sq, QROW, (ks), wn * WTN + 16 * j, lane);  \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i) pf[i] = tr_frag_tn(sp, PROW, (ks), wm * WTM + 16 * i, lane);  \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i)                                                        \
        _Pragma("unroll") for (int j = 0; j < NREP; ++j)                                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);          \
  }
    const bool late = wave >= NW / 2;  // stagger the DMA issue of the two waves that share a SIMD (see gemm_nt_kernel)
    if (more && !late) { STAGE_TN(wr_now, t_next); }
    KSTEP_TN(0);
    if (more && late) { STAGE_TN(wr_now, t_next); }
    KSTEP_TN(1);
  }

  // D[row = n][col = k]: lane col = lane&15, rows 4*(lane>>4) + r
  float* out = p.slab + (int64_t)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int k = k0 + wn * WTN + 16 * j + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * WTM + 16 * i + 4 * g + r;
        if (n < p.N && k < p.K) out[(int64_t)n * p.K + k] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
static int env_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}

// Tile configurations.  NT: 1 = 256x256 (2x4 waves) 2 stages; 2 = 128x128 (2x2) 2 stages, 2 workgroups/CU;
// 3 = 256x128 (4x2) 3 stages; 4 = 128x256 (2x4) 3 stages; 5 = 256x128 2 stages.  0 = auto.
int launch_gemm_nt(const GemmNT& p, hipStream_t st) {
  if (!(p.M > 0 && p.N > 0 && p.K > 0)) { set_error("gemm_nt: empty problem"); return TAD_EINVAL; }
  if (p.K % BK) { set_error("gemm_nt: K=%d must be a multiple of %d", p.K, BK); return TAD_EINVAL; }
  if (p.N % 4) { set_error("gemm_nt: N=%d must be a multiple of 4", p.N); return TAD_EINVAL; }
  if ((int64_t)p.M * p.K * 2 >= (1ll << 32) || (int64_t)p.N * p.K * 2 >= (1ll << 32)) { set_error("gemm_nt: operand exceeds 4 GiB"); return TAD_EINVAL; }
  static const int forced = env_int("TAD_GEMM_NT_VARIANT");
  int v = forced;
  if (v == 0) v = (p.M >= 2048 && p.N >= 128) ? 3 : 2;
  auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  switch (v) {
    case 1: hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 4, 2>), dim3(tiles(256, 256)), dim3(512), 0, st, p); break;
    case 3: hipLaunchKernelGGL((gemm_nt_kernel<256, 128, 4, 2, 3>), dim3(tiles(256, 128)), dim3(512), 0, st, p); break;
    case 4: hipLaunchKernelGGL((gemm_nt_kernel<128, 256, 2, 4, 3>), dim3(tiles(128, 256)), dim3(512), 0, st, p); break;
    case 5: hipLaunchKernelGGL((gemm_nt_kernel<256, 128, 4, 2, 2>), dim3(tiles(256, 128)), dim3(512), 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 2, 2, 2>), dim3(tiles(128, 128)), dim3(256), 0, st, p); break;
  }
  return check_launch("gemm_nt");
}

// TN: 1 = 256x256 (2x4) 2 stages; 3 = 256x128 (4x2) 3 stages.  Splits over the reduction dim target ~1 workgroup per CU.
static int tn_variant() {
  static const int forced = env_int("TAD_GEMM_TN_VARIANT");
  return forced ? forced : 1;
}
static int tn_plan(int64_t Mr, int N, int K, int* splits, int* rows_per_split) {
  const int bn = tn_variant() == 1 ? 256 : 128;
  const int tiles = ((N + 255) / 256) * ((K + bn - 1) / bn);
  const int64_t ktiles = (Mr + BK - 1) / BK;
  int s = (256 + tiles - 1) / tiles;
  if (s > ktiles) s = (int)ktiles;
  if (s < 1) s = 1;
  const int64_t per = (ktiles + s - 1) / s;
  s = (int)((ktiles + per - 1) / per);
  *splits = s;
  *rows_per_split = (int)(per * BK);
  return tiles;
}

size_t gemm_tn_workspace_bytes(int64_t Mr, int N, int K) {
  int s, r;
  tn_plan(Mr, N, K, &s, &r);
  return (size_t)s * (size_t)N * (size_t)K * sizeof(float);
}

int launch_gemm_tn(const uint16_t* P, const uint16_t* Q, float* out, int accumulate, void* ws, size_t ws_bytes, int64_t Mr, int N,
                   int K, hipStream_t st) {
  if (!(Mr > 0 && N > 0 && K > 0)) { set_error("gemm_tn: empty problem"); return TAD_EINVAL; }
  if (N % 8 || K % 8) { set_error("gemm_tn: N=%d and K=%d must be multiples of 8", N, K); return TAD_EINVAL; }
  if (Mr * (int64_t)N * 2 >= (1ll << 32) || Mr * (int64_t)K * 2 >= (1ll << 32)) { set_error("gemm_tn: operand exceeds 4 GiB"); return TAD_EINVAL; }
  GemmTN p;
  p.P = P; p.Q = Q; p.slab = (float*)ws; p.Mr = (int)Mr; p.N = N; p.K = K;
  int splits;
  const int tiles = tn_plan(Mr, N, K, &splits, &p.rows_per_split);
  if (ws_bytes < (size_t)splits * N * K * sizeof(float)) { set_error("gemm_tn: workspace too small"); return TAD_ENOSPACE; }
  if (tn_variant() == 1)
    hipLaunchKernelGGL((gemm_tn_kernel<256, 256, 2, 4, 2>), dim3(tiles * splits), dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<256, 128, 4, 2, 3>), dim3(tiles * splits), dim3(512), 0, st, p);
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  return launch_reduce_partials(p.slab, out, splits, (int64_t)N * K, accumulate, st);
}

}  // namespace tad

using namespace tad;

extern "C" {

int tad_linear_fwd(const uint16_t* x, const uint16_t* w, const float* bias, void* y, int y_dtype, int epilogue, uint16_t* preact,
                   const float* residual, const float* gamma, const float* rowscale, int rows_per_scale, int64_t M, int N, int K,
                   tad_stream_t stream) {
  TAD_REQUIRE(x && w && y, "linear_fwd: null pointer");
  TAD_REQUIRE(y_dtype == TAD_F32 || y_dtype == TAD_BF16, "linear_fwd: bad y_dtype %d", y_dtype);
  TAD_REQUIRE(epilogue >= TAD_EPI_BIAS && epilogue <= TAD_EPI_BIAS_RESIDUAL, "linear_fwd: bad epilogue %d", epilogue);
  TAD_REQUIRE(!rowscale || rows_per_scale > 0, "linear_fwd: rows_per_scale must be positive");
  TAD_REQUIRE(M > 0 && M < (1ll << 31), "linear_fwd: bad M");
  GemmNT p{};
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.c_bf16 = (y_dtype == TAD_BF16);
  p.M = (int)M; p.N = N; p.K = K;
  p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  if (epilogue == TAD_EPI_BIAS_GELU) { p.epi = EPI_GELU; p.preact = preact; }
  else if (epilogue == TAD_EPI_BIAS_RESIDUAL) { p.epi = EPI_RESIDUAL; p.residual = residual; p.gamma = gamma; p.rowscale = rowscale; }
  else p.epi = EPI_PLAIN;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

int tad_linear_bwd_input(const uint16_t* dy, const uint16_t* wT, void* dx, int dx_dtype, const uint16_t* gelu_preact, int64_t M, int N,
                         int K, tad_stream_t stream) {
  TAD_REQUIRE(dy && wT && dx, "linear_bwd_input: null pointer");
  TAD_REQUIRE(dx_dtype == TAD_F32 || dx_dtype == TAD_BF16, "linear_bwd_input: bad dx_dtype %d", dx_dtype);
  TAD_REQUIRE(M > 0 && M < (1ll << 31), "linear_bwd_input: bad M");
  GemmNT p{};
  p.A = dy; p.B = wT; p.C = dx; p.c_bf16 = (dx_dtype == TAD_BF16);
  p.M = (int)M; p.N = K; p.K = N;  // dx[M,K] = dy[M,N] * (wT[K,N])^T
  p.rows_per_scale = 1;
  p.epi = gelu_preact ? EPI_DGELU : EPI_PLAIN;
  p.dgelu_h = gelu_preact;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

size_t tad_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
  const size_t a = gemm_tn_workspace_bytes(M, N, K);
  const size_t b = tad_colsum_workspace_bytes(M, N);
  return a > b ? a : b;
}

int tad_linear_bwd_weight(const uint16_t* dy, const uint16_t* x, float* dW, float* db, int accumulate, void* ws, size_t ws_bytes,
                          int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(dy && x && dW && ws, "linear_bwd_weight: null pointer");
  int rc = launch_gemm_tn(dy, x, dW, accumulate, ws, ws_bytes, M, N, K, (hipStream_t)stream);
  if (rc) return rc;
  if (db) rc = tad_colsum_bf16(dy, db, accumulate, ws, ws_bytes, M, N, stream);
  return rc;
}

// ---- PatchEmbed = im2col + NT GEMM with bias and broadcast pos_embed in the epilogue
int tad_patch_embed_fwd(const float* x, const uint16_t* w_bf16, const float* bias, const float* pos, float* out, uint16_t* cols, int B,
                        int C, int T, int H, int W, int tubelet, int patch, int D, tad_stream_t stream) {
  TAD_REQUIRE(x && w_bf16 && out && cols, "patch_embed_fwd: null pointer");
  int rc = tad_im2col_tubelets(x, cols, B, C, T, H, W, tubelet, patch, stream);
  if (rc) return rc;
  const int ntok = (T / tubelet) * (H / patch) * (W / patch);
  GemmNT p{};
  p.A = cols; p.B = w_bf16; p.C = out; p.bias = bias; p.c_bf16 = 0;
  p.M = B * ntok; p.N = D; p.K = C * tubelet * patch * patch;
  p.rows_per_scale = 1;
  p.epi = EPI_RESIDUAL;
  p.residual = pos;
  p.res_mod = ntok;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

size_t tad_patch_embed_bwd_workspace_bytes(int64_t M, int D, int K) { return tad_linear_bwd_weight_workspace_bytes(M, D, K); }

int tad_patch_embed_bwd(const uint16_t* dy_bf16, const uint16_t* cols, float* dW, float* db, void* ws, size_t ws_bytes, int64_t M, int D,
                        int K, tad_stream_t stream) {
  return tad_linear_bwd_weight(dy_bf16, cols, dW, db, 0, ws, ws_bytes, M, D, K, stream);
}

}  // extern "C"
