// Fused space-time attention forward (flash-style, non-causal, head_dim 64) for gfx950.
//
// qkv is the packed output of the qkv Linear: [B, N, 3, H, 64] bf16.  One workgroup = 4 waves = 128 query rows of one
// (batch, head); each wave owns 32 query rows.  K/V tiles of 64 keys are staged global -> LDS by LDS-DMA (double-buffered,
// the DMA of tile t+1 issued before the MFMA work on tile t), one barrier per tile.
//
// The score tile is computed *transposed*: S^T = K * Q^T with v_mfma_f32_32x32x16_bf16, so the query index sits on the
// lane (lane & 31) and the keys on the accumulator registers.  The softmax row reductions are then in-lane max/add over
// registers plus ONE exchange between the two 32-lane halves (v_permlane32_swap), and the P^T accumulator registers are,
// after a pairwise bf16 pack, directly the B operand of the O^T += V^T * P^T MFMA (no LDS round trip for P).  V^T
// fragments come from the row-major V tile with ds_read_b64_tr_b16 (inline asm with hand-counted waits: common.h).  Both LDS
// images are XOR-swizzled so all reads are bank-conflict free (tools/lds_bank_sim.py).
#include "common.h"

TAD_NAMESPACE_BEGIN

constexpr int KV_TILE = 64; // keys per LDS tile
constexpr int Q_WAVE = 32;  // query rows per wave
constexpr int Q_BLOCK = 128;
constexpr float RESCALE_THR = 8.0f;  // log2 units
#ifndef TAD_FWD_ROWSUM_VALU
#define TAD_FWD_ROWSUM_VALU 0  // 1: row sums of P as f32 adds of the unrounded exponentials + one half swap per tile, instead of 4 MFMAs
#endif
#ifndef TAD_FWD_PV_F16
#define TAD_FWD_PV_F16 0  // (bf16 build only; experiment) 1: the P V product in IEEE half -- P in [0, 2^8] rounded to f16 (11 significant bits instead
                          // of 8), the V fragments converted bf16 -> f16 in registers (exact while |v| < 65504; 3 vector instructions per pair)
#endif
#if defined(TAD_OPND_F16) && TAD_FWD_PV_F16
#undef TAD_FWD_PV_F16
#define TAD_FWD_PV_F16 0
#endif
#ifndef TAD_FWD_CNEG
#define TAD_FWD_CNEG 0  // (experiment) 1 (pre-scaled q, no dropout, head_dim 64): the running row maximum is subtracted by the matrix pipe -- sixteen registers hold
                        // -m_run and are the C operand of the first score product of every tile -- instead of one v_sub per score (32 of ~110 vector
                        // instructions per tile); costs 16 registers
#endif
#ifndef TAD_FWD_PIPE
#define TAD_FWD_PIPE 0  // (experiment) 1: the exponentials / 16-bit packing of key group g + 1 are placed between the P V matrix instructions of group g
                        // (program order pinned by scheduling barriers), instead of all exponentials in front of all P V products
#endif
#ifndef TAD_FWD_STAGGER
#define TAD_FWD_STAGGER 0  // (experiment) N: a wave sleeps (hardware wave slot & 3) x N x 256 cycles before the first tile (see the loop head)
#endif
#ifndef TAD_FWD_ABL
#define TAD_FWD_ABL 0  // timing-only ablations of the forward tile body (experiments; WRONG results): bit 0 no exponentials, 1 no row
                       // maximum, 2 no P V products / V reads / row sums, 3 no K Q^T products / K reads, 4 no DMA and no barrier in
                       // the loop, 5 no V reads (P V products fed from the K fragments' registers)
#endif


__device__ __forceinline__ int swk(int key) { return (key >> 1) & 7; }
__device__ __forceinline__ int swv(int key) { return ((key >> 1) & 1) << 2; }

__device__ __forceinline__ float half_swap_max(float x) {
  // exchange between lane l and l^32, return max of both
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef __attribute__((ext_vector_type(8))) short s16x8_t;
#if TAD_FWD_PV_F16
typedef __attribute__((ext_vector_type(8))) _Float16 pv16x8;
typedef _Float16 pv16_t;
#define PV_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define PV_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
// eight bf16 -> eight f16: each 32-bit word holds two bf16; word << 16 / word & 0xffff0000 are their f32 images
__device__ __forceinline__ pv16x8 v_to_pv(const op16x8& v) {
  typedef __attribute__((ext_vector_type(4))) uint32_t w32x4;
  const w32x4 w = __builtin_bit_cast(w32x4, v);
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  pv16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const h2 t = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)));  // (exact: 8 significant bits)
    r[2 * i] = t[0];
    r[2 * i + 1] = t[1];
  }
  return r;
}
#else
typedef op16x8 pv16x8;
typedef op16_t pv16_t;
#define PV_MFMA_32x32x16(a, b, c) TAD_MFMA_32x32x16(a, b, c)
#define PV_MFMA_16x16x32(a, b, c) TAD_MFMA_16x16x32(a, b, c)
__device__ __forceinline__ pv16x8 v_to_pv(const op16x8& v) { return v; }
#endif

// DMA_MODE: see attn_bwd.hip (0: next tile's LDS-DMA pieces at the top of the tile; 2: timing-only ablation, ablation builds)
// QS: the q third of qkv already carries the factor scale * log2(e) (tad_linear_fwd_qkv's q_prescale): the scores leave the matrix
// pipe in log2 units.  Without it (plain q, flash-attn's contract) the factor is applied to the f32 scores: one v_fma per score where the
// pre-scaled form has a v_sub, same numerics as rounds 1-3.
// DROP: attention dropout (modeling_finetune.py:99-101; flash_attention_class.py:59-61) -- the softmax is normalised by the row sum of
// ALL probabilities (taken as f32 vector adds here), the P V product sees keep ? P / (1 - p) : 0 with the counter-based mask of
// common.h (drop_keep), which the two backward kernels and the oracle regenerate.
// HD: head dim 64, or 80 (the "huge" factories, modeling_finetune.py:390-398) handled as 64 + 16: the first 64 dims of a K / V tile
// keep the 128-byte-row LDS image and everything built on it; dims 64..79 travel in a SIDE image of 32-byte rows (one more 1-KiB DMA
// piece per wave and tile), add a fifth k-step to the score products and a third (half-used) d tile to the P V product.
template <int HD, bool OUT_BF16, bool QS, bool DROP, int DMA_MODE>
__global__ __launch_bounds__(256, TAD_FWD_ROWSUM_VALU ? 4 : 1) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, void* __restrict__ out,
                                                       uint16_t* __restrict__ out_lo, float* __restrict__ lse, int N, int H, int B,
                                                       float scale, const Drop drop) {
  constexpr bool VSUM = TAD_FWD_ROWSUM_VALU || DROP;  // row sums of P by vector adds instead of MFMAs
  static_assert(HD == 64 || HD == 80, "head dim");
  constexpr bool X = HD == 80;                                                   // 16 extra dims in the side images
  constexpr int NKS = HD / 16, NDT = X ? 3 : 2;                                  // k-steps of the score product, d tiles of the output
  constexpr int TILE_BYTES = KV_TILE * 128;                                      // 8 KiB: dims 0..63
  constexpr int SIDE_BYTES = X ? KV_TILE * 32 : 0;                               // 2 KiB: dims 64..79
  constexpr int BUF_BYTES = 2 * TILE_BYTES + 2 * SIDE_BYTES;                     // [K main | V main | K side | V side]
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) pair re-read the same K/V (400 KB); dealt round-robin over the
  // 8 XCDs every L2 fetched them again (rocprofv3 FETCH_SIZE: 5.9x the algorithmic bytes), so consecutive logical ids -- the
  // blocks of one pair -- are kept on one XCD
  const int nblk = (N + Q_BLOCK - 1) / Q_BLOCK;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int qblk = lin % nblk, pair = lin / nblk;
  const int head = pair % H, b = pair / H;
  const int q0 = qblk * Q_BLOCK + wave * Q_WAVE;
  const bool wave_live = q0 < N;  // wave-uniform
  const int ql = lane & 31, h5 = lane >> 5;
  const int64_t tok_stride = (int64_t)3 * H * HD;  // elements per token
  const uint16_t* base = qkv + (int64_t)b * N * tok_stride + head * HD;
  const uint16_t* kbase = base + (int64_t)H * HD;
  const uint16_t* vbase = base + (int64_t)2 * H * HD;
  const float c = scale * 1.44269504088896340736f;  // scale * log2(e)

  // Q^T fragments (B operand): lane (q, h5) holds Q[q][16ks + 8h5 .. +7]
  op16x8 qf[NKS];
  {
    int qrow = q0 + ql;
    if (qrow > N - 1) qrow = N - 1;  // clamped rows are computed but never stored
    const uint16_t* qp = base + (int64_t)qrow * tok_stride + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const op16x8*>(qp + 16 * ks);
  }

  // staging: K/V tiles go global -> LDS by LDS-DMA (no staging registers, no ds_write): a 1-KiB piece = 8 keys x 128 B; wave w
  // moves pieces w and w+4 of K and of V.  The LDS destination is lane-linear, so the swizzle is applied to the per-lane
  // SOURCE chunk.  Keys past the end of the tensor read as zero (buffer bounds check); keys >= N of this sequence are masked.
  const uint32_t qkv_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)tok_stride * 2u;
  const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const int dkey = lane >> 3, dch = lane & 7;
  uint32_t dma_k[2], dma_v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int key = (wave + 4 * i) * 8 + dkey;
    const uint32_t rowb = (uint32_t)(((int64_t)b * N + key) * tok_stride * 2) + (uint32_t)(head * HD * 2);
    dma_k[i] = rowb + (uint32_t)(H * HD * 2) + (uint32_t)((dch ^ swk(key)) << 4);
    dma_v[i] = rowb + (uint32_t)(2 * H * HD * 2) + (uint32_t)((dch ^ swv(key)) << 4);
  }
  const uint32_t tile_step = (uint32_t)(tok_stride * 2);  // bytes per key
  // side images (HD 80): one piece per wave and tile = 32 keys x 32 B; waves 0 / 1 move the two halves of K's, waves 2 / 3 of V's
  const uint32_t dma_s = (uint32_t)(((int64_t)b * N + 32 * (wave & 1) + (lane >> 1)) * tok_stride * 2) + (uint32_t)(head * HD * 2) +
                         (uint32_t)((1 + (wave >> 1)) * H * HD * 2) + 128u + (uint32_t)((lane & 1) << 4);
#define DMA_SIDE_(buf, kv0)                                                                                                 \
  if constexpr (X) {                                                                                                        \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(lds + (buf) * BUF_BYTES + 2 * TILE_BYTES + (wave >> 1) * SIDE_BYTES + (wave & 1) * 1024), \
                                             16, dma_s + (uint32_t)(kv0) * tile_step, 0, 0, 0);                             \
  }
#define DMA_K_(buf, kv0)                                                                                                    \
  {                                                                                                                         \
    char* kl_ = lds + (buf) * BUF_BYTES;                                                                                    \
    const uint32_t adv_ = (uint32_t)(kv0) * tile_step;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + (wave + 4 * i) * 1024), 16, dma_k[i] + adv_, 0, 0, 0); \
  }
#define DMA_V_(buf, kv0)                                                                                                    \
  {                                                                                                                         \
    char* kl_ = lds + (buf) * BUF_BYTES;                                                                                    \
    const uint32_t adv_ = (uint32_t)(kv0) * tile_step;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + TILE_BYTES + (wave + 4 * i) * 1024), 16, dma_v[i] + adv_, 0, 0, 0); \
  }
#define DMA_TILE(buf, kv0) { DMA_K_(buf, kv0); DMA_V_(buf, kv0); DMA_SIDE_(buf, kv0); }

  f32x16 o[NDT];  // (HD 80: of o[2] only rows 0..15 = dims 64..79 mean something)
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

  // V^T fragment addressing (transposed reads): 16-lane group G = lane>>4, lane li in group
  const int G = lane >> 4, li = lane & 15;
  const int v_q = li >> 2, v_p = li & 3;
  const int v_h = G >> 1, v_dc = 16 * (G & 1) + 4 * v_p;  // d offset inside a 32-wide d tile

  // lane-constant LDS addresses of the transposed V reads (keys 4*v_h + v_q (+8) of a 16-key group, d tile dt) in buffer 0;
  // swv only looks at key bit 1, i.e. at v_q
  uint32_t v_rd[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int col = dt * 32 + v_dc, key = 4 * v_h + v_q;
    v_rd[dt] = lds_addr(lds) + (uint32_t)(key * 128 + (((col >> 3) ^ swv(key)) << 4) + (col & 7) * 2);
  }

  // lane-constant LDS addresses of the K row fragments (key lane&31 of a 32-key block, chunk 2ks + h5) in buffer 0, block 0: the
  // swizzle only looks at key bits 1..3, so block kt and the buffer are plain byte offsets
  uint32_t k_rd[NKS];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd[ks] = lds_addr(lds) + (uint32_t)(ql * 128 + (((2 * ks + h5) ^ swk(ql)) << 4));
  if constexpr (X) k_rd[NKS - 1] = lds_addr(lds) + (uint32_t)(2 * TILE_BYTES + ql * 32 + h5 * 16);  // side image: dims 64 + 8 h5 .. +7 of key ql
  // side image of V, transposed reads: the 16-lane groups with G & 1 = 1 would address dims 80..95: they repeat the other group's
  // address (rows 16..31 of the third d tile are never stored)
  const uint32_t vs_rd = lds_addr(lds) + (uint32_t)(2 * TILE_BYTES + SIDE_BYTES + (4 * v_h + v_q) * 32 + 8 * v_p);

  // Row sums of P^T out of the matrix pipe instead of 32 v_add per lane and tile (the kernel is VALU-issue bound, and the sum then uses
  // the same bf16-rounded P as the PV product) -- as ONE v_mfma_f32_16x16x32_bf16 per P fragment (half the matrix time and a quarter
  // of the accumulator registers of a 32x32x16 against all-ones).  Fed to the 16x16x32 instruction, a P fragment register reads as
  // B[k-group kg = lane>>4][column lane&15]: kg 0 / 2 = keys 0..7 / 8..15 of query lane&15, kg 1 / 3 = the same of query 16 + (lane&15).
  // The selector A has ones in rows 0 and 8 over kg {0, 2} and in rows 4 and 12 over kg {1, 3}; output register 0 of lane l is
  // D[4 (l>>4)][l & 15]: lanes 0-15 and 32-47 get the sum of query l & 15, lanes 16-31 and 48-63 that of query 16 + (l & 15) -- each
  // lane its own query (lane & 31), no cross-lane step.
  pv16x8 sel;
  {
    const int m = lane & 15, kg = lane >> 4;
    const bool on = ((m & 7) == 0 && (kg & 1) == 0) || ((m & 7) == 4 && (kg & 1) == 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) sel[e] = (pv16_t)(on ? 1.0f : 0.0f);
  }

  // One K/V tile of 64 keys out of LDS ring slot BUF (a literal: every LDS address below is then lane-constant + immediate).
  //
  // All eight K fragments of a tile are requested in one batch (asm reads, counted waits): compiled from plain loads the chain ran
  // read -> s_waitcnt -> MFMA eight times per tile, one exposed LDS latency each (a wave spent 31 % of its life at s_waitcnt, round-3
  // PMC).  Their 32 registers are the ones the P / V fragments use later in the tile.
  //
  // Round 4 tried to take the offset subtraction out of the vector pipe as well (accumulators started from a per-wave LDS table of
  // -m; or offset 0 with a separately compiled general body): 32 of ~110 vector instructions per tile less, and the kernel 1.4 %
  // faster -- it is not bound by vector issue (experiments/README.md r04) -- while every variant with two definitions of the score or
  // output registers cost 24-56 VGPRs (a wave per SIMD).  One body, one v_sub per score.
  const int nt = (N + KV_TILE - 1) / KV_TILE;
  const uint32_t drop_row = (uint32_t)((b * H + head) * N + min(q0 + ql, N - 1));  // (DROP) the lane's row of the keep mask
  constexpr bool CNEG = TAD_FWD_CNEG && QS && !DROP && !X && !TAD_FWD_ABL;
  float m_run = CNEG ? 0.f : -1e30f, l_run = 0.f;  // running row maximum (units of the scores as the matrix pipe delivers them), row sum of P
  f32x16 negm;                        // (CNEG) -m_run in all sixteen registers: the C operand of a tile's first score products
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  auto fwd_tile = [&](auto BUFC, const int T) {
    constexpr int BUF = decltype(BUFC)::value;
    const int kv0 = T * KV_TILE;
    if (T + 1 < nt && DMA_MODE == 0 && !(TAD_FWD_ABL & 16)) DMA_TILE(BUF ^ 1, kv0 + KV_TILE);
    if (wave_live) {  // waves whose 32 query rows all lie past the sequence only help staging the tiles
      op16x8 kf[2][NKS];
      f32x16 s[2];
      if constexpr (!(TAD_FWD_ABL & 8)) {
      static_for<0, 2 * NKS>([&](auto ic) {
        constexpr int i_ = decltype(ic)::value, kt = i_ / NKS, ks = i_ % NKS;
        kf[kt][ks] = lds_read_b128<op16x8, BUF * BUF_BYTES + kt * 32 * (ks < 4 ? 128 : 32)>(k_rd[ks]);
      });
      if constexpr (!CNEG) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      }
      static_for<0, 2 * NKS>([&](auto ic) {
        constexpr int i_ = decltype(ic)::value, kt = i_ / NKS, ks = i_ % NKS;
        lds_wait<2 * NKS - 1 - i_>(kf[kt][ks]);
        if constexpr (CNEG && ks == 0) s[kt] = TAD_MFMA_32x32x16(kf[kt][ks], qf[ks], negm);  // scores - m_run
        else s[kt] = TAD_MFMA_32x32x16(kf[kt][ks], qf[ks], s[kt]);
      });
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = o[kt][r] * 1e-3f + (float)T;  // (something live and data dependent)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) kf[kt][ks] = qf[ks];
        }
      }
      if (kv0 + KV_TILE > N) {  // ragged last tile: mask keys >= N (one lane value against 32 literals: written with the key
                                // index on the left the compiler computes all 32 indices in front of this branch)
        int lim = N - kv0 - 4 * h5;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 32 + (r & 3) + 8 * (r >> 2) >= lim) s[kt][r] = -1e30f;
      }
      float mloc = s[0][0];
      if constexpr (!(TAD_FWD_ABL & 2)) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kt][r]);
      }
      // deferred rescale: keep the old running max while the tile max exceeds it by < RESCALE_THR (log2 units): P may reach
      // 2^THR instead of 1 (harmless in f32 / 16 bits) and the O-wide multiply disappears from almost every tile.  The decision
      // precedes the exponentiation of this tile (textbook order).
      const float m_tile = half_swap_max(mloc);
      const float cq = QS ? 1.f : c;  // log2 units per score unit
      if constexpr (CNEG) {
        // m_tile is relative to m_run here.  The first tile always moves the reference to its own maximum (m_run starts at 0, not at -inf: -inf
        // as a C operand would swallow the scores); later tiles only when they exceed it by the threshold
        if (T == 0 || __any(m_tile > RESCALE_THR)) {
          const float d = T == 0 ? m_tile : fmaxf(m_tile, 0.f);  // shift of the reference
          const float alpha = fast_exp2(-d);
          m_run += d;
          l_run *= alpha;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) negm[r] = -m_run;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] -= d;
        }
      } else
      if (__any((m_tile - m_run) * cq > RESCALE_THR)) {
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = fast_exp2((m_run - m_new) * cq);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      const float mc = m_run * c;  // (plain q only)
      pv16x8 pf[2][2];
      float psum[4] = {0.f, 0.f, 0.f, 0.f};
      constexpr bool PIPE = TAD_FWD_PIPE && !X && !DROP && !TAD_FWD_ABL;
      // elements [j0, j1) of key group g_ = 2 kt + s2 (keys 16 g_ .. 16 g_ + 15): exponential, row-sum contribution, 16-bit P
      auto p_part = [&](auto gc, int j0, int j1) {
        constexpr int g_ = decltype(gc)::value, kt = g_ >> 1, s2 = g_ & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < j0 || j >= j1) continue;
          const float sh = CNEG ? s[kt][8 * s2 + j] : QS ? s[kt][8 * s2 + j] - m_run : fmaf(s[kt][8 * s2 + j], c, -mc);
          const float pe = (TAD_FWD_ABL & 1) ? sh : fast_exp2(sh);
          if (VSUM) psum[(j + 8 * s2) & 3] += pe;
          if (DROP) pf[kt][s2][j] = (pv16_t)(drop_keep(drop, drop_row, (uint32_t)(kv0 + kt * 32 + acc_row(8 * s2 + j, h5))) ? pe * drop.inv_keep : 0.f);
          else pf[kt][s2][j] = (pv16_t)pe;
        }
      };
      if constexpr (PIPE) {
        p_part(std::integral_constant<int, 0>{}, 0, 8);
      } else {
        static_for<0, 4>([&](auto gc) { p_part(gc, 0, 8); });
      }
      f32x4 rs = {0.f, 0.f, 0.f, 0.f};
      if constexpr (TAD_FWD_ABL & 4) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) asm volatile("" ::"v"(pf[kt][s2]));
      } else if constexpr (TAD_FWD_ABL & 32) {
        static_for<0, 4>([&](auto gc) {
          constexpr int g_ = decltype(gc)::value;
          rs = PV_MFMA_16x16x32(sel, pf[g_ >> 1][g_ & 1], rs);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) o[dt] = PV_MFMA_32x32x16(v_to_pv(kf[dt][g_]), pf[g_ >> 1][g_ & 1], o[dt]);
        });
      } else {
      // V^T fragments through the asm reads of common.h (the builtin made the compiler drain the DMA of the next tile here):
      // group g = 2 kt + s2 covers keys 16g .. 16g+15; the reads of group g+1 are issued before the MFMAs of group g
      s16x4 vlo[2][NDT], vhi[2][NDT];  // [group parity][dt]
      auto v_issue = [&](auto gc, auto pc) {
        constexpr int g_ = decltype(gc)::value, par = decltype(pc)::value;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          vlo[par][dt] = lds_tr16_b64<BUF * BUF_BYTES + TILE_BYTES + g_ * 16 * 128>(v_rd[dt]);
          vhi[par][dt] = lds_tr16_b64<BUF * BUF_BYTES + TILE_BYTES + g_ * 16 * 128 + 8 * 128>(v_rd[dt]);
        }
        if constexpr (X) {
          vlo[par][NDT - 1] = lds_tr16_b64<BUF * BUF_BYTES + g_ * 16 * 32>(vs_rd);
          vhi[par][NDT - 1] = lds_tr16_b64<BUF * BUF_BYTES + g_ * 16 * 32 + 8 * 32>(vs_rd);
        }
      };
      v_issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      static_for<0, 4>([&](auto gc) {
        constexpr int g_ = decltype(gc)::value, par = g_ & 1;
        if constexpr (g_ < 3) v_issue(std::integral_constant<int, g_ + 1>{}, std::integral_constant<int, par ^ 1>{});
        if constexpr (PIPE) {
          // group g's three matrix instructions with group g + 1's exponentials between them, in this order
          __builtin_amdgcn_sched_barrier(0);
          rs = PV_MFMA_16x16x32(sel, pf[g_ >> 1][g_ & 1], rs);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g_ < 3) p_part(std::integral_constant<int, g_ + 1>{}, 0, 2);
          lds_wait<(g_ < 3 ? 4 : 0)>(vlo[par][0], vhi[par][0], vlo[par][1], vhi[par][1]);
          __builtin_amdgcn_sched_barrier(0);
          o[0] = PV_MFMA_32x32x16(v_to_pv(join_tr(vlo[par][0], vhi[par][0])), pf[g_ >> 1][g_ & 1], o[0]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g_ < 3) p_part(std::integral_constant<int, g_ + 1>{}, 2, 5);
          __builtin_amdgcn_sched_barrier(0);
          o[1] = PV_MFMA_32x32x16(v_to_pv(join_tr(vlo[par][1], vhi[par][1])), pf[g_ >> 1][g_ & 1], o[1]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (g_ < 3) p_part(std::integral_constant<int, g_ + 1>{}, 5, 8);
          __builtin_amdgcn_sched_barrier(0);
        } else {
        if (!VSUM) rs = PV_MFMA_16x16x32(sel, pf[g_ >> 1][g_ & 1], rs);
        if constexpr (X) lds_wait<(g_ < 3 ? 6 : 0)>(vlo[par][0], vhi[par][0], vlo[par][1], vhi[par][1], vlo[par][NDT - 1], vhi[par][NDT - 1]);
        else lds_wait<(g_ < 3 ? 4 : 0)>(vlo[par][0], vhi[par][0], vlo[par][1], vhi[par][1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = PV_MFMA_32x32x16(v_to_pv(join_tr(vlo[par][dt], vhi[par][dt])), pf[g_ >> 1][g_ & 1], o[dt]);
        }
      });
      }
      if (VSUM) l_run += half_swap_sum((psum[0] + psum[1]) + (psum[2] + psum[3]));
      else l_run += rs[0];  // the lane's own query: see `sel`
    }
    if constexpr (!(TAD_FWD_ABL & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };

  DMA_TILE(0, 0);
#if TAD_FWD_STAGGER
  {
    // (experiment) the workgroups that share a CU start together and run identical tile bodies: are the waves of a SIMD phase-locked (all in their
    // MFMA block, then all in their softmax)?  Delay a wave by (its hardware wave slot & 3) quarters of a tile body before the first barrier.
    const uint32_t slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 3u;  // hwreg(HW_REG_HW_ID, 0, 4): wave slot in its SIMD
    for (uint32_t i = 0; i < slot * (uint32_t)TAD_FWD_STAGGER; ++i) __builtin_amdgcn_s_sleep(4);  // 4 x 64 cycles
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    fwd_tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) fwd_tile(std::integral_constant<int, 1>{}, t + 1);
  }

  // ---- epilogue
  const float l_tot = l_run;  // already the full row sum (see FWD_TILE)
  const float inv = 1.f / l_tot;
  const int qrow = q0 + ql;
  if (OUT_BF16) {
    if (wave_live) {  // (the tile ring is free: the loop ended on a barrier) whole rows through the LDS, see store_rows_via_lds
      const int64_t obase0 = (((int64_t)b * N + q0) * H + head) * HD;
      uint2 pk[2][4], lo[2][4];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float v0 = o[dt][4 * r4 + 0] * inv, v1 = o[dt][4 * r4 + 1] * inv, v2 = o[dt][4 * r4 + 2] * inv, v3 = o[dt][4 * r4 + 3] * inv;
          pk[dt][r4].x = pack_op16x2(v0, v1);
          pk[dt][r4].y = pack_op16x2(v2, v3);
          // what the 16-bit rounding dropped (see tad_attn_fwd: the backward's delta is taken of out + out_lo).  The packed words are
          // made opaque first: with f16 operands the compiler otherwise re-derives the rounded value for this subtraction with
          // v_fma_mixlo_f16 (ONE rounding of o * inv) while the stored word is v_cvt_pk_f16_f32 of the f32 product (two roundings) --
          // on a tie of the f32 product the two land on different neighbours and out + out_lo is off by a whole f16 step.
          asm volatile("" : "+v"(pk[dt][r4].x), "+v"(pk[dt][r4].y));
          lo[dt][r4].x = pack_op16x2(v0 - op16_lo_f32(pk[dt][r4].x), v1 - op16_hi_f32(pk[dt][r4].x));
          lo[dt][r4].y = pack_op16x2(v2 - op16_lo_f32(pk[dt][r4].y), v3 - op16_hi_f32(pk[dt][r4].y));
        }
      store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, pk, (uint16_t*)out + obase0, (int64_t)H * HD, N - q0, lane);
      if (out_lo) store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, lo, out_lo + obase0, (int64_t)H * HD, N - q0, lane);
      if constexpr (X) {  // dims 64..79: registers 0..7 of the third d tile
        uint2 pk2[2], lo2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const float v0 = o[NDT - 1][4 * g + 0] * inv, v1 = o[NDT - 1][4 * g + 1] * inv, v2 = o[NDT - 1][4 * g + 2] * inv, v3 = o[NDT - 1][4 * g + 3] * inv;
          pk2[g].x = pack_op16x2(v0, v1);
          pk2[g].y = pack_op16x2(v2, v3);
          asm volatile("" : "+v"(pk2[g].x), "+v"(pk2[g].y));
          lo2[g].x = pack_op16x2(v0 - op16_lo_f32(pk2[g].x), v1 - op16_hi_f32(pk2[g].x));
          lo2[g].y = pack_op16x2(v2 - op16_lo_f32(pk2[g].y), v3 - op16_hi_f32(pk2[g].y));
        }
        store_side16(pk2, (uint16_t*)out + obase0 + 64, (int64_t)H * HD, N - q0, lane);
        if (out_lo) store_side16(lo2, out_lo + obase0 + 64, (int64_t)H * HD, N - q0, lane);
      }
    }
  } else if (qrow < N) {
    const int64_t obase = (((int64_t)b * N + qrow) * H + head) * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d = dt * 32 + 8 * r4 + 4 * h5;
        *reinterpret_cast<float4*>((float*)out + obase + d) =
            make_float4(o[dt][4 * r4 + 0] * inv, o[dt][4 * r4 + 1] * inv, o[dt][4 * r4 + 2] * inv, o[dt][4 * r4 + 3] * inv);
      }
    if constexpr (X) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
        *reinterpret_cast<float4*>((float*)out + obase + 64 + 8 * g + 4 * h5) =
            make_float4(o[NDT - 1][4 * g + 0] * inv, o[NDT - 1][4 * g + 1] * inv, o[NDT - 1][4 * g + 2] * inv, o[NDT - 1][4 * g + 3] * inv);
    }
  }
  if (qrow < N && h5 == 0 && lse) lse[((int64_t)b * H + head) * N + qrow] = QS ? (m_run + __log2f(l_tot)) * 0.69314718055994530942f : m_run * scale + __logf(l_tot);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Q64 (experiment, round 6; tad_attn_tuning("fwd_q64", 1) / TAD_ATTN_FWD_Q64=1; head_dim 64, pre-scaled q, 16-bit output, no dropout):
// the same forward with SIXTY-FOUR query rows per wave -- two 32-row halves that share every K and V^T fragment the wave reads -- and two
// waves per workgroup (128 query rows per workgroup as before, so the grid and the ragged last block are unchanged).  Why: per 32 x 32 x 16
// matrix instruction the production kernel reads 1 KiB of fragments from the LDS (a wave re-reads the whole 64-key K and V tile for its 32
// rows), twice what the four-wave GEMM loop reads per matrix-pipe cycle; sharing the fragments between two row halves halves the LDS read
// bytes per score, the lever MI355X_MICROARCH.md ranks next to "fewer VALU instructions" for a kernel that runs at the chip's power limit.
// The price: ~200 registers per lane (two score / output / Q sets), i.e. TWO waves per SIMD instead of four.  Same arithmetic in the same
// order per row as attn_fwd_kernel: results are bit-identical.
template <bool HAS_LO>
__global__ __launch_bounds__(128, 2) void attn_fwd_q64_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, uint16_t* __restrict__ out_lo,
                                                              float* __restrict__ lse, int N, int H, int B) {
  constexpr int HD = 64, NKS = 4, TILE_BYTES = KV_TILE * 128, BUF_BYTES = 2 * TILE_BYTES, NWV = 2, QH = 2;
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = (N + Q_BLOCK - 1) / Q_BLOCK;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int qblk = lin % nblk, pair = lin / nblk;
  const int head = pair % H, b = pair / H;
  const int q0 = qblk * Q_BLOCK + wave * (QH * Q_WAVE);
  const bool wave_live = q0 < N;  // wave-uniform
  const int ql = lane & 31, h5 = lane >> 5;
  const int64_t tok_stride = (int64_t)3 * H * HD;
  const uint16_t* base = qkv + (int64_t)b * N * tok_stride + head * HD;

  op16x8 qf[QH][NKS];
#pragma unroll
  for (int hh = 0; hh < QH; ++hh) {
    int qrow = q0 + 32 * hh + ql;
    if (qrow > N - 1) qrow = N - 1;  // clamped rows are computed but never stored
    const uint16_t* qp = base + (int64_t)qrow * tok_stride + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[hh][ks] = *reinterpret_cast<const op16x8*>(qp + 16 * ks);
  }
  // staging as in attn_fwd_kernel; a tile's 8 + 8 one-KiB pieces are shared by TWO waves: wave w moves pieces w, w + 2, w + 4, w + 6 of K and of V
  const uint32_t qkv_bytes = (uint32_t)B * (uint32_t)N * (uint32_t)tok_stride * 2u;
  const auto rs_qkv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)qkv_bytes, 0x00020000);
  const int dkey = lane >> 3, dch = lane & 7;
  uint32_t dma_k[4], dma_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int key = (wave + NWV * i) * 8 + dkey;
    const uint32_t rowb = (uint32_t)(((int64_t)b * N + key) * tok_stride * 2) + (uint32_t)(head * HD * 2);
    dma_k[i] = rowb + (uint32_t)(H * HD * 2) + (uint32_t)((dch ^ swk(key)) << 4);
    dma_v[i] = rowb + (uint32_t)(2 * H * HD * 2) + (uint32_t)((dch ^ swv(key)) << 4);
  }
  const uint32_t tile_step = (uint32_t)(tok_stride * 2);
#define Q64_DMA_TILE(buf, kv0)                                                                                                          \
  {                                                                                                                                     \
    char* kl_ = lds + (buf) * BUF_BYTES;                                                                                                \
    const uint32_t adv_ = (uint32_t)(kv0) * tile_step;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + (wave + NWV * i) * 1024), 16, dma_k[i] + adv_, 0, 0, 0);           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_qkv, LDS_PTR(kl_ + TILE_BYTES + (wave + NWV * i) * 1024), 16, dma_v[i] + adv_, 0, 0, 0); \
  }
  f32x16 o[QH][2];
#pragma unroll
  for (int hh = 0; hh < QH; ++hh)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[hh][dt][r] = 0.f;
  const int G = lane >> 4, li = lane & 15;
  const int v_q = li >> 2, v_p = li & 3;
  const int v_h = G >> 1, v_dc = 16 * (G & 1) + 4 * v_p;
  uint32_t v_rd[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int col = dt * 32 + v_dc, key = 4 * v_h + v_q;
    v_rd[dt] = lds_addr(lds) + (uint32_t)(key * 128 + (((col >> 3) ^ swv(key)) << 4) + (col & 7) * 2);
  }
  uint32_t k_rd[NKS];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd[ks] = lds_addr(lds) + (uint32_t)(ql * 128 + (((2 * ks + h5) ^ swk(ql)) << 4));
  op16x8 sel;  // row-sum selector of attn_fwd_kernel
  {
    const int m = lane & 15, kg = lane >> 4;
    const bool on = ((m & 7) == 0 && (kg & 1) == 0) || ((m & 7) == 4 && (kg & 1) == 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) sel[e] = (op16_t)(on ? 1.0f : 0.0f);
  }
  const int nt = (N + KV_TILE - 1) / KV_TILE;
  float m_run[QH] = {-1e30f, -1e30f}, l_run[QH] = {0.f, 0.f};
  auto fwd_tile = [&](auto BUFC, const int T) {
    constexpr int BUF = decltype(BUFC)::value;
    const int kv0 = T * KV_TILE;
    if (T + 1 < nt) Q64_DMA_TILE(BUF ^ 1, kv0 + KV_TILE);
    if (wave_live) {
      op16x8 kf[2][NKS];
      f32x16 s[QH][2];
      static_for<0, 2 * NKS>([&](auto ic) {
        constexpr int i_ = decltype(ic)::value, kt = i_ / NKS, ks = i_ % NKS;
        kf[kt][ks] = lds_read_b128<op16x8, BUF * BUF_BYTES + kt * 32 * 128>(k_rd[ks]);
      });
#pragma unroll
      for (int hh = 0; hh < QH; ++hh)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[hh][kt][r] = 0.f;
      static_for<0, 2 * NKS>([&](auto ic) {
        constexpr int i_ = decltype(ic)::value, kt = i_ / NKS, ks = i_ % NKS;
        lds_wait<2 * NKS - 1 - i_>(kf[kt][ks]);
        s[0][kt] = TAD_MFMA_32x32x16(kf[kt][ks], qf[0][ks], s[0][kt]);
        s[1][kt] = TAD_MFMA_32x32x16(kf[kt][ks], qf[1][ks], s[1][kt]);
      });
      if (kv0 + KV_TILE > N) {
        int lim = N - kv0 - 4 * h5;
        asm volatile("" : "+v"(lim));
#pragma unroll
        for (int hh = 0; hh < QH; ++hh)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 32 + (r & 3) + 8 * (r >> 2) >= lim) s[hh][kt][r] = -1e30f;
      }
      op16x8 pf[QH][2][2];
#pragma unroll
      for (int hh = 0; hh < QH; ++hh) {
        float mloc = s[hh][0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[hh][kt][r]);
        const float m_tile = half_swap_max(mloc);
        if (__any((m_tile - m_run[hh]) > RESCALE_THR)) {
          const float m_new = fmaxf(m_run[hh], m_tile);
          const float alpha = fast_exp2(m_run[hh] - m_new);
          m_run[hh] = m_new;
          l_run[hh] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[hh][dt][r] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[hh][kt][s2][j] = (op16_t)fast_exp2(s[hh][kt][8 * s2 + j] - m_run[hh]);
      }
      f32x4 rs[QH] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      s16x4 vlo[2][2], vhi[2][2];  // [group parity][dt]
      auto v_issue = [&](auto gc, auto pc) {
        constexpr int g_ = decltype(gc)::value, par = decltype(pc)::value;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          vlo[par][dt] = lds_tr16_b64<BUF * BUF_BYTES + TILE_BYTES + g_ * 16 * 128>(v_rd[dt]);
          vhi[par][dt] = lds_tr16_b64<BUF * BUF_BYTES + TILE_BYTES + g_ * 16 * 128 + 8 * 128>(v_rd[dt]);
        }
      };
      v_issue(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      static_for<0, 4>([&](auto gc) {
        constexpr int g_ = decltype(gc)::value, par = g_ & 1;
        if constexpr (g_ < 3) v_issue(std::integral_constant<int, g_ + 1>{}, std::integral_constant<int, par ^ 1>{});
#pragma unroll
        for (int hh = 0; hh < QH; ++hh) rs[hh] = TAD_MFMA_16x16x32(sel, pf[hh][g_ >> 1][g_ & 1], rs[hh]);
        lds_wait<(g_ < 3 ? 4 : 0)>(vlo[par][0], vhi[par][0], vlo[par][1], vhi[par][1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const op16x8 vf = join_tr(vlo[par][dt], vhi[par][dt]);
#pragma unroll
          for (int hh = 0; hh < QH; ++hh) o[hh][dt] = TAD_MFMA_32x32x16(vf, pf[hh][g_ >> 1][g_ & 1], o[hh][dt]);
        }
      });
#pragma unroll
      for (int hh = 0; hh < QH; ++hh) l_run[hh] += rs[hh][0];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  Q64_DMA_TILE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    fwd_tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) fwd_tile(std::integral_constant<int, 1>{}, t + 1);
  }
#undef Q64_DMA_TILE
  // ---- epilogue (the tile ring is free: the loop ended on a barrier): per 32-row half exactly attn_fwd_kernel's
#pragma unroll
  for (int hh = 0; hh < QH; ++hh) {
    const int qh0 = q0 + 32 * hh;
    if (qh0 >= N) continue;  // (wave-uniform)
    const float inv = 1.f / l_run[hh];
    const int64_t obase0 = (((int64_t)b * N + qh0) * H + head) * HD;
    uint2 pk[2][4], lo[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float v0 = o[hh][dt][4 * r4 + 0] * inv, v1 = o[hh][dt][4 * r4 + 1] * inv, v2 = o[hh][dt][4 * r4 + 2] * inv, v3 = o[hh][dt][4 * r4 + 3] * inv;
        pk[dt][r4].x = pack_op16x2(v0, v1);
        pk[dt][r4].y = pack_op16x2(v2, v3);
        asm volatile("" : "+v"(pk[dt][r4].x), "+v"(pk[dt][r4].y));  // (see attn_fwd_kernel: one rounding of the stored word, then its residual)
        lo[dt][r4].x = pack_op16x2(v0 - op16_lo_f32(pk[dt][r4].x), v1 - op16_hi_f32(pk[dt][r4].x));
        lo[dt][r4].y = pack_op16x2(v2 - op16_lo_f32(pk[dt][r4].y), v3 - op16_hi_f32(pk[dt][r4].y));
      }
    store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, pk, out + obase0, (int64_t)H * HD, N - qh0, lane);
    if (HAS_LO) store_rows_via_lds(lds + wave * ROW_PATCH_BYTES, lo, out_lo + obase0, (int64_t)H * HD, N - qh0, lane);
    const int qrow = qh0 + ql;
    if (qrow < N && h5 == 0 && lse) lse[((int64_t)b * H + head) * N + qrow] = (m_run[hh] + __log2f(l_run[hh])) * 0.69314718055994530942f;
  }
}

TAD_NAMESPACE_END

using namespace tad;

namespace tad { namespace knobs { extern int attn_dma_mode, attn_fwd_q64; } }  // attn_bwd.hip (tad_attn_tuning)

extern "C" int tad_attn_fwd(const uint16_t* qkv, void* out, int out_dtype, uint16_t* out_lo, float* lse, int B, int N, int H, int d,
                            float scale, int q_prescaled, float dropout_p, uint32_t seed, tad_stream_t stream) {
  TAD_REQUIRE(qkv && out, "attn_fwd: null pointer");
  TAD_REQUIRE(!out_lo || out_dtype == TAD_OP16, "attn_fwd: out_lo (the rounding residual) goes with a 16-bit output");
  TAD_REQUIRE(d == 64 || d == 80, "attn_fwd: head_dim must be 64 or 80 (got %d)", d);
  const int HD = d;
  TAD_REQUIRE(B > 0 && N > 0 && H > 0 && H <= 65535 && B <= 65535, "attn_fwd: bad shape B=%d N=%d H=%d", B, N, H);
  TAD_REQUIRE(out_dtype == TAD_F32 || out_dtype == TAD_OP16, "attn_fwd: bad out_dtype %d", out_dtype);
  TAD_REQUIRE(scale > 0.f, "attn_fwd: scale must be positive");
  Drop drop;
  TAD_REQUIRE(make_drop(dropout_p, seed, &drop), "attn_fwd: dropout_p=%g outside [0, 1)", (double)dropout_p);
  TAD_REQUIRE(dropout_p == 0.f || (int64_t)B * H * N < (1ll << 32), "attn_fwd: B*H*N too large for the dropout mask's row index");
  // the kernel addresses qkv through ONE buffer descriptor with 32-bit byte offsets (K/V staging by LDS-DMA)
  TAD_REQUIRE((int64_t)B * N * 3 * H * HD * 2 < (1ll << 32), "attn_fwd: qkv of %lld bytes exceeds the 4 GiB buffer descriptor (B=%d N=%d H=%d)",
              (long long)B * N * 3 * H * HD * 2, B, N, H);
  TAD_REQUIRE((int64_t)((N + Q_BLOCK - 1) / Q_BLOCK) * H * B < (1ll << 31), "attn_fwd: grid too large");
  const dim3 grid((unsigned)(((N + Q_BLOCK - 1) / Q_BLOCK) * H * B)), block(256);
  // (experiment) sixty-four query rows per wave: attn_fwd_q64_kernel, same grid, 128 threads -- only the production contract of the training step
  if (tad::knobs::attn_fwd_q64 && d == 64 && q_prescaled && dropout_p == 0.f && out_dtype == TAD_OP16) {
    if (out_lo) hipLaunchKernelGGL((attn_fwd_q64_kernel<true>), grid, dim3(128), 0, (hipStream_t)stream, qkv, (uint16_t*)out, out_lo, lse, N, H, B);
    else hipLaunchKernelGGL((attn_fwd_q64_kernel<false>), grid, dim3(128), 0, (hipStream_t)stream, qkv, (uint16_t*)out, out_lo, lse, N, H, B);
    return check_launch("attn_fwd_q64");
  }
#define LAUNCH_FWD___(H_, O_, Q_, D_, M_) hipLaunchKernelGGL((attn_fwd_kernel<H_, O_, Q_, D_, M_>), grid, block, 0, (hipStream_t)stream, qkv, out, out_lo, lse, N, H, B, scale, drop)
#define LAUNCH_FWD__(O_, Q_, D_, M_) { if (d == 64) LAUNCH_FWD___(64, O_, Q_, D_, M_); else LAUNCH_FWD___(80, O_, Q_, D_, M_); }
#define LAUNCH_FWD_(O_, Q_, M_) { if (dropout_p > 0.f) LAUNCH_FWD__(O_, Q_, true, M_) else LAUNCH_FWD__(O_, Q_, false, M_) }
#define LAUNCH_FWD(M_)                                                                                \
  {                                                                                                   \
    if (out_dtype == TAD_OP16) { if (q_prescaled) LAUNCH_FWD_(true, true, M_) else LAUNCH_FWD_(true, false, M_) }    \
    else { if (q_prescaled) LAUNCH_FWD_(false, true, M_) else LAUNCH_FWD_(false, false, M_) }         \
    return check_launch("attn_fwd");                                                                  \
  }
#ifdef TAD_GEMM_ABLATION
  if (tad::knobs::attn_dma_mode == 2) LAUNCH_FWD(2)
#endif
  LAUNCH_FWD(0)
#undef LAUNCH_FWD
#undef LAUNCH_FWD_
#undef LAUNCH_FWD__
#undef LAUNCH_FWD___
}
