// The four-wave instantiations of gemm_nt_kernel (csrc/gemm.hip, "W4"): 256 x 256 tile, 2 x 2 waves of 128 x 128 outputs, one wave per
// SIMD.  A wave's 256 accumulator registers only fit the AGPR half of gfx950's unified register file, so this translation unit is
// compiled WITHOUT -mllvm -amdgpu-mfma-vgpr-form=1 (simple_tad_amd/build.py), which the rest of the library uses to keep its (at most 128)
// accumulators in arch VGPRs.  The kernel template, its epilogues and the parameter block are those of gemm.hip; only the K loop differs.
#define TAD_GEMM_W4_TU
#include "gemm.hip"
// (gemm.hip's TAD_NAMESPACE_BEGIN is still open here: the part of that file that closes it is skipped under TAD_GEMM_W4_TU)

// ------------------------------------------------------------------------------------------------------------------------------------
// gemm_tn with four waves (weight gradients): slab[s][n][k] = sum over the rows m of split s of P[m][n] * Q[m][k], one 256 x 256 tile of
// (n, k) per workgroup, 128 x 128 per wave, one wave per SIMD.  The epilogue of this GEMM is a plain store of the accumulators (the
// slabs are summed by reduce_dw_kernel) and a workgroup runs 50-200 reduction tiles, so nothing but the loop matters -- the case the
// four-wave loop is made for.  Loop structure = W4 of gemm_nt_kernel: two fragment sets (k-step 0 / 1 of a 64-row reduction tile); while
// the 64 MFMAs of one k-step run, the 32 transposed 8-byte reads of the next one are issued four behind each group of eight MFMAs;
// one counted wait + ONE barrier per reduction tile, at the half; the reduction tile after next is requested (two 1-KiB LDS-DMA pieces
// behind each group) into the ring slot that barrier has just freed.  LDS images, swizzle and fragment layout are gemm_tn_kernel's.
// Results are bit-identical to gemm_tn_kernel<256, 256, 2, 4, 2> (same instruction, same order over the reduction rows).
// The 16 bias-sum accumulators do not fit the AGPR half beside the 256 of the tile, and this translation unit's MFMAs are selected in
// their AGPR form: as a builtin the four bias MFMAs per half made the compiler shuttle accumulators between the two register files around
// them.  Written as inline asm with VGPR operands the instruction is encoded in its VGPR form (destination = third source).
// The compiler does not know that the statement is a matrix instruction, so none of its hazard handling applies: the operands must be
// registers that no vector instruction writes shortly before (RAW) or after (WAR: the matrix pipe reads its sources over several
// cycles) the statement -- the callers pass fragment registers as the LDS delivered them and a constant made opaque at kernel entry;
// the leading / trailing s_nop cover what is left.
__device__ __forceinline__ void mfma_16x16x32_vgpr(f32x4& d, const op16x8& a, const op16x8& b) {
#ifdef TAD_OPND_F16
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 3" : "+v"(d) : "v"(a), "v"(b));
#else
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 3" : "+v"(d) : "v"(a), "v"(b));
#endif
}

// timing experiments (compile-time, wrong results): the loop without its LDS-DMA pieces / fragment reads / wait + barrier
#ifndef TNW4_ABL_NO_DMA
#define TNW4_ABL_NO_DMA 0
#endif
#ifndef TNW4_ABL_NO_READS
#define TNW4_ABL_NO_READS 0
#endif
#ifndef TNW4_ABL_NO_SYNC
#define TNW4_ABL_NO_SYNC 0
#endif
#ifndef TNW4_2BAR
#define TNW4_2BAR 1  // 1: TWO barriers per reduction tile (top and half) and ONE LDS-DMA piece behind every fragment group of BOTH halves -- the
                     // k-step-0 rows of a ring slot are released at the top of the tile, the k-step-1 rows at the half, so the tile after next
                     // streams into them as they become free and every wait leaves a whole tile's pieces (16) in flight; 0: one barrier, the 16
                     // pieces behind the groups of the second half, vmcnt(0)
#endif
#ifndef TNW4_DMA_GROUPS
#define TNW4_DMA_GROUPS 8  // groups of the second half over which the 16 LDS-DMA pieces of a reduction tile are spread: 8, 4 or 2
#endif
__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(const GemmTN p) {
  constexpr int NW = 4, BM = 256, BN = 256, PROW = BM * 2, QROW = BN * 2, P_BYTES = BK * PROW, Q_BYTES = BK * QROW, STAGE_BYTES = P_BYTES + Q_BYTES;
  static_assert(STAGE_BYTES == 65536 && PROW == 512, "slot toggle = bit 16 of the LDS address");
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = (p.K + BN - 1) / BN;
  const int tiles = tiles_k * ((p.N + BM - 1) / BM);
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lin / tiles;
  const int tile = lin - split * tiles;
  const int tn_ = tile / tiles_k, tk_ = tile - tn_ * tiles_k;
  const int n0 = tn_ * BM, k0 = tk_ * BN;
  const int mr0 = split * p.rows_per_split;
  const int nt = p.rows_per_split / BK;  // >= 2 (launcher)
  // PAIR (GemmTN::N1): tiles of output rows >= N1 belong to the second problem -- its own operands, P with its own leading dimension
  const bool seg2 = p.N1 > 0 && n0 >= p.N1;  // (uniform)
  const uint16_t* const Pg = seg2 ? p.P2 : p.P;
  const uint16_t* const Qg = seg2 ? p.Q2 : p.Q;
  const int ldp = p.N1 > 0 ? (seg2 ? p.N - p.N1 : p.N1) : p.N;  // columns of this tile's P operand
  const int np0 = seg2 ? n0 - p.N1 : n0;                        // the tile's first column in it
  const int p_bytes = (int)((int64_t)p.Mr * ldp * 2), q_bytes = (int)((int64_t)p.Mr * p.K * 2);
  // DMA: piece = 1 KiB = 2 rows of 512 B; piece i of wave w = rows 2 (4 i + w), + 1; lane -> row lane / 32, 16-byte chunk lane % 32
  uint32_t p_off[8], q_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (i * NW + wave) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ sw_tn(row);
    int col = np0 + chunk * 8;
    if (col > ldp - 8) col = ldp - 8;  // (columns beyond N only feed outputs that are never stored; the clamp keeps the address in the row)
    p_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(ldp * 2) + (uint32_t)(col * 2);
    col = k0 + chunk * 8;
    if (col > p.K - 8) col = p.K - 8;
    q_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(p.K * 2) + (uint32_t)(col * 2);
  }
#define TNW4_PIECE(SLOT, idx, t_)                                                                                                            \
  if ((idx) < 8) w4_dma_piece(Pg, p_bytes, lds + (SLOT) + (((idx) & 7) * NW + wave) * 1024, p_off[(idx) & 7], (uint32_t)(t_) * BK * (uint32_t)(ldp * 2)); \
  else w4_dma_piece(Qg, q_bytes, lds + (SLOT) + P_BYTES + (((idx) & 7) * NW + wave) * 1024, q_off[(idx) & 7], (uint32_t)(t_) * BK * (uint32_t)(p.K * 2))
  // transposed fragment reads (gemm_tn_kernel): 16-lane group g covers reduction rows 8 g .. + 7 of a k-step, lane li of the group
  // supplies row li >> 2 (+ 4 for the second read), columns c0 + 4 (li & 3) .. + 3
  const int g = lane >> 4, li = lane & 15;
  uint32_t p_rd[8], q_rd[8];  // address of fragment i's first read, k-step 0, in the slot being read (toggled by v_xor)
  {
    const int r0 = 8 * g + (li >> 2);
    const uint32_t l0 = lds_addr(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // (register slot i of the P side holds row fragment i ^ wn: the bias MFMAs below then take slots 0, 2, 4, 6 in BOTH waves of a
      //  row pair -- wave wn sums the fragments of parity wn -- without a select or a branch in front of them)
      const int cp = wm * 128 + 16 * (i ^ wn) + 4 * (li & 3), cq = wn * 128 + 16 * i + 4 * (li & 3);
      p_rd[i] = l0 + (uint32_t)(r0 * PROW + (((cp >> 3) ^ sw_tn(r0)) << 4) + (cp & 7) * 2);
      q_rd[i] = l0 + (uint32_t)P_BYTES + (uint32_t)(r0 * QROW + (((cq >> 3) ^ sw_tn(r0)) << 4) + (cq & 7) * 2);
    }
  }
  // bias gradient = column sums of P (one extra MFMA per row fragment against an all-ones operand), shared out as in gemm_tn_kernel:
  // the tiles_k workgroups of a row panel take turns over the reduction tiles, the two waves of a row split the fragments
  const bool bias_on = p.bias_slab != nullptr && !seg2;  // (PAIR: only the first problem has bias column sums)
  op16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (op16_t)1.0f;
  asm volatile("" : "+v"(ones));  // (opaque: not re-materialised by a vector move in front of the asm MFMA that reads it)
  f32x4 bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // piece order inside a tile: the k-step-0 rows first (P pieces 0..3, Q pieces 0..3 of this wave), then the k-step-1 rows -- vmcnt counts in
  // issue order, and the k-step-0 rows are needed half a tile earlier
#define TNW4_K0IDX(g) ((g) < 4 ? (g) : 8 + ((g) - 4))
#define TNW4_K1IDX(g) ((g) < 4 ? 4 + (g) : 12 + ((g) - 4))
  static_for<0, 8>([&](auto ic) { TNW4_PIECE(0, TNW4_K0IDX(decltype(ic)::value), 0); });
  static_for<0, 8>([&](auto ic) { TNW4_PIECE(0, TNW4_K1IDX(decltype(ic)::value), 0); });
  static_for<0, 8>([&](auto ic) { TNW4_PIECE(STAGE_BYTES, TNW4_K0IDX(decltype(ic)::value), 1); });
  static_for<0, 8>([&](auto ic) { TNW4_PIECE(STAGE_BYTES, TNW4_K1IDX(decltype(ic)::value), 1); });
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  block_barrier();
  s16x4 pl[2][8], ph[2][8], ql[2][8], qh[2][8];  // [fragment set = k-step][fragment]: the two 8-byte halves of a fragment
#define TNW4_READ(SET, KS, i)                                            \
  ql[SET][i] = lds_tr16_b64<(KS) * 32 * QROW>(q_rd[i]);                  \
  qh[SET][i] = lds_tr16_b64<(KS) * 32 * QROW + 4 * QROW>(q_rd[i]);       \
  pl[SET][i] = lds_tr16_b64<(KS) * 32 * PROW>(p_rd[i]);                  \
  ph[SET][i] = lds_tr16_b64<(KS) * 32 * PROW + 4 * PROW>(p_rd[i])
  static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; TNW4_READ(0, 0, i); });
  int slot = 0;
  auto tn_tile = [&](auto NEXTC, auto NEXT2C, int t) {
    constexpr bool NEXT = decltype(NEXTC)::value, NEXT2 = decltype(NEXT2C)::value;
    const bool bias_now = bias_on && (t % tiles_k == tk_);
    // ---- first half: k-step 0; the k-step-1 fragments of this slot arrive behind the groups
    lds_wait<0>(ql[0][0], qh[0][0], ql[0][1], qh[0][1], ql[0][2], qh[0][2], ql[0][3], qh[0][3]);
    lds_wait<0>(ql[0][4], qh[0][4], ql[0][5], qh[0][5], ql[0][6], qh[0][6], ql[0][7], qh[0][7]);
    lds_wait<0>(pl[0][0], ph[0][0], pl[0][1], ph[0][1], pl[0][2], ph[0][2], pl[0][3], ph[0][3]);
    lds_wait<0>(pl[0][4], ph[0][4], pl[0][5], ph[0][5], pl[0][6], ph[0][6], pl[0][7], ph[0][7]);
    if constexpr (TNW4_2BAR && !TNW4_ABL_NO_SYNC) {
      // top of the tile: every wave holds this tile's k-step-0 fragments -> those rows of the slot are free; this tile's k-step-1 rows
      // (requested two tiles ago, 16 younger pieces behind them) have landed
      if constexpr (NEXT2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      block_barrier();
    }
    static_for<0, 8>([&](auto gc) {
      constexpr int i = decltype(gc)::value;
      if constexpr (TNW4_2BAR && NEXT2 && !TNW4_ABL_NO_DMA) { TNW4_PIECE(slot, TNW4_K0IDX(i), t + 2); }
      if constexpr (!TNW4_ABL_NO_READS) { TNW4_READ(1, 1, i); }
      const op16x8 pf = join_tr(pl[0][i], ph[0][i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = TAD_MFMA_16x16x32(join_tr(ql[0][j], qh[0][j]), pf, acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    // (bias column sums: ONE uniform branch per half, behind its groups -- a branch per group cut the matrix instruction stream into
    //  pieces and made the compiler copy accumulators around it; the fragments of this set are intact until the next tile's reads)
    if (bias_now) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) mfma_16x16x32_vgpr(bacc[ib], join_tr(pl[0][2 * ib], ph[0][2 * ib]), ones);  // (slot 2 ib = fragment 2 ib + wn)
    }
    // ---- second half: k-step 1; the next reduction tile has landed -> its k-step-0 fragments; the tile after next into this slot
    lds_wait<0>(ql[1][0], qh[1][0], ql[1][1], qh[1][1], ql[1][2], qh[1][2], ql[1][3], qh[1][3]);
    lds_wait<0>(ql[1][4], qh[1][4], ql[1][5], qh[1][5], ql[1][6], qh[1][6], ql[1][7], qh[1][7]);
    lds_wait<0>(pl[1][0], ph[1][0], pl[1][1], ph[1][1], pl[1][2], ph[1][2], pl[1][3], ph[1][3]);
    lds_wait<0>(pl[1][4], ph[1][4], pl[1][5], ph[1][5], pl[1][6], ph[1][6], pl[1][7], ph[1][7]);
    if constexpr (NEXT) {
      if constexpr (!TNW4_ABL_NO_SYNC) {
        if constexpr (TNW4_2BAR && NEXT2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // (the next tile's k-step-0 rows; 16 younger pieces stay in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_barrier();
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { p_rd[i] ^= (uint32_t)STAGE_BYTES; q_rd[i] ^= (uint32_t)STAGE_BYTES; }
    }
    static_for<0, 8>([&](auto gc) {
      constexpr int i = decltype(gc)::value;
      if constexpr (NEXT2) {
        // the 16 pieces of the tile after next go out behind the first TNW4_DMA_GROUPS groups (the last piece is needed one half tile
        // + one tile later: spread over all eight groups it had 0.9 us to land)
        if constexpr (TNW4_2BAR) {
          if constexpr (!TNW4_ABL_NO_DMA) { TNW4_PIECE(slot, TNW4_K1IDX(i), t + 2); }
        } else if constexpr (i < TNW4_DMA_GROUPS && !TNW4_ABL_NO_DMA) {
          static_for<0, 16 / TNW4_DMA_GROUPS>([&](auto pc) {
            constexpr int piece = i * (16 / TNW4_DMA_GROUPS) + decltype(pc)::value;
            TNW4_PIECE(slot, piece, t + 2);
          });
        }
      }
      if constexpr (NEXT && !TNW4_ABL_NO_READS) { TNW4_READ(0, 0, i); }
      const op16x8 pf = join_tr(pl[1][i], ph[1][i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = TAD_MFMA_16x16x32(join_tr(ql[1][j], qh[1][j]), pf, acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (bias_now) {
#pragma unroll
      for (int ib = 0; ib < 4; ++ib) mfma_16x16x32_vgpr(bacc[ib], join_tr(pl[1][2 * ib], ph[1][2 * ib]), ones);  // (slot 2 ib = fragment 2 ib + wn)
    }
    slot ^= STAGE_BYTES;
  };
  int t = 0;
  for (; t + 2 < nt; ++t) tn_tile(std::true_type{}, std::true_type{}, t);
  tn_tile(std::true_type{}, std::false_type{}, t);
  tn_tile(std::false_type{}, std::false_type{}, t + 1);
#undef TNW4_PIECE
#undef TNW4_READ

  // Hazard fence for the asm bias MFMAs: the compiler does not know that bacc is written by a matrix instruction, so nothing
  // orders the vector reads of bacc below behind the last of them but the code in between.  A 16 x 16 x 32 MFMA needs 8 passes (32 cycles) before
  // its result may be read by a VALU / store instruction; the trailing s_nop 3 of the statement covers 4.  Budget assumed: 16 + 16 wait states
  // here >= the remaining 28, whatever the compiler schedules between the loop and the stores.
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(bacc[0]), "+v"(bacc[1]), "+v"(bacc[2]), "+v"(bacc[3])::"memory");
  if (bias_on && li == 0) {
    float* bo = p.bias_slab + ((int64_t)split * tiles_k + tk_) * p.N;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * 128 + 16 * (ib * 2 + wn) + 4 * g + r;
        if (n < p.N) bo[n] = bacc[ib][r];
      }
  }
  // D[row = k][col = n]: lane (li, g) holds k = 4 g .. + 3 of output row n = li (gemm_tn_kernel)
  float* out = p.slab + (int64_t)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = n0 + wm * 128 + 16 * (i ^ wn) + li;  // (slot i = row fragment i ^ wn)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + wn * 128 + 16 * j + 4 * g;
      if (n < p.N && k < p.K)
        *reinterpret_cast<float4*>(out + (int64_t)n * p.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

int launch_gemm_tn_w4(const GemmTN& p, int grid, hipStream_t st) {
  hipLaunchKernelGGL(gemm_tn_w4_kernel, dim3(grid), dim3(256), 0, st, p);
  return check_launch("gemm_tn_w4");
}

// grid_persist > 0: persistent launch of that many workgroups (a multiple of 8, one per CU); else one workgroup per tile
int launch_gemm_nt_w4(const GemmNT& p, int grid_persist, hipStream_t st) {
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
#define W4_LAUNCH(E, O)                                                                                                          \
  do {                                                                                                                           \
    if (grid_persist > 0) hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 2, 2, E, O, true, false>), dim3(grid_persist), dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 2, 2, E, O, false, false>), dim3(tiles), dim3(256), 0, st, p);        \
  } while (0)
  switch (p.epi) {
    case EPI_PLAIN: if (p.c_bf16) W4_LAUNCH(EPI_PLAIN, true); else W4_LAUNCH(EPI_PLAIN, false); break;
    case EPI_GELU: if (p.c_bf16) W4_LAUNCH(EPI_GELU, true); else W4_LAUNCH(EPI_GELU, false); break;
    case EPI_DGELU: if (p.c_bf16) W4_LAUNCH(EPI_DGELU, true); else W4_LAUNCH(EPI_DGELU, false); break;
    case EPI_RESIDUAL:
      if (p.c_bf16) { set_error("gemm_nt (4-wave): no 16-bit residual epilogue"); return TAD_EINVAL; }
      W4_LAUNCH(EPI_RESIDUAL, false);
      break;
    default: set_error("gemm_nt (4-wave): no kernel for epilogue %d", p.epi); return TAD_EINVAL;
  }
#undef W4_LAUNCH
  return check_launch("gemm_nt_w4");
}

TAD_NAMESPACE_END
