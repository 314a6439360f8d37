// experiments/r05: the W4D kernel (four waves, direct stores, ONE K-tile stream across tile boundaries) as it stood when it was measured
// and set aside (experiments/README.md, round 5).  Not compiled by simple_tad_amd/build.py.  To rebuild: paste this block into
// simple_tad_amd/csrc/gemm_w4.hip in front of launch_gemm_nt_w4 and route the bias-only persistent launches to it:
//   if (p.epi == EPI_PLAIN && grid_persist > 0 && p.c_bf16 && p.M % 256 == 0 && p.N % 256 == 0 && (p.bias_seg <= 0 || p.bias_seg % 256 == 0) &&
//       p.colscale_cols % 256 == 0) { hipLaunchKernelGGL((gemm_nt_w4d_kernel<true, false>), dim3(grid_persist), dim3(256), 0, st, p); ... }
// TAD_W4D_SPREAD 1 = the measured form (results bit-identical to the eight-wave kernels; qkv forward 174.9 us against 171.9).
#ifndef TAD_W4D_SPREAD
#define TAD_W4D_SPREAD 1
#endif
// ------------------------------------------------------------------------------------------------------------------------------------
// W4D: the four-wave kernel for the bias-only ("plain") epilogue, persistent, with ONE stream of K-tiles across tile boundaries.
//
// A workgroup walks its XCD's tile list as gemm_nt_kernel<PERSIST> does.  The K loop is the W4 loop of gemm.hip (per K-tile and wave 128
// MFMAs 16 x 16 x 32 in two halves; the 16 fragments of the next half arrive behind the groups of the current one; K-tile kt + 2 is
// requested into the ring slot of K-tile kt behind the tile's one barrier) -- but the ring does not stop at the end of a tile: the last
// two K-tiles of a tile request the first two K-tiles of the NEXT tile, the last K-tile reads the next tile's first fragments, and the
// epilogue of a tile runs inside its last K-tile: behind the matrix instructions of fragment group g, group g - 1 (16 output rows x 128
// columns per wave, final since its own last matrix instruction was issued a group earlier) is read out of the accumulator file,
// converted, stored straight from the MFMA layout (16 rows x 64 contiguous bytes per store instruction) and re-initialised with the next
// tile's bias.  Between two tiles the matrix pipe sees one group's epilogue (the eighth) and nothing else: no drain, no prologue, no
// re-start of the DMA stream.  The LDS holds nothing but the ring (128 KiB).
// Results are bit-identical to the eight-wave kernels (same instruction, same order of the K-tiles, bias as the initial accumulator).
// CSCALE: GemmNT::colscale is in use (the q third of the qkv Linear); a compile-time variant so that the other launches carry neither its
// multiplications nor a branch (a branch inside the last K-tile makes the compiler copy accumulators around it).
template <bool OUT_BF16, bool CSCALE>
__global__ __launch_bounds__(256, 1) void gemm_nt_w4d_kernel(const GemmNT p) {
  constexpr int NW = 4, BM = 256, BN = 256, ROWB = BK * 2, RPP = 8, A_BYTES = BM * ROWB, STAGE_BYTES = 2 * A_BYTES;
  static_assert(STAGE_BYTES == 65536, "slot toggle = bit 16 of the LDS address");
  constexpr int BIAS_OFF = 2 * STAGE_BYTES;  // behind the ring: the 256 bias values of a tile's columns (f32), staged by LDS-DMA
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES + 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, kq = lane >> 4;
  // ---- tile list (gemm_nt_kernel, PERSIST)
  const int GROUP_M = p.group_m;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int per_group = GROUP_M * tiles_n;
  const int xcd = blockIdx.x & 7;
  const int t_first = xcd_remap(xcd, tiles_m * tiles_n);
  const int t_end = t_first + (tiles_m * tiles_n >> 3) + ((xcd < ((tiles_m * tiles_n) & 7)) ? 1 : 0);
  const int t_step = gridDim.x >> 3;
  int t_cur = t_first + (blockIdx.x >> 3);
  if (t_cur >= t_end) return;  // (uniform per workgroup: fewer tiles than workgroups on this XCD)
  auto decode = [&](int tile, int& m0, int& n0) {
    const int grp = tile / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_grp = tile - grp * per_group;
    m0 = (first_m + in_grp % gsz) * BM;
    n0 = (in_grp / gsz) * BN;
  };
  const uint32_t lda_b = (uint32_t)(p.K * 2);
  const int a_bytes = (int)((int64_t)p.M * lda_b), b_bytes = (int)((int64_t)p.N * p.K * 2);
  const int nk = p.K / BK;  // >= 2 (launcher)
  // ---- DMA addressing (stage_tile's layout: piece i of wave w = LDS rows 32 i + 8 w .. + 7).  M and N are multiples of 256 here (the
  // launcher's rule), so no piece ever leaves its operand and everything but two lane constants can travel in the instruction's SCALAR
  // offset: row r = 32 i + r0 with r0 = 8 w + lane / 8, and the swizzle of its 16-byte chunk looks at bit 5 of r only through i & 1.
  // (Sixteen per-piece offsets in VGPRs, recomputed per tile, were what this kernel spilled -- and a spill reload waits with vmcnt(0).)
  const int r0 = wave * RPP + (lane >> 3), dchunk = lane & 7;
  const uint32_t la_e = (uint32_t)r0 * lda_b + (uint32_t)((dchunk ^ sw_nt(r0)) * 16);       // pieces 0, 2, 4, 6
  const uint32_t la_o = (uint32_t)r0 * lda_b + (uint32_t)((dchunk ^ sw_nt(r0 + 32)) * 16);  // pieces 1, 3, 5, 7
  uint32_t sA, sB;         // byte offset of the streamed tile's first x / w row
  int a_lim = a_bytes, b_lim = b_bytes;  // descriptor sizes of the stream: 0 once there is no next tile (every piece out of range: nothing fetched)
  auto tile_offsets = [&](int m0_, int n0_, bool valid) {
    sA = (uint32_t)m0_ * lda_b;
    sB = (uint32_t)n0_ * lda_b;
    a_lim = valid ? a_bytes : 0;
    b_lim = valid ? b_bytes : 0;
  };
#define W4D_PIECE(SLOT, idx, kt_)                                                                                                       \
  if ((idx) < 8) w4_dma_piece(p.A, a_lim, lds + (SLOT) + (((idx) & 7) * NW + wave) * 1024, ((idx) & 1) ? la_o : la_e,                   \
                              sA + (uint32_t)(((idx) & 7) * 32) * lda_b + (uint32_t)(kt_) * ROWB);                                       \
  else w4_dma_piece(p.B, b_lim, lds + (SLOT) + A_BYTES + (((idx) & 7) * NW + wave) * 1024, ((idx) & 1) ? la_o : la_e,                   \
                    sB + (uint32_t)(((idx) & 7) * 32) * lda_b + (uint32_t)(kt_) * ROWB)
  // ---- fragment addressing (gemm_nt_kernel): x rows 16 i + c of the wave's 128; w rows permuted so that a lane holds consecutive columns
  uint32_t a_ad[2][4], b_ad[2][4];  // [k-step][fragment & 3] in the slot being read; fragment q + 4 lies 8 KiB further on
  {
    const uint32_t l0 = lds_addr(lds);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ra = wm * 128 + q * 16 + c;
      const int rl = OUT_BF16 ? (32 * (q >> 1) + 8 * (c >> 2) + 4 * (q & 1) + (c & 3)) : (16 * q + 4 * (c >> 2) + (c & 3));
      const int rb = wn * 128 + rl;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        a_ad[ks][q] = l0 + (uint32_t)(ra * ROWB) + (uint32_t)((((4 * ks) + kq) ^ sw_nt(ra)) << 4);
        b_ad[ks][q] = l0 + (uint32_t)A_BYTES + (uint32_t)(rb * ROWB) + (uint32_t)((((4 * ks) + kq) ^ sw_nt(rb)) << 4);
      }
    }
  }
  // output column of accumulator (j, element 0) relative to the wave's first column
  auto col_of = [&](int j) { return OUT_BF16 ? (32 * (j >> 1) + 8 * kq + 4 * (j & 1)) : (16 * j + 4 * kq); };
  // The bias of a tile's 256 columns travels global -> LDS as ONE LDS-DMA piece (wave 0: 64 lanes x 16 bytes) together with the K-tiles
  // -- no registers hold it while the fragments are live -- and is read back, four values at a time, where a fragment group's accumulators
  // are (re-)initialised.  Columns past the operand read as zero through the descriptor's bounds check; the qkv Linear's middle third
  // (no k bias) is a descriptor of size zero.  A tile never straddles two thirds (bias_seg % 256 == 0: the launcher's rule).
  auto request_bias = [&](int n0_, bool valid) {
    if (wave != 0) return;
    const float* src = p.bias;
    int cols = (p.bias && valid) ? p.N - n0_ : 0, first = n0_;
    if (cols > 0 && p.bias_seg > 0) {
      const int third = n0_ / p.bias_seg;
      first = n0_ - third * p.bias_seg;
      src = third == 2 ? p.bias2 : p.bias;
      cols = third == 1 ? 0 : p.bias_seg - first;
    }
    if (cols <= 0) { src = reinterpret_cast<const float*>(p.B); first = 0; cols = 0; }  // (any valid address; nothing is read)
    w4_dma_piece(src + first, cols * 4, lds + BIAS_OFF, (uint32_t)lane * 16u, 0u);
  };
  const uint32_t bias_ad = lds_addr(lds) + (uint32_t)BIAS_OFF + (uint32_t)(wn * 512 + kq * (OUT_BF16 ? 32 : 16));
  // accumulators of fragment group G <- the staged bias (asm reads: the compiler must not see an LDS read here, it would drain the DMA
  // stream in front of it)
  auto init_group = [&](auto GC, f32x4 (&acc_)[8][8]) {
    constexpr int G = decltype(GC)::value;
    static_for<0, 2>([&](auto hc) {  // two batches of four: 16 registers at a time
      constexpr int h = decltype(hc)::value;
      f32x4 b4[4];
      static_for<0, 4>([&](auto jc) {
        constexpr int j = 4 * h + decltype(jc)::value;
        b4[j & 3] = lds_read_b128<f32x4, (OUT_BF16 ? (32 * (j >> 1) + 4 * (j & 1)) * 4 : 64 * j)>(bias_ad);
      });
      lds_wait<0>(b4[0], b4[1], b4[2], b4[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc_[G][4 * h + j] = b4[j];
    });
  };

  int m0, n0;
  decode(t_cur, m0, n0);
  tile_offsets(m0, n0, true);
  static_for<0, 16>([&](auto ic) { W4D_PIECE(0, decltype(ic)::value, 0); });
  request_bias(n0, true);
  static_for<0, 16>([&](auto ic) { W4D_PIECE(STAGE_BYTES, decltype(ic)::value, 1); });
  f32x4 acc[8][8];
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // K-tile 0 of the first tile and its bias (K-tile 1's sixteen pieces stay in flight)
  block_barrier();
  static_for<0, 8>([&](auto gc) { init_group(gc, acc); });
  op16x8 fa[2][8], fb[2][8];
  static_for<0, 8>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    fb[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[0][g & 3]);
    fa[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[0][g & 3]);
  });
  int slot = 0;  // byte offset of the ring slot of the K-tile being computed
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  const uint32_t st_lane = ((uint32_t)(wm * 128 + c) * (uint32_t)p.N + (uint32_t)(wn * 128 + (OUT_BF16 ? 8 : 4) * kq)) * ESZ;  // the lane's part of a store offset
  const auto c_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((uint32_t)p.M * (uint32_t)p.N * ESZ), 0x00020000);

  // ONE copy of the K-tile body for the whole kernel, walked by a flat loop over the K-tiles of all of this workgroup's tiles; what differs
  // between K-tiles is decided by uniform scalars (`pre`: the tile's last but one -- the DMA stream moves on to the next tile; `last`: the
  // tile's last -- fragment group g - 1 is stored behind the matrix instructions of group g).  Separate copies of the body for those cases
  // (tried first) made the register allocator number the 256 accumulators differently in each copy and reconcile them with ~500 copies
  // per tile.
  int em0 = m0, en0 = n0;  // origin of the tile being computed ((m0, n0): of the tile being streamed)
  bool has_next = t_cur + t_step < t_end;
  // epilogue of fragment group G: rows em0 + wm * 128 + 16 G + c, the wave's 128 columns; then the group starts the next tile's sums
  auto store_group = [&](auto GC) {
    constexpr int G = decltype(GC)::value;
    const float cs_tile = (CSCALE && en0 < p.colscale_cols) ? p.colscale : 1.f;  // (scalar; a tile lies inside or outside the scaled range)
    // one lane constant (st_lane) + a scalar offset per store instruction: M and N are multiples of 256, nothing is out of range
    const uint32_t s_row = ((uint32_t)(em0 + 16 * G) * (uint32_t)p.N + (uint32_t)en0) * ESZ;
#pragma unroll
    for (int jj = 0; jj < (OUT_BF16 ? 4 : 8); ++jj) {
      const uint32_t s_off = s_row + (uint32_t)(jj * (OUT_BF16 ? 32 : 16) * ESZ);
      if constexpr (OUT_BF16) {
        f32x4 lo4 = acc[G][2 * jj], hi4 = acc[G][2 * jj + 1];
        if constexpr (CSCALE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo4[e] *= cs_tile; hi4[e] *= cs_tile; }
        }
        const u32x4 v = u32x4{pack_op16x2(lo4[0], lo4[1]), pack_op16x2(lo4[2], lo4[3]), pack_op16x2(hi4[0], hi4[1]), pack_op16x2(hi4[2], hi4[3])};
        if (!(DBG_BITS(p) & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, c_rs, st_lane, s_off, TAD_STORE_AUX);  // (ablation 4, timing only: no stores)
        else asm volatile("" ::"v"(v));
      } else {
        f32x4 v4 = acc[G][jj];
        if constexpr (CSCALE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] *= cs_tile;
        }
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v4[0]), __float_as_uint(v4[1]), __float_as_uint(v4[2]), __float_as_uint(v4[3])}, c_rs, st_lane,
                                               s_off, TAD_STORE_AUX);
      }
    }
    init_group(GC, acc);  // the group starts the next tile's sums (without a next tile the values are not used)
  };
  for (int kt = 0;;) {
    const bool last = kt == nk - 1, pre = kt == nk - 2;
    // ---- first half: k-step 0 products; the k-step-1 fragments of this slot arrive behind the groups
    lds_wait<0>(fb[0][0], fb[0][1], fb[0][2], fb[0][3], fb[0][4], fb[0][5], fb[0][6], fb[0][7]);
    lds_wait<0>(fa[0][0], fa[0][1], fa[0][2], fa[0][3], fa[0][4], fa[0][5], fa[0][6], fa[0][7]);
    static_for<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      fb[1][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[1][g & 3]);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] = TAD_MFMA_16x16x32(fb[0][j], fa[0][g], acc[g][j]);
      __builtin_amdgcn_sched_barrier(0);
      // (behind the group: its x fragment is dead, so the k-step-1 fragment can take its registers -- 36 instead of 64 for the x side)
      fa[1][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[1][g & 3]);
    });
    // ---- second half: k-step 1 products.  The K-tile behind this one (of this tile, or the next tile's first) has landed and every wave
    // has this slot's fragments in registers: the slot takes the K-tile after that one
    lds_wait<0>(fb[1][0], fb[1][1], fb[1][2], fb[1][3], fb[1][4], fb[1][5], fb[1][6], fb[1][7]);
    lds_wait<0>(fa[1][0], fa[1][1], fa[1][2], fa[1][3], fa[1][4], fa[1][5], fa[1][6], fa[1][7]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    block_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) { a_ad[ks][q] ^= (uint32_t)STAGE_BYTES; b_ad[ks][q] ^= (uint32_t)STAGE_BYTES; }
    if (pre) {  // from here on the DMA stream belongs to the next tile (none: descriptors of size zero, nothing is fetched)
      decode(has_next ? t_cur + t_step : t_cur, m0, n0);
      tile_offsets(m0, n0, has_next);
      request_bias(n0, has_next);
    }
    const int ktn = kt + 2 < nk ? kt + 2 : kt + 2 - nk;  // K-tile (of the streamed tile) that goes into this slot
    static_for<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      W4D_PIECE(slot, 2 * g, ktn);
      W4D_PIECE(slot, 2 * g + 1, ktn);
      fb[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[0][g & 3]);  // (last K-tile of the last tile: unused)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] = TAD_MFMA_16x16x32(fb[1][j], fa[1][g], acc[g][j]);
      __builtin_amdgcn_sched_barrier(0);
      fa[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[0][g & 3]);
#if TAD_W4D_SPREAD
      if constexpr (g >= 1) {
        if (__builtin_expect(last, 0)) store_group(std::integral_constant<int, g - 1>{});
      }
#endif
    });
    slot ^= STAGE_BYTES;
    if (__builtin_expect(last, 0)) {
#if !TAD_W4D_SPREAD
      static_for<0, 7>([&](auto gc) { store_group(gc); });
#endif
      store_group(std::integral_constant<int, 7>{});
      if (!has_next) break;
      t_cur += t_step;
      has_next = t_cur + t_step < t_end;
      em0 = m0;
      en0 = n0;
      kt = 0;
    } else {
      ++kt;
    }
  }
#undef W4D_PIECE
}

