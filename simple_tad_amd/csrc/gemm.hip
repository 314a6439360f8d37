// 16-bit-operand (bf16 / f16, see common.h) MFMA GEMMs for the Linear layers of the Video-ViT path (gfx950, wave64).
//
//  gemm_nt : C[M,N] = A[M,K] * B[N,K]^T (+ fused epilogue)      forward Linear and input-gradient (with W^T)
//  gemm_tn : C[N,K] = P[Mr,N]^T * Q[Mr,K]  (split over Mr)        weight gradient
//
// Both stage 64-deep K-tiles global -> LDS with 16-byte LDS-DMA loads (buffer_load ... lds; out-of-range rows
// read as zero through the buffer descriptor's bounds check), double-buffered, one barrier per K-tile, and use
// v_mfma_f32_16x16x32_{bf16,f16}.  The LDS images are XOR-swizzled on the 16-byte chunk index (the swizzle is applied
// to the per-lane *source* address because the DMA destination is lane-linear) so that every ds_read_b128 /
// ds_read_b64_tr_b16 below is bank-conflict free (tools/lds_bank_sim.py).
//
// gemm_nt computes C^T tiles (W rows feed the MFMA "A" operand) with a permuted assignment of W rows to fragment
// lanes, so that each lane ends up holding 4 (f32) or 8 (bf16) *consecutive* output columns of one output row.  Above
// 1.5 tiles per CU it runs persistently (one workgroup per CU walks a tile list and fetches the next tile's first
// K-tile under the current epilogue); the launcher picks the tile shape -- or whole rounds of 256 x 256 tiles plus a
// second launch for the remaining rows -- from a small cost model.  Epilogue (bias / GELU / residual / layer-scale /
// drop-path scale / GELU-backward): accumulators transposed through the LDS so that global accesses cover whole rows,
// raw buffer loads / stores without per-lane branches, what it reads fetched one chunk ahead; bias-only bf16 outputs are
// stored straight from the MFMA layout instead.  DESIGN.md section 3 and docs/DESIGN_HISTORY.md section 3.1 have the measurements behind each choice.
#include <math.h>
#include <stdlib.h>
#include <string>
#include "common.h"

TAD_NAMESPACE_BEGIN

int launch_reduce_partials(const float* partial, float* out, int splits, int64_t n, int accumulate, hipStream_t st);
int launch_reduce_col_ranges(const float* partial, int N, int splits, int c0, float* out0, int c1, float* out1, int n, int accumulate,
                             hipStream_t st);
int launch_reduce_dw(const float* partial, float* out, int splits, int64_t n, int accumulate, const float* partial2, int rows2, int n2,
                     float* out2a, float* out2b, int n2a, int c2b, hipStream_t st);
int launch_reduce_dw_pair(const float* partial, float* outA, float* outB, int64_t nA, int splits, int64_t n, int accumulate, const float* partial2,
                          int rows2, int n2, float* out2a, float* out2b, int n2a, int c2b, hipStream_t st);

// EPI_RESMOD = EPI_RESIDUAL with the residual row taken modulo res_mod ("+ pos_embed" of the patch embedding): a variant of its
// own so that the integer division stays out of the Linear kernels
// Ablation switches (TAD_GEMM_DEBUG, see GemmNT::debug) cost scalar branches inside the K loops: compiled in only with
// -DTAD_GEMM_ABLATION (python -m simple_tad_amd.build reads TAD_BUILD_ABLATION=1); production builds see a constant 0.
#ifdef TAD_GEMM_ABLATION
#define DBG_BITS(p) ((p).debug)
#else
#define DBG_BITS(p) 0
#endif

enum { EPI_PLAIN = 0, EPI_GELU = 1, EPI_RESIDUAL = 2, EPI_DGELU = 3, EPI_RESMOD = 4 };

struct GemmNT {
  const uint16_t* A;  // [M,K]
  const uint16_t* B;  // [N,K]
  void* C;            // [M,N] f32 or bf16
  const float* bias;  // [N] or null
  const float* bias2; // with bias_seg > 0: columns [0, seg) take bias[n], [2 seg, 3 seg) take bias2[n - 2 seg], the rest 0 (qkv Linear)
  int bias_seg;
  float colscale;          // EPI_PLAIN: columns [0, colscale_cols) are multiplied by colscale before the one rounding to the output
  int colscale_cols;       // type (0 = off; a multiple of 8).  The q third of the qkv Linear: q * scale * log2(e) (tad_linear_fwd_qkv)
  const float* residual;   // [M or res_mod, N] f32 or null
  const float* gamma;      // [N] or null
  const float* rowscale;   // [ceil(M/rows_per_scale)] or null
  uint16_t* preact;        // [M,N] bf16 or null (EPI_GELU)
  const uint16_t* dgelu_h; // [M,N] bf16 (EPI_DGELU)
  int rows_per_scale;
  int res_mod;  // >0: residual row index = (row_base + m) % res_mod (pos_embed broadcast over the batch)
  int row_base;  // row of the whole problem that this launch's row 0 is (a launch may cover a row range of a Linear): rowscale / res_mod
  int c_bf16;
  int epi;
  int M, N, K;
  int debug;  // ablation (TAD_GEMM_DEBUG, timing only, wrong results): 1 = no DMA inside the K loop, 2 = no MFMA, 4 = no epilogue,
              // 8 = (gemm_tn) no fragment reads and no MFMA: staging and barriers only, 16 = (gemm_tn) unswizzled DMA source
  // split-K launch (SPLITK kernels: the under-filled last round of a Linear, see launch_gemm_nt): every 256 x 256 tile is computed by
  // sk_splits workgroups, each over its share of the K-tiles; sk_ws holds their f32 partial tiles [tile][split][256][256], sk_cnt one
  // arrival counter per tile (zeroed by the launcher), sk_err a word IN PINNED HOST MEMORY that is set if a wait gave up: the host looks at
  // it at the start of every later Linear launch and fails that call loudly (launch_gemm_nt)
  float* sk_ws;
  unsigned* sk_cnt;
  unsigned* sk_err;
  int sk_splits;
  int sk_mode;  // 0: partial tiles combined inside the launch (arrival counters); 1: this launch only leaves the partial tiles (no wait);
                // 2: this launch only combines what a mode-1 launch left (no K loop) -- the "deferred" split-K plan, see launch_gemm_nt
  int group_m;        // tile raster: row panels swept per column panel before moving to the next column panel (L2 reuse)
  unsigned long long* stamps;  // debug timeline (tad_linear_debug_stamps): per workgroup 64 slots of 4 x s_memrealtime, or null
};

int launch_gemm_nt_w4(const GemmNT& p, int grid_persist, hipStream_t st);  // csrc/gemm_w4.hip: the four-wave 256 x 256 kernels
struct GemmTN;
int launch_gemm_tn_w4(const GemmTN& p, int grid, hipStream_t st);

constexpr int BK = 64;             // K-tile depth (bf16 elements) -> 128-byte LDS rows
constexpr int ROW_BYTES = BK * 2;  // 128

// swizzle of the 16-byte chunk index (0..7) within a 128-byte LDS row
__device__ __forceinline__ int sw_nt(int row) { return ((row >> 1) & 7) ^ (((row >> 4) & 3) << 1); }

// one wave issues PIECES 1-KiB LDS-DMA pieces: piece i of wave w lands at tile + (i*NW + w)*1024; off[i] is the lane's byte
// offset into the buffer resource (already swizzled), `add` the per-tile advance
// SCALAR_ADD: `add` goes into the instruction's scalar offset instead of a v_add per piece (8 short-lived VGPRs at the point of the
// loop where the fragments are live too: the DGELU / residual 256 x 256 kernels spilled there, and the reload's vmcnt(0) drained the
// DMA).  Only legal when off[i] alone decides whether the access is inside the operand -- true for gemm_nt, whose `add` moves along
// a row (rows >= M have off[i] >= gbytes already), not for gemm_tn, whose `add` moves down the rows -- so the bounds check gives the
// same answer whether or not the hardware includes the scalar offset in it.
template <int PIECES, int NW, bool SCALAR_ADD = false>
__device__ __forceinline__ void stage_tile(const void* gbase, int gbytes, char* tile, const uint32_t* off, uint32_t add, int wave) {
  // descriptor over the whole operand: reads past the end return 0 (rows beyond M / N)
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gbase), 0, gbytes, 0x00020000);
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    if (SCALAR_ADD) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * NW + wave) * 1024), 16, off[i], add, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (i * NW + wave) * 1024), 16, off[i] + add, 0, 0, 0);
  }
}

#ifndef TAD_STORE_AUX
#define TAD_STORE_AUX 2  // cache policy of the gemm_nt output stores: 2 = nt (streamed: the 77-308 MB outputs do not displace the operand panels the
                         // other workgroups of the XCD are re-reading; measured 691.3 -> 695.8 clips/s over four alternating pairs of runs; 0 = default,
                         // 16 = sc1 measured neutral)
#endif
#ifndef TAD_EPI_DB
#define TAD_EPI_DB 0  // 1: double-buffered transposition chunks in the persistent 256 x 256 kernel's LDS epilogues (see EPI_DB).  Measured null in
                      // round 4 (tools/ab_gemm.py, eight in-model shapes: sum 1537.0 vs 1534.8 us; fc1 -3, dX(fc2) +3.5): the epilogue is bound by its
                      // vector work (GELU polynomials, conversions), not by the chunk barriers -- off
#endif
#ifndef TAD_W4_PREFETCH
#define TAD_W4_PREFETCH 0  // (experiment, round 5; 4 or 8 to build it) four-wave gemm_nt kernels: each wave touches one cache line per row of the x panel's K-tile
                           // kt + 2 + TAD_W4_PREFETCH (a 4-byte LDS-DMA into a sink) right behind the LDS-DMA pieces of K-tile kt + 2, so that the lines are in the
                           // L2 when their own pieces ask for them -- for the bias-only Linears with K >= 2560 (an x panel of 1.2 MB and more per tile row, streamed
                           // from beyond the last-level cache).  Bit-identical.  Measured (tools/ab_gemm.py, 15 rounds): dX(fc1) 214.7 -> 204.4 us, every other
                           // shape 0-1.5 % slower (the wait that leaves the touch in flight), block sum 1465.9 -> 1460.5; the step 722.7 -> 723.0 clips/s over
                           // three alternations: nothing.  With the touch on every shape the K = 768 / 2304 ones lose 2 %; distance 8 is no better.  Off
#endif
#ifndef TAD_NT_PEEL
#define TAD_NT_PEEL 1  // last K-tile of the persistent bias-only 16-bit kernel peeled (see PEEL in gemm_nt_kernel); 0 = the round-1 schedule
#endif

// One 1-KiB LDS-DMA piece of the four-wave kernel (W4 in gemm_nt_kernel): `dst` = LDS address of the piece, `off` = the lane's byte offset
// into the operand (swizzled), `add` = the K-tile's advance along the row (scalar offset: see SCALAR_ADD above).  A function, not a macro
// inside the kernel's nested generic lambdas: with the builtin called there, the HOST pass of hipcc dropped the kernel's launch stub
// without a diagnostic.
__device__ __forceinline__ void w4_dma_piece(const void* gbase, int gbytes, char* dst, uint32_t off, uint32_t add) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gbase), 0, gbytes, 0x00020000), LDS_PTR(dst), 16, off, add, 0, 0);
}

// wait until at most `stages_in_flight` later stages (LOADS DMA instructions each, per wave) plus EXTRA younger vector-memory
// instructions are still outstanding
template <int LOADS, int EXTRA = 0>
__device__ __forceinline__ void wait_stage(int stages_in_flight) {
  static_assert(3 * LOADS + EXTRA <= 63, "vmcnt immediate is 6 bits");
  if (stages_in_flight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS + EXTRA) : "memory");
  else if (stages_in_flight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS + EXTRA) : "memory");
  else if (stages_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + EXTRA) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXTRA) : "memory");
}
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// barrier between LDS producers and consumers that leaves global stores / loads in flight
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// PERSIST: one workgroup per CU walks a strided list of tiles (see the comment at the tile loop).
// DIRECT: the epilogue runs on the accumulator registers and stores straight from the MFMA layout (16 rows x 64 contiguous bytes
// per store instruction); otherwise the accumulators are transposed through the LDS first (whole rows per instruction).
// SPLITK: one workgroup = one SHARE of a tile's K-tiles (grid = tiles x p.sk_splits, all of them resident at once: the launcher keeps
// the grid within one workgroup per CU).  The workgroup leaves its f32 partial tile in p.sk_ws, announces it on the tile's arrival
// counter, waits until all shares of the tile are there, and then finishes ITS rows of the tile: sum over the shares, epilogue,
// store.  Nobody waits before publishing, so the wait cannot deadlock while the grid is resident; it is bounded all the same.
// The hand-off follows cdna_hip_programming.md Guideline 16 (plain stores, every wave's vmcnt(0), barrier, one lane's agent-scope
// release, counter add; one relaxed poll, one agent-scope acquire, barrier, plain loads) and depends on no placement.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int EPI, bool OUT_BF16, bool PERSIST, bool DIRECT, bool SPLITK = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 1) void gemm_nt_kernel(const GemmNT p) {
  static_assert(!SPLITK || (!PERSIST && !DIRECT && BM == 256 && BN == 256 && (EPI == EPI_PLAIN || EPI == EPI_RESIDUAL)), "split-K variant");
  constexpr int BKT = BK;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr bool IS_RES = (EPI == EPI_RESIDUAL || EPI == EPI_RESMOD);
  // W4: FOUR waves, one per SIMD, 128 x 128 outputs each (256 accumulator registers in the AGPR half of the unified file, so this
  // instantiation lives in a translation unit of its own, csrc/gemm_w4.hip, compiled without -amdgpu-mfma-vgpr-form) and a K loop whose
  // order of LDS reads, LDS-DMA pieces and matrix instructions is written out by hand (see W4 below).  Against the 8-wave form a
  // K-tile needs a third fewer LDS fragment bytes per matrix instruction (32 x 16-byte reads per 128 MFMAs instead of 24 per 64).
  constexpr bool W4 = NW == 4 && BM == 256 && BN == 256 && STAGES == 2 && !SPLITK && !DIRECT;
  constexpr int ROWB = BKT * 2;               // bytes per LDS row
  constexpr int RPP = 1024 / ROWB;            // rows per 1-KiB DMA piece
  constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
  constexpr int KSTEPS = BKT / 32;
  constexpr int LOADS = BM / (RPP * NW) + BN / (RPP * NW);
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MREP = WTM / 16, NREP = WTN / 16;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "tile rows must split into whole DMA pieces per wave");
  // one epilogue chunk: CROWS rows of f32, padded stride.  The persistent kernel keeps ring slot 0 out of the epilogue's way
  // (the next tile's first K-tile lands there meanwhile), so its chunks must fit the LDS behind slot 0.
  // EPI_DB (persistent 256 x 256 kernel): TWO chunk buffers of 32 rows, the accumulators of chunk q + 1 are dropped into one while the row
  // pass of chunk q reads the other -- one barrier per chunk instead of two, and the LDS writes run beside the row pass
  constexpr bool EPI_DB = TAD_EPI_DB && PERSIST && !DIRECT && !SPLITK && BM == 256 && BN == 256 && NW == 8;
  constexpr int CROWS = EPI_DB ? 32 : BM < 128 ? BM : BM == 192 ? 64 : (((BN > 128 || (NW == 4 && BM == 256)) && IS_RES && !OUT_BF16) ? 32 : (((PERSIST && BN > 128) || NW == 4) ? 64 : 128));
  constexpr int EPI_OFF = PERSIST ? STAGE_BYTES : 0;
  constexpr int EPI_BYTES = DIRECT ? 0 : (EPI_DB ? 2 : 1) * CROWS * (BN * 4 + 16);
  constexpr int LDS_MAIN = STAGES * STAGE_BYTES > EPI_OFF + EPI_BYTES ? STAGES * STAGE_BYTES : EPI_OFF + EPI_BYTES;
  constexpr bool W4PF = W4 && TAD_W4_PREFETCH > 0;                       // L2 prefetch of the x panel (see TAD_W4_PREFETCH)
  constexpr int PF_SINK = (LDS_MAIN + 1023) / 1024 * 1024;               // 4 x 256 bytes nobody reads
  constexpr int LDS_BYTES = W4PF ? PF_SINK + 1024 : LDS_MAIN;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // Tile order: the XCD remap gives each XCD (private 4 MiB L2) a contiguous range of logical tile ids; inside that range the
  // ids sweep GROUP_M row-panels for one column-panel before moving to the next column-panel, so the ~32 workgroups that are
  // resident on an XCD at any time touch only GROUP_M A-panels and ~32/GROUP_M W-panels (both stay L2-resident).
  //
  // PERSIST: the grid is one workgroup per CU (a multiple of 8).  Workgroup (xcd = blockIdx & 7, j = blockIdx >> 3) walks the
  // ids first + j, first + j + step, ... of its XCD's range, so at any time the XCD works on ~step consecutive ids as above.
  // What is gained over one launch-scheduled workgroup per tile: the first K-tile of the next tile (with the register-layout
  // epilogue: its whole prologue) is fetched under the epilogue, and no workgroup launch sits between two tiles.
  const int GROUP_M = p.group_m;  // row panels per column-panel group (launch_nt_variant; tad_linear_tuning("group_m"))
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int per_group = GROUP_M * tiles_n;
  int t_cur, t_end, t_step, t_first = 0;
  const int xcd = blockIdx.x & 7;
  if (PERSIST) {
    const int j = blockIdx.x >> 3;
    t_first = xcd_remap(xcd, tiles_m * tiles_n);  // first id of this XCD's range
    t_cur = t_first + j;
    t_end = t_first + (tiles_m * tiles_n >> 3) + ((xcd < ((tiles_m * tiles_n) & 7)) ? 1 : 0);
    t_step = gridDim.x >> 3;
  } else if (SPLITK) {
    // logical id = tile * splits + share: the XCD remap keeps consecutive logical ids -- the shares of one tile -- on one XCD
    // (speed only: the partial tiles then travel through one L2)
    t_cur = xcd_remap(blockIdx.x, gridDim.x) / p.sk_splits;
    t_end = t_cur + 1;
    t_step = 1;
  } else {
    t_cur = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    t_end = t_cur + 1;
    t_step = 1;
  }
  const int sk_share = SPLITK ? xcd_remap(blockIdx.x, gridDim.x) % p.sk_splits : 0;
  int m0, n0;
#define DECODE_TILE(tile)                                  \
  {                                                        \
    const int grp = (tile) / per_group;                    \
    const int first_m = grp * GROUP_M;                     \
    const int gsz = min(tiles_m - first_m, GROUP_M);       \
    const int in_grp = (tile) - grp * per_group;           \
    m0 = (first_m + in_grp % gsz) * BM;                    \
    n0 = (in_grp / gsz) * BN;                              \
  }

  // (debug 128, timing only: A rows 128 bytes further apart than K elements -- the caller over-allocates A -- to see what the row stride costs)
  const uint32_t lda_b = (uint32_t)(p.K * 2) + ((DBG_BITS(p) & 128) ? 128u : 0u);
  const int a_bytes = (int)((int64_t)p.M * lda_b), b_bytes = (int)((int64_t)p.N * p.K * 2);

  // ---- DMA addressing: one wave-instruction fills RPP LDS rows (1 KiB); lane -> (row lane/CPR, physical chunk lane%CPR)
  const int drow = lane / CPR, dchunk = lane % CPR;
  uint32_t a_off[BM / (RPP * NW)], b_off[BN / (RPP * NW)];
  uint32_t pf_off = 0;  // (W4PF) byte offset of the x row this lane touches: the wave's 64 rows, one per lane
#define TILE_OFFSETS()                                                                                                 \
  {                                                                                                                    \
    if constexpr (W4PF) pf_off = (EPI == EPI_PLAIN && p.K >= 2560) ? (uint32_t)(m0 + ((lane & 7) * NW + wave) * RPP + (lane >> 3)) * lda_b : 0x80000000u; /* (out of range: dropped) */ \
    _Pragma("unroll") for (int i = 0; i < BM / (RPP * NW); ++i) {                                                      \
      const int row = (i * NW + wave) * RPP + drow;                                                                    \
      a_off[i] = (uint32_t)(((DBG_BITS(p) & 64) ? 0 : m0) + row) * lda_b + (uint32_t)((dchunk ^ sw_nt(row)) * 16);  /* (debug 64, timing only: every tile reads the A rows of tile 0 -- A always L2-resident) */ \
    }                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < BN / (RPP * NW); ++i) {                                                      \
      const int row = (i * NW + wave) * RPP + drow;                                                                    \
      b_off[i] = (uint32_t)(n0 + row) * (uint32_t)(p.K * 2) + (uint32_t)((dchunk ^ sw_nt(row)) * 16);           \
    }                                                                                                                  \
  }
#define STAGE_NT(buf, kt) \
  stage_tile<BM / (RPP * NW), NW, true>(p.A, a_bytes, lds + (buf) * STAGE_BYTES, a_off, (uint32_t)((kt) + kt0) * ROWB, wave); \
  stage_tile<BN / (RPP * NW), NW, true>(p.B, b_bytes, lds + (buf) * STAGE_BYTES + A_BYTES, b_off, (uint32_t)((kt) + kt0) * ROWB, wave)

  // ---- fragment addressing
  const int c = lane & 15, kq = lane >> 4;
  // A-operand rows (output rows m): 16 consecutive rows per m-rep
  uint32_t a_rd[MREP];
  int a_sw[MREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int row = wm * WTM + i * 16 + c;
    a_rd[i] = row * ROWB;
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
    const int rl = OUT_BF16 ? (32 * (j >> 1) + 8 * (c >> 2) + 4 * (j & 1) + (c & 3)) : (16 * j + 4 * (c >> 2) + (c & 3));
    const int row = wn * WTN + rl;
    b_rd[j] = row * ROWB;
    b_sw[j] = sw_nt(row);
  }

  const int nk_all = p.K / BKT;
  const int kt0 = SPLITK ? sk_share * nk_all / p.sk_splits : 0;                       // this workgroup's K-tiles: [kt0, kt0 + nk)
  const int nk = SPLITK ? (p.sk_mode == 2 ? 0 : (sk_share + 1) * nk_all / p.sk_splits - kt0) : nk_all;
#define FRAG_A(dst, base, ks) \
  _Pragma("unroll") for (int i = 0; i < MREP; ++i)  \
      dst[i] = *reinterpret_cast<const op16x8*>((base) + a_rd[i] + ((((KSTEPS > 1 ? 4 * (ks) : 0) + kq) ^ a_sw[i]) << 4))
#define FRAG_B(dst, base, ks) \
  _Pragma("unroll") for (int j = 0; j < NREP; ++j)  \
      dst[j] = *reinterpret_cast<const op16x8*>((base) + b_rd[j] + ((((KSTEPS > 1 ? 4 * (ks) : 0) + kq) ^ b_sw[j]) << 4))
#define MFMA_BLOCK(afr, bfr_)                                                                            \
  if (!(DBG_BITS(p) & 2)) {                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < NREP; ++j)                                                 \
            acc[i][j] = TAD_MFMA_16x16x32(bfr_[j], afr[i], acc[i][j]);    \
  } else {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i) asm volatile("" ::"v"(afr[i]));                     \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j) asm volatile("" ::"v"(bfr_[j]));                    \
  }
  const bool late = wave >= NW / 2;  // wave-uniform (scalar branches); the MFMA code is shared by both halves
  // epilogue geometry (see the epilogue below)
  constexpr int MREP_C = CROWS / (16 * WAVES_M);     // m-fragments each wave contributes to a chunk
  constexpr int NCHUNK = MREP / MREP_C;
  constexpr int CSTRIDE = BN * 4 + 16;               // padded row stride (bytes): conflict-free 16-byte writes
  constexpr int CPL = OUT_BF16 ? 8 : 4;              // columns per lane in the row pass (16-byte stores)
  constexpr int LPR = BN / CPL, RPI = 64 / LPR;      // lanes per row, rows per wave-instruction
  constexpr int NR = CROWS / (NW * RPI);             // row-instructions per wave per chunk
  constexpr int BATCH_MAX = (IS_RES && BN > 128) ? 2 : 4;  // rows of LDS reads in flight per lane (register budget: the other chunks' accumulators are live)
  constexpr int BATCH = NR < BATCH_MAX ? NR : BATCH_MAX;
  static_assert(MREP % MREP_C == 0 && MREP_C >= 1 && CROWS % (NW * RPI) == 0 && NR % BATCH == 0, "chunking");
  char* const epi_lds = lds + EPI_OFF;

  DECODE_TILE(t_cur);
  TILE_OFFSETS();
  if (0 < nk) { STAGE_NT(0, 0); }
#ifdef TAD_GEMM_ABLATION
  int stamp_i = 0;
#endif
  bool first_tile = true;
#ifdef TAD_GEMM_ABLATION
// slots 0..15: s_memrealtime (100 MHz) per event; slots 16 + k (k = 0, 1): s_memtime (shader clock) at tile start / K-loop end, so that
// (d memtime / d memrealtime) x 100 MHz is the clock the chip holds INSIDE the K loop (MI355X_MICROARCH.md, DVFS item 6)
#define STAMP(k)                                                                                          \
  if (p.stamps && tid == 0 && stamp_i < 64) {                                                             \
    p.stamps[((size_t)blockIdx.x * 64 + stamp_i) * 32 + (k)] = __builtin_amdgcn_s_memrealtime();          \
    if ((k) < 2) p.stamps[((size_t)blockIdx.x * 64 + stamp_i) * 32 + 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  }
#else
#define STAMP(k) (void)0  // timeline stamps (tad_linear_debug_stamps) exist in ablation builds only
#endif
  for (;;) {
  STAMP(0);
  const int em0 = m0, en0 = n0;  // this tile; (m0, n0) move on to the next one when its first K-tile is prefetched
  const int etile = t_cur;
  const bool qtile = EPI == EPI_PLAIN && p.colscale_cols > 0 && n0 < p.colscale_cols;  // (uniform) see GemmNT::colscale
  bool peeled = false;            // TAD_NT_PEEL: this tile's last K-tile carried its epilogue and the next tile's first prefetch
  // Global accesses of the epilogue's row pass are raw buffer loads / stores: rows >= M fall outside the descriptor (loads return
  // 0, stores are dropped), columns >= N get an out-of-range offset explicitly.  No per-lane branches, and the barriers of the
  // epilogue wait for LDS traffic only (lgkmcnt) -- a __syncthreads() would also drain every store issued so far (vmcnt(0)).
  constexpr uint32_t OOB = 0x80000000u;  // >= any descriptor size accepted by the launcher
  constexpr int ST_AUX = TAD_STORE_AUX;  // cache policy of the output stores
  constexpr int ESZ = OUT_BF16 ? 2 : 4;
  const uint32_t mn_elems = (uint32_t)p.M * (uint32_t)p.N;
  const auto c_rs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(mn_elems * ESZ), 0x00020000);
  const auto pre_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.preact, 0, (int)(mn_elems * 2), 0x00020000);
  const auto h_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dgelu_h, 0, (int)(mn_elems * 2), 0x00020000);
  const auto res_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, (int)((EPI == EPI_RESMOD ? (uint32_t)p.res_mod * (uint32_t)p.N : mn_elems) * 4), 0x00020000);
  const int col = CPL * (lane % LPR);
  const int n = en0 + col;
  const bool nvalid = n < p.N;
  const bool full = (n + CPL <= p.N);             // N % 4 == 0: a bf16 lane has either 8 or 4 valid columns
  const bool n8 = CPL == 4 || (p.N & 7) == 0;     // uniform: every valid lane is a full lane -> one 16-byte access per row
  // What the epilogue READS besides the accumulators (f32 residual rows / bf16 pre-activation rows) is fetched one chunk ahead:
  // chunk 0 during the last K-tile of the main loop, chunk q + 1 while chunk q is processed.  Fetched on demand, each batch of
  // rows exposed a full HBM latency (4 batches x ~3 us per 256 x 128 f32 tile: longer than that tile's K loop at K = 768).
  constexpr bool HAS_EXTRA = !SPLITK && (IS_RES || EPI == EPI_DGELU);  // (the split-K variant reads them in its combine pass)
  constexpr int EXW = (IS_RES) ? CPL / 4 : 1;
  constexpr int NJ = OUT_BF16 ? NREP / 2 : NREP;  // DIRECT: 16-byte column groups per lane and m-fragment
  u32x4 extra[2][HAS_EXTRA ? (DIRECT ? NJ : NR) : 1][EXW];
  // DIRECT addressing: acc[i][j] of lane (c, kq) = out[em0 + wm*WTM + 16i + c][en0 + wn*WTN + cg(j) .. +3], cg as in the LDS path
#define DIRECT_COL(jj) (en0 + wn * WTN + (OUT_BF16 ? 32 * (jj) + 8 * kq : 16 * (jj) + 4 * kq))
#define ISSUE_EXTRA_D(i, buf)                                                                                           \
  if (HAS_EXTRA && (!IS_RES || p.residual)) {                                                               \
    const int m = em0 + wm * WTM + 16 * (i) + c;                                                                        \
    _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                                                 \
      const int nn = DIRECT_COL(jj);                                                                                    \
      const bool fulld = nn + CPL <= p.N;                                                                               \
      uint32_t o = nn < p.N ? (uint32_t)m * (uint32_t)p.N + (uint32_t)nn : OOB;                                         \
      if (IS_RES) {                                                                                        \
        if (EPI == EPI_RESMOD) o = (nn < p.N && m < p.M) ? (uint32_t)((m + p.row_base) % p.res_mod) * (uint32_t)p.N + (uint32_t)nn : OOB;  \
        const uint32_t rb = o == OOB ? OOB : o * 4;                                                                     \
        extra[buf][jj][0] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, rb, 0, 0);                                    \
        if (CPL == 8) extra[buf][jj][EXW - 1] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, fulld ? rb + 16 : OOB, 0, 0); \
      } else {                                                                                                          \
        const uint32_t hb = o == OOB ? OOB : o * 2;                                                                     \
        if (CPL == 8) {                                                                                                 \
          if (n8) extra[buf][jj][0] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, hb, 0, 0);                            \
          else {                                                                                                        \
            const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(h_rs, hb, 0, 0);                                      \
            const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(h_rs, fulld ? hb + 8 : OOB, 0, 0);                    \
            extra[buf][jj][0] = u32x4{lo[0], lo[1], hi[0], hi[1]};                                                      \
          }                                                                                                             \
        } else {                                                                                                        \
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(h_rs, hb, 0, 0);                                        \
          extra[buf][jj][0] = u32x4{lo[0], lo[1], 0u, 0u};                                                              \
        }                                                                                                               \
      }                                                                                                                 \
    }                                                                                                                   \
  }
#define ISSUE_EXTRA(q, buf)                                                                                             \
  if (HAS_EXTRA && (!IS_RES || p.residual)) {                                                               \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                                                    \
      const int lr = (r * NW + wave) * RPI + lane / LPR;                                                                \
      const int m = em0 + (lr / (16 * MREP_C)) * WTM + 16 * MREP_C * (q) + lr % (16 * MREP_C);                          \
      uint32_t o = nvalid ? (uint32_t)m * (uint32_t)p.N + (uint32_t)n : OOB;                                            \
      if (IS_RES) {                                                                                        \
        if (EPI == EPI_RESMOD) o = (nvalid && m < p.M) ? (uint32_t)((m + p.row_base) % p.res_mod) * (uint32_t)p.N + (uint32_t)n : OOB;     \
        const uint32_t rb = o == OOB ? OOB : o * 4;                                                                     \
        extra[buf][r][0] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, rb, 0, 0);                                     \
        if (CPL == 8) extra[buf][r][EXW - 1] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, full ? rb + 16 : OOB, 0, 0); \
      } else {                                                                                                          \
        const uint32_t hb = o == OOB ? OOB : o * 2;                                                                     \
        if (CPL == 8) {                                                                                                 \
          if (n8) extra[buf][r][0] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, hb, 0, 0);                             \
          else {                                                                                                        \
            const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(h_rs, hb, 0, 0);                                      \
            const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(h_rs, full ? hb + 8 : OOB, 0, 0);                     \
            extra[buf][r][0] = u32x4{lo[0], lo[1], hi[0], hi[1]};                                                       \
          }                                                                                                             \
        } else {                                                                                                        \
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(h_rs, hb, 0, 0);                                        \
          extra[buf][r][0] = u32x4{lo[0], lo[1], 0u, 0u};                                                               \
        }                                                                                                               \
      }                                                                                                                 \
    }                                                                                                                   \
  }
  // accumulators start from the bias (a per-column constant = per (j, kq, r) constant in this layout): no bias add later
  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int nc = n0 + wn * WTN + (OUT_BF16 ? (32 * (j >> 1) + 8 * kq + 4 * (j & 1)) : (16 * j + 4 * kq));
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI != EPI_DGELU && p.bias && nc < p.N && (!SPLITK || sk_share == 0)) {
      if (EPI == EPI_PLAIN && p.bias_seg > 0) {
        if (nc < p.bias_seg || nc >= 2 * p.bias_seg) {
          const float4 t = *reinterpret_cast<const float4*>(nc < p.bias_seg ? p.bias + nc : p.bias2 + (nc - 2 * p.bias_seg));
          b4 = f32x4{t.x, t.y, t.z, t.w};
        }
      } else {
        const float4 t = *reinterpret_cast<const float4*>(p.bias + nc);
        b4 = f32x4{t.x, t.y, t.z, t.w};
      }
    }
#pragma unroll
    for (int i = 0; i < MREP; ++i) acc[i][j] = b4;
  }
  if constexpr (W4) {
    // ---- W4 K loop.  Per K-tile and wave: 128 MFMAs (16 x 16 x 32) in two halves of 64 -- k-step 0 and k-step 1, each as 8 groups of
    // 8 (one x fragment against the 8 w fragments) -- with two fragment sets: while the matrix pipe works through one k-step the 16
    // fragments of the next one arrive (two 16-byte reads behind every group), so the only place the wave waits for the LDS is a counted
    // wait at the half boundary, by which time the data has had 64 MFMAs to land.  K-tile kt + 2 is requested (two LDS-DMA pieces per
    // group) into the ring slot of K-tile kt during kt's SECOND half, behind the one barrier of the tile: every wave has its k-step-1
    // fragments of that slot in registers by then, and the k-step-0 fragments were taken during the previous tile.  So a piece has a
    // whole tile (about 1.1 us) to arrive, one barrier per K-tile orders everything, and nothing but the half-boundary waits separates
    // two matrix instructions.  Every LDS read is inline asm with an immediate offset (common.h): addresses are 32 lane constants
    // computed once per kernel ([slot][k-step][fragment & 3]; fragment q + 4 lies 8 KiB behind fragment q with the same swizzle).
    // (the ring slot is toggled in the address REGISTERS, one v_xor each per K-tile, and in a scalar for the DMA destination: two copies
    //  of the body selected by a branch make the compiler route the 256 accumulators and the fragments through PHI copies and spill)
    static_assert(STAGE_BYTES == 65536, "slot toggle = bit 16 of the LDS address");
    uint32_t a_ad[2][4], b_ad[2][4];  // [k-step][fragment & 3], pointing into the slot that is being read
    {
      const uint32_t l0 = lds_addr(lds);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a_ad[ks][q] = l0 + a_rd[q] + (uint32_t)((((4 * ks) + kq) ^ a_sw[q]) << 4);
          b_ad[ks][q] = l0 + (uint32_t)A_BYTES + b_rd[q] + (uint32_t)((((4 * ks) + kq) ^ b_sw[q]) << 4);
        }
    }
    static_assert(MREP == 8 && NREP == 8 && LOADS == 16, "W4 geometry");
    op16x8 fa[2][8], fb[2][8];  // [fragment set = k-step][fragment]
    // one LDS-DMA piece of K-tile kt into the ring slot at byte offset SLOT: pieces 0..7 are x rows, 8..15 w rows (stage_tile's layout).
#define W4_PIECE(SLOT, idx, kt_)                                                                                                          \
  if ((idx) < 8) w4_dma_piece(p.A, a_bytes, lds + (SLOT) + (((idx) & 7) * NW + wave) * 1024, a_off[(idx) & 7], (uint32_t)(kt_) * ROWB);    \
  else w4_dma_piece(p.B, b_bytes, lds + (SLOT) + A_BYTES + (((idx) & 7) * NW + wave) * 1024, b_off[(idx) & 7], (uint32_t)(kt_) * ROWB)
    // K-tile 1 (K-tile 0 is on its way: issued before the loop, or under the previous tile's epilogue); nk >= 2 (the launcher's rule)
    static_for<0, 16>([&](auto ic) { W4_PIECE(STAGE_BYTES, decltype(ic)::value, 1); });
    first_tile = false;
    // (W4PF) the touch is always the YOUNGEST vector-memory operation behind a K-tile's pieces, so every wait for those pieces leaves exactly
    // one operation in flight (vmcnt(1) / + 1), and a touch has two K-tiles to land before it would hold up the pieces requested behind it
#define W4_TOUCH(kt_)                                                                                                                     \
    if constexpr (W4PF) {                                                                                                                 \
      const int ktp = min((int)(kt_), nk - 1);                                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A), 0, a_bytes, 0x00020000),    \
                                               LDS_PTR(lds + PF_SINK + wave * 256), 4, pf_off, (uint32_t)(ktp + kt0) * ROWB, 0, 0);       \
    }
    W4_TOUCH(1 + TAD_W4_PREFETCH);
    wait_stage<LOADS, W4PF ? 1 : 0>(1);
    block_barrier();
    static_for<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      fb[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[0][g & 3]);
      fa[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[0][g & 3]);
    });
    int slot = 0;  // byte offset of the ring slot of K-tile kt
    // NEXT / NEXT2 (compile-time): there is a K-tile kt + 1 / kt + 2.  The steady state is ONE copy of the body inside the loop; the last
    // two K-tiles follow it as straight-line code.
    auto w4_tile = [&](auto NEXTC, auto NEXT2C, int kt) {
      constexpr bool NEXT = decltype(NEXTC)::value, NEXT2 = decltype(NEXT2C)::value;
      // ---- first half: k-step 0 products; the k-step-1 fragments of this slot are requested behind each group
      lds_wait<0>(fb[0][0], fb[0][1], fb[0][2], fb[0][3], fb[0][4], fb[0][5], fb[0][6], fb[0][7]);
      lds_wait<0>(fa[0][0], fa[0][1], fa[0][2], fa[0][3], fa[0][4], fa[0][5], fa[0][6], fa[0][7]);
      static_for<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        fb[1][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[1][g & 3]);
        fa[1][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[1][g & 3]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = TAD_MFMA_16x16x32(fb[0][j], fa[0][g], acc[g][j]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // ---- second half: k-step 1 products; K-tile kt + 1 has landed -> its k-step-0 fragments; K-tile kt + 2 requested into this slot
      lds_wait<0>(fb[1][0], fb[1][1], fb[1][2], fb[1][3], fb[1][4], fb[1][5], fb[1][6], fb[1][7]);
      lds_wait<0>(fa[1][0], fa[1][1], fa[1][2], fa[1][3], fa[1][4], fa[1][5], fa[1][6], fa[1][7]);
      if constexpr (NEXT) {
        if constexpr (W4PF) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        block_barrier();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int q = 0; q < 4; ++q) { a_ad[ks][q] ^= (uint32_t)STAGE_BYTES; b_ad[ks][q] ^= (uint32_t)STAGE_BYTES; }
      }
      static_for<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (NEXT2) {
          W4_PIECE(slot, 2 * g, kt + 2);
          W4_PIECE(slot, 2 * g + 1, kt + 2);
        }
        if constexpr (NEXT) {
          fb[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(b_ad[0][g & 3]);
          fa[0][g] = lds_read_b128<op16x8, (g >> 2) * 8192>(a_ad[0][g & 3]);
        }
        if constexpr (NEXT2 && g == 7) { W4_TOUCH(kt + 2 + TAD_W4_PREFETCH); }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] = TAD_MFMA_16x16x32(fb[1][j], fa[1][g], acc[g][j]);
        __builtin_amdgcn_sched_barrier(0);
      });
      slot ^= STAGE_BYTES;
    };
    int kt = 0;
    for (; kt + 2 < nk; ++kt) w4_tile(std::true_type{}, std::true_type{}, kt);
    w4_tile(std::true_type{}, std::false_type{}, kt);
    w4_tile(std::false_type{}, std::false_type{}, kt + 1);
    // what the epilogue's first chunk reads besides the accumulators: requested BEHIND the K loop here, not inside its last K-tile as in the
    // eight-wave kernels -- beside two live fragment sets the prefetch registers made the residual / DGELU instantiations spill
    if constexpr (HAS_EXTRA) { ISSUE_EXTRA(0, 0); }
    // (persistent: the address registers must point at slot 0 again for the next tile, whose K-tile 0 lands there)
    if ((nk & 1) == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q) { a_ad[ks][q] ^= (uint32_t)STAGE_BYTES; b_ad[ks][q] ^= (uint32_t)STAGE_BYTES; }
    }
#undef W4_PIECE
#undef W4_TOUCH
  } else {
    // K-tile 0 is already on its way (issued before the loop, or under the previous tile's epilogue); with the DIRECT epilogue
    // (which leaves the LDS alone) so are the other prologue stages of every tile but the first
    if (!DIRECT || first_tile) {
#pragma unroll
      for (int st = 1; st < STAGES - 1; ++st)
        if (st < nk) { STAGE_NT(st, st); }
    }
    first_tile = false;
    int rd = 0, wr = STAGES - 1;
    // PEEL (TAD_NT_PEEL, on by default; persistent 256 x 256 kernel with the register-layout bias-only bf16 epilogue, even number of K-tiles): the last
    // K-tile is taken out of the loop and run row fragment by row fragment -- both k-steps of row i back to back, then the conversion and the
    // two 16-byte stores of row i - 1 while row i's eight MFMAs execute -- so the store epilogue runs beside the last 1/nk of the matrix work
    // instead of behind it; the next tile's first K-tile is requested at the TOP of this K-tile (ring slot 0 was last read one K-tile ago and
    // the barrier that opens this K-tile proves it), and the barrier behind the K loop disappears (the next tile's first barrier orders its DMA
    // into slot 1 behind this tile's last reads).  Measured against the unpeeled build (bit-identical results): qkv forward 202 -> 192 us, the four
    // bias-only shapes of a block 677 -> 660 us, the training step 684.4 -> 686.7 clips/s over three alternating pairs.
    constexpr bool PEEL_OK = TAD_NT_PEEL && PERSIST && DIRECT && EPI == EPI_PLAIN && OUT_BF16 && BM == 256 && BN == 256 && KSTEPS == 2 &&
                             STAGES == 2;
    const bool peel = PEEL_OK && nk >= 2 && (nk & 1) == 0 && !(DBG_BITS(p) & 7);
    peeled = peel;
    const int nk_loop = peel ? nk - 1 : nk;
    for (int kt = 0; kt < nk_loop; ++kt) {
      // tile kt has landed once all but the younger stages' DMAs of this wave are done; the barrier then (a) publishes every
      // wave's part of tile kt and (b) proves all waves finished reading tile kt-1, whose buffer the next DMA overwrites
      wait_stage<LOADS>(min(STAGES - 2, nk - 1 - kt));
      block_barrier();
      if (HAS_EXTRA && kt == nk - 1) {
        if (DIRECT) { ISSUE_EXTRA_D(0, 0); } else { ISSUE_EXTRA(0, 0); }
      }
      const char* sa = lds + rd * STAGE_BYTES;
      const char* sb = sa + A_BYTES;
      const bool more = kt + STAGES - 1 < nk;
      const int wr_now = wr, kt_next = kt + STAGES - 1;
      rd = (rd + 1 == STAGES) ? 0 : rd + 1;
      wr = (wr + 1 == STAGES) ? 0 : wr + 1;
      // Issuing a tile's LDS-DMA pieces blocks the issuing wave for ~100 cycles per piece.  The two waves that share a SIMD
      // (wave w and w + NW/2) therefore issue them at different times: the older half before its first k-step, the younger
      // half between its two k-steps, so the SIMD's matrix pipe always has one wave feeding it.
      const bool dma = more && !(DBG_BITS(p) & 1);
      op16x8 af[MREP], bfr[NREP];
      if (dma && !late) { STAGE_NT(wr_now, kt_next); }
      FRAG_B(bfr, sb, 0);
      FRAG_A(af, sa, 0);
      MFMA_BLOCK(af, bfr);
      if (dma && late) { STAGE_NT(wr_now, kt_next); }
      FRAG_B(bfr, sb, 1);
      FRAG_A(af, sa, 1);
      MFMA_BLOCK(af, bfr);
    }
    if constexpr (PEEL_OK) {
      if (peel) {
        wait_stage<LOADS>(0);
        block_barrier();
        const char* sa = lds + rd * STAGE_BYTES;  // rd == 1 (nk even): slot 0 is free
        const char* sb = sa + A_BYTES;
        if (t_cur + t_step < t_end) {  // the next tile's first K-tile, under this one's MFMAs
          DECODE_TILE(t_cur + t_step);
          TILE_OFFSETS();
          STAGE_NT(0, 0);
        }
        op16x8 b0[NREP], b1[NREP];
        FRAG_B(b0, sb, 0);
        FRAG_B(b1, sb, 1);
        const bool n8p = (p.N & 7) == 0;
        auto store_row = [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const int m = em0 + wm * WTM + 16 * i + c;
#pragma unroll
          for (int jj = 0; jj < NREP / 2; ++jj) {
            const int nn = en0 + wn * WTN + 32 * jj + 8 * kq;
            const bool fulld = nn + 8 <= p.N;
            const uint32_t o = nn < p.N ? (uint32_t)m * (uint32_t)p.N + (uint32_t)nn : OOB;
            const uint32_t ob = o == OOB ? OOB : o * 2;
            if (qtile) {  // (uniform) a tile with columns of the pre-scaled range: tad_linear_fwd_qkv's q_prescale
              const float cs = nn < p.colscale_cols ? p.colscale : 1.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) { acc[i][2 * jj][e] *= cs; acc[i][2 * jj + 1][e] *= cs; }
            }
            const u32x2 lo = u32x2{pack_op16x2(acc[i][2 * jj][0], acc[i][2 * jj][1]), pack_op16x2(acc[i][2 * jj][2], acc[i][2 * jj][3])};
            const u32x2 hi = u32x2{pack_op16x2(acc[i][2 * jj + 1][0], acc[i][2 * jj + 1][1]), pack_op16x2(acc[i][2 * jj + 1][2], acc[i][2 * jj + 1][3])};
            if (n8p) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, c_rs, ob, 0, ST_AUX);
            else {
              __builtin_amdgcn_raw_buffer_store_b64(lo, c_rs, ob, 0, ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64(hi, c_rs, fulld ? ob + 8 : OOB, 0, ST_AUX);
            }
          }
        };
        static_for<0, MREP>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const op16x8 a0 = *reinterpret_cast<const op16x8*>(sa + a_rd[i] + (((0 + kq) ^ a_sw[i]) << 4));
          const op16x8 a1 = *reinterpret_cast<const op16x8*>(sa + a_rd[i] + (((4 + kq) ^ a_sw[i]) << 4));
#pragma unroll
          for (int j = 0; j < NREP; ++j) acc[i][j] = TAD_MFMA_16x16x32(b0[j], a0, acc[i][j]);
#pragma unroll
          for (int j = 0; j < NREP; ++j) acc[i][j] = TAD_MFMA_16x16x32(b1[j], a1, acc[i][j]);
          if constexpr (i >= 1) {
            store_row(std::integral_constant<int, i - 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        store_row(std::integral_constant<int, MREP - 1>{});
      }
    }
  }

  // ---- epilogue.  Straight from the MFMA layout a global access touches 16 rows x 64 bytes per instruction (24.6 B/clk per CU
  // against 70 for whole rows: tools/micro/store_rate.hip), so except for the bias-only bf16 case (DIRECT) the accumulators are
  // transposed through the LDS in chunks of CROWS rows and every global load / store of a wave covers whole contiguous rows
  // (512 B - 1 KiB runs).  The variant (EPI, OUT_BF16) is a template parameter, offsets are 32-bit buffer offsets, the rows read
  // besides the accumulators come one chunk ahead (ISSUE_EXTRA), and the LDS reads of a chunk are batched BATCH rows at a time.
  t_cur += t_step;
  if (!peeled) block_barrier();  // every wave is done with the ring
  const bool has_next = PERSIST && t_cur < t_end;
  STAMP(1);
  if (has_next && !peeled) {
    DECODE_TILE(t_cur);
    TILE_OFFSETS();
    if (0 < nk) { STAGE_NT(0, 0); }
    if (DIRECT) {
#pragma unroll
      for (int st = 1; st < STAGES - 1; ++st)
        if (st < nk) { STAGE_NT(st, st); }
    }
  }
  // Per-row scale (drop-path keep / scale of the row's clip): a 256-row tile lies in at most two groups of rows_per_scale rows
  // (1568 tokens per clip), so the tile takes its one or two scales through scalar loads here.  A per-row global load inside the row
  // pass costs a `s_waitcnt vmcnt(0)` per row -- which on CDNA4 also drains every store and the residual rows fetched ahead.
  // The 256-row tiles are only launched with groups of at least 256 rows (launch_nt_variant) and carry no other path; the 128-row tile
  // keeps the per-row load for shorter groups (tiny problems).
  constexpr bool RS_ALWAYS_TILE = BM >= 256;
  float rs_lo = 1.f, rs_hi = 1.f;
  int rs_split = 0x7fffffff;
  const bool rs_tile = IS_RES && p.rowscale && (RS_ALWAYS_TILE || p.rows_per_scale >= BM);
  if (rs_tile) {
    const int g0 = (em0 + p.row_base) / p.rows_per_scale;
    const int last = (em0 + BM - 1 < p.M ? em0 + BM - 1 : p.M - 1) + p.row_base;
    rs_lo = p.rowscale[g0];
    rs_hi = p.rowscale[last / p.rows_per_scale];
    rs_split = (g0 + 1) * p.rows_per_scale - p.row_base;  // first row (of this launch) in the second group
  }
  if (DIRECT && !peeled && !((DBG_BITS(p) & 4) && p.M > 1)) {
    float gam[CPL];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      if (i + 1 < MREP) { ISSUE_EXTRA_D(i + 1, (i + 1) & 1); }
      const int m = em0 + wm * WTM + 16 * i + c;
      float rsc = 1.f;
      if (RS_ALWAYS_TILE || rs_tile) rsc = m < rs_split ? rs_lo : rs_hi;
      else if (IS_RES && p.rowscale && m < p.M) rsc = p.rowscale[(m + p.row_base) / p.rows_per_scale];
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int nn = DIRECT_COL(jj);
        const bool fulld = nn + CPL <= p.N;
        const uint32_t o = nn < p.N ? (uint32_t)m * (uint32_t)p.N + (uint32_t)nn : OOB;
        const uint32_t ob = o == OOB ? OOB : o * ESZ;
        float v[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) v[e] = OUT_BF16 ? acc[i][2 * jj + (e >> 2)][e & 3] : acc[i][jj][e & 3];
        if (qtile) {
          const float cs = nn < p.colscale_cols ? p.colscale : 1.f;
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] *= cs;
        }
        if (EPI == EPI_GELU) {
          if (p.preact) {
            const uint32_t pb = o == OOB ? OOB : o * 2;
            const u32x2 lo = u32x2{pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3])};
            if (CPL == 8) {
              const u32x2 hi = u32x2{pack_op16x2(v[CPL - 4], v[CPL - 3]), pack_op16x2(v[CPL - 2], v[CPL - 1])};
              if (n8) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, pre_rs, pb, 0, ST_AUX);
              else {
                __builtin_amdgcn_raw_buffer_store_b64(lo, pre_rs, pb, 0, ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(hi, pre_rs, fulld ? pb + 8 : OOB, 0, ST_AUX);
              }
            } else {
              __builtin_amdgcn_raw_buffer_store_b64(lo, pre_rs, pb, 0, ST_AUX);
            }
          }
          gelu_fast_row<CPL / 2>(v);
        } else if (EPI == EPI_DGELU) {
          const u32x4 hh = extra[i & 1][jj][0];
          const uint32_t hw[4] = {hh[0], hh[1], hh[2], hh[3]};
          gelu_grad_fast_row<CPL / 2>(v, hw);
        } else if (IS_RES) {
          if (p.gamma || p.rowscale) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) gam[e] = (p.gamma && nn < p.N && (e < 4 || fulld)) ? p.gamma[nn + e] : 1.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] *= gam[e] * rsc;
          }
          if (p.residual) {
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
              const u32x4 rr = extra[i & 1][jj][e4 < EXW ? e4 : 0];
              v[4 * e4 + 0] += __uint_as_float(rr[0]); v[4 * e4 + 1] += __uint_as_float(rr[1]);
              v[4 * e4 + 2] += __uint_as_float(rr[2]); v[4 * e4 + 3] += __uint_as_float(rr[3]);
            }
          }
        }
        if (OUT_BF16) {
          const u32x2 lo = u32x2{pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3])};
          const u32x2 hi = u32x2{pack_op16x2(v[CPL - 4], v[CPL - 3]), pack_op16x2(v[CPL - 2], v[CPL - 1])};
          if (n8) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, c_rs, ob, 0, ST_AUX);
          else {
            __builtin_amdgcn_raw_buffer_store_b64(lo, c_rs, ob, 0, ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(hi, c_rs, fulld ? ob + 8 : OOB, 0, ST_AUX);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, c_rs, ob, 0, ST_AUX);
        }
      }
    }
  }
  if constexpr (SPLITK) {
    // ---- split-K: (1) the partial tile goes to the workspace as whole rows, through the LDS transposition of the ordinary epilogue
    const int S = p.sk_splits;
    float* const part = p.sk_ws + ((size_t)etile * S + sk_share) * (size_t)(BM * BN);
    const auto part_rs = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, BM * BN * 4, 0x00020000);
    if (p.sk_mode != 2) {
#pragma unroll
    for (int q = 0; q < NCHUNK; ++q) {
#pragma unroll
      for (int ii = 0; ii < MREP_C; ++ii) {
        const int lr = wm * (16 * MREP_C) + ii * 16 + c;
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          const int cc = wn * WTN + (OUT_BF16 ? (32 * (j >> 1) + 8 * kq + 4 * (j & 1)) : (16 * j + 4 * kq));
          *reinterpret_cast<f32x4*>(epi_lds + lr * CSTRIDE + cc * 4) = acc[q * MREP_C + ii][j];
        }
      }
      lds_barrier();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int lr = (r * NW + wave) * RPI + lane / LPR;
        const int trow = (lr / (16 * MREP_C)) * WTM + 16 * MREP_C * q + lr % (16 * MREP_C);
#pragma unroll
        for (int e4 = 0; e4 < CPL / 4; ++e4) {
          const u32x4 t = *reinterpret_cast<const u32x4*>(epi_lds + lr * CSTRIDE + col * 4 + 16 * e4);
          // write-through (sc1): the partial tile leaves the XCD's L2 as it is stored, so publishing it needs no agent-scope release
          // (a release fence writes back EVERY dirty line of the L2: 8 us and more with 256 KB freshly written per workgroup).
          // The deferred plan (sk_mode 1) is ordered by the kernel boundary and stores with the ordinary output policy.
          if (p.sk_mode == 0) __builtin_amdgcn_raw_buffer_store_b128(t, part_rs, (uint32_t)((trow * BN + col + 4 * e4) * 4), 0, 16);
          else __builtin_amdgcn_raw_buffer_store_b128(t, part_rs, (uint32_t)((trow * BN + col + 4 * e4) * 4), 0, ST_AUX);
        }
      }
      if (q + 1 < NCHUNK) lds_barrier();
    }
    }
    if (p.sk_mode == 1) break;  // (deferred plan: a later launch combines; the kernel boundary orders the two)
    // ---- (2) publish, wait for the other shares of this tile (Guideline 16, form R1: write-through stores, every storing wave
    // drains them, barrier, ONE lane adds to the counter; ONE relaxed poll, one agent-scope acquire, barrier, then plain loads)
    if (p.sk_mode == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(p.sk_cnt + etile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(p.sk_cnt + etile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22)) {  // (seconds: a share of this tile is not running -- the grid was not resident) give up loudly
          __hip_atomic_store(p.sk_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (pinned host memory: sk_error_word)
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    }
    // ---- (3) this workgroup's rows of the tile: sum over the shares (share 0 carries the bias), epilogue, store
    float gam[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) gam[e] = (IS_RES && p.gamma && nvalid && (e < 4 || full)) ? p.gamma[n + e] : 1.f;
    const int r0 = sk_share * BM / S, r1 = (sk_share + 1) * BM / S;
    const float* const tile_ws = p.sk_ws + (size_t)etile * S * (size_t)(BM * BN);
    const auto ws_rs = __builtin_amdgcn_make_buffer_rsrc((void*)tile_ws, 0, S * BM * BN * 4, 0x00020000);
    // rows r0 + (it * NW + wave) * RPI + lane / LPR; RB row-instructions per batch so that ~16 partial-tile loads are in flight per
    // lane (one dependent round trip per batch instead of one per row); SS = number of shares as a literal
    auto combine = [&](auto SSC) {
      constexpr int SS = decltype(SSC)::value;
      constexpr int RB = (16 / (SS * (CPL / 4))) > 0 ? (16 / (SS * (CPL / 4))) : 1;
      for (int row0 = r0 + wave * RPI + lane / LPR; row0 - (wave * RPI + lane / LPR) < r1; row0 += RB * NW * RPI) {
        u32x4 pv[RB][SS][CPL / 4], rr[RB][CPL / 4];
        uint32_t off[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const int row = row0 + b * NW * RPI;
          const bool live = row < r1;
          const int m = em0 + row;
          off[b] = (live && nvalid) ? (uint32_t)m * (uint32_t)p.N + (uint32_t)n : OOB;
#pragma unroll
          for (int sh = 0; sh < SS; ++sh)
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4)
              pv[b][sh][e4] = __builtin_amdgcn_raw_buffer_load_b128(ws_rs, live ? (uint32_t)(((sh * BM + row) * BN + col + 4 * e4) * 4) : OOB, 0, 0);
          if (IS_RES && p.residual) {
            const uint32_t rb = off[b] == OOB ? OOB : off[b] * 4;
            rr[b][0] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, rb, 0, 0);
            if (CPL == 8) rr[b][CPL / 4 - 1] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, full ? rb + 16 : OOB, 0, 0);
          }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const int m = em0 + row0 + b * NW * RPI;
          float v[CPL];
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[e] = __uint_as_float(pv[b][0][e >> 2][e & 3]);
#pragma unroll
          for (int sh = 1; sh < SS; ++sh)  // fixed order: share 0 (it carries the bias), 1, 2, ...
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] += __uint_as_float(pv[b][sh][e >> 2][e & 3]);
          if (IS_RES) {
            if (p.gamma || p.rowscale) {
              float rsc;
              if (RS_ALWAYS_TILE || rs_tile) rsc = m < rs_split ? rs_lo : rs_hi;
              else rsc = (p.rowscale && m < p.M) ? p.rowscale[(m + p.row_base) / p.rows_per_scale] : 1.f;
#pragma unroll
              for (int e = 0; e < CPL; ++e) v[e] *= gam[e] * rsc;
            }
            if (p.residual) {
#pragma unroll
              for (int e = 0; e < CPL; ++e) v[e] += __uint_as_float(rr[b][e >> 2][e & 3]);
            }
          }
          const uint32_t ob = off[b] == OOB ? OOB : off[b] * ESZ;
          if (OUT_BF16) {
            const u32x2 lo = u32x2{pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3])};
            const u32x2 hi = u32x2{pack_op16x2(v[CPL - 4], v[CPL - 3]), pack_op16x2(v[CPL - 2], v[CPL - 1])};
            if (n8) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, c_rs, ob, 0, ST_AUX);
            else {
              __builtin_amdgcn_raw_buffer_store_b64(lo, c_rs, ob, 0, ST_AUX);
              __builtin_amdgcn_raw_buffer_store_b64(hi, c_rs, full ? ob + 8 : OOB, 0, ST_AUX);
            }
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, c_rs, ob, 0, ST_AUX);
          }
        }
      }
    };
    switch (S) {
      case 2: combine(std::integral_constant<int, 2>{}); break;
      case 3: combine(std::integral_constant<int, 3>{}); break;
      case 4: combine(std::integral_constant<int, 4>{}); break;
      case 5: combine(std::integral_constant<int, 5>{}); break;
      case 6: combine(std::integral_constant<int, 6>{}); break;
      case 7: combine(std::integral_constant<int, 7>{}); break;
      default: combine(std::integral_constant<int, 8>{}); break;
    }
  }
  if (!SPLITK && !DIRECT && !((DBG_BITS(p) & 4) && p.M > 1)) {
  float gam[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) gam[e] = (IS_RES && p.gamma && nvalid && (e < 4 || full)) ? p.gamma[n + e] : 1.f;
  // (1) every wave drops its MREP_C x NREP fragments of chunk qq into chunk buffer qq & 1 (EPI_DB) / the one buffer
#define DROP_CHUNK(qq)                                                                                                       \
  _Pragma("unroll") for (int ii = 0; ii < MREP_C; ++ii) {                                                                    \
    const int lr = wm * (16 * MREP_C) + ii * 16 + c;                                                                         \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j) {                                                                       \
      const int cc = wn * WTN + (OUT_BF16 ? (32 * (j >> 1) + 8 * kq + 4 * (j & 1)) : (16 * j + 4 * kq));                     \
      *reinterpret_cast<f32x4*>(epi_lds + (EPI_DB ? ((qq) & 1) * CROWS * CSTRIDE : 0) + lr * CSTRIDE + cc * 4) = acc[(qq) * MREP_C + ii][j]; \
    }                                                                                                                        \
  }
  if (EPI_DB) {
    DROP_CHUNK(0);
    lds_barrier();
  }
#pragma unroll
  for (int q = 0; q < NCHUNK; ++q) {
    if (!EPI_DB) {
      DROP_CHUNK(q);
      lds_barrier();
    }
    if (q < 6) { STAMP(4 + 2 * q); }
    if (q + 1 < NCHUNK) { ISSUE_EXTRA(q + 1, (q + 1) & 1); }
    if (EPI_DB && q + 1 < NCHUNK) { DROP_CHUNK(q + 1); }  // (its barrier is the one that ends this chunk)
    const char* const epi_rd = epi_lds + (EPI_DB ? (q & 1) * CROWS * CSTRIDE : 0);
    // (2) row-contiguous pass: local row lr <-> tile row (lr / (16*MREP_C))*WTM + 16*MREP_C*q + lr % (16*MREP_C)
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += BATCH) {
      float v[BATCH][CPL];
      uint32_t off[BATCH];  // element offset m*N + n, or OOB
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const int lr = ((r0 + b) * NW + wave) * RPI + lane / LPR;
        const int m = em0 + (lr / (16 * MREP_C)) * WTM + 16 * MREP_C * q + lr % (16 * MREP_C);
        off[b] = nvalid ? (uint32_t)m * (uint32_t)p.N + (uint32_t)n : OOB;
#pragma unroll
        for (int e4 = 0; e4 < CPL / 4; ++e4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(epi_rd + lr * CSTRIDE + col * 4 + 16 * e4);
          v[b][4 * e4 + 0] = t[0]; v[b][4 * e4 + 1] = t[1]; v[b][4 * e4 + 2] = t[2]; v[b][4 * e4 + 3] = t[3];
        }
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        const uint32_t ob = off[b] == OOB ? OOB : off[b] * ESZ;  // byte offset into C
        if (qtile) {
          const float cs = n < p.colscale_cols ? p.colscale : 1.f;
#pragma unroll
          for (int e = 0; e < CPL; ++e) v[b][e] *= cs;
        }
        if (EPI == EPI_GELU) {
          if (p.preact) {
            const uint32_t pb = off[b] == OOB ? OOB : off[b] * 2;
            const u32x2 lo = u32x2{pack_op16x2(v[b][0], v[b][1]), pack_op16x2(v[b][2], v[b][3])};
            if (CPL == 8) {
              const u32x2 hi = u32x2{pack_op16x2(v[b][CPL - 4], v[b][CPL - 3]), pack_op16x2(v[b][CPL - 2], v[b][CPL - 1])};
              if (n8) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, pre_rs, pb, 0, ST_AUX);
              else {
                __builtin_amdgcn_raw_buffer_store_b64(lo, pre_rs, pb, 0, ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(hi, pre_rs, full ? pb + 8 : OOB, 0, ST_AUX);
              }
            } else {
              __builtin_amdgcn_raw_buffer_store_b64(lo, pre_rs, pb, 0, ST_AUX);
            }
          }
          gelu_fast_row<CPL / 2>(v[b]);
        } else if (EPI == EPI_DGELU) {
          const u32x4 hh = extra[q & 1][r0 + b][0];
          const uint32_t hw[4] = {hh[0], hh[1], hh[2], hh[3]};
          gelu_grad_fast_row<CPL / 2>(v[b], hw);
        } else if (IS_RES) {
          if (p.gamma || p.rowscale) {
            const int lr = ((r0 + b) * NW + wave) * RPI + lane / LPR;
            const int m = em0 + (lr / (16 * MREP_C)) * WTM + 16 * MREP_C * q + lr % (16 * MREP_C);
            float rsc;
            if (RS_ALWAYS_TILE || rs_tile) rsc = m < rs_split ? rs_lo : rs_hi;
            else rsc = (p.rowscale && m < p.M) ? p.rowscale[(m + p.row_base) / p.rows_per_scale] : 1.f;
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[b][e] *= gam[e] * rsc;
          }
          if (p.residual) {
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
              const u32x4 rr = extra[q & 1][r0 + b][e4 < EXW ? e4 : 0];
              v[b][4 * e4 + 0] += __uint_as_float(rr[0]); v[b][4 * e4 + 1] += __uint_as_float(rr[1]);
              v[b][4 * e4 + 2] += __uint_as_float(rr[2]); v[b][4 * e4 + 3] += __uint_as_float(rr[3]);
            }
          }
        }
        if (OUT_BF16) {
          const u32x2 lo = u32x2{pack_op16x2(v[b][0], v[b][1]), pack_op16x2(v[b][2], v[b][3])};
          const u32x2 hi = u32x2{pack_op16x2(v[b][CPL - 4], v[b][CPL - 3]), pack_op16x2(v[b][CPL - 2], v[b][CPL - 1])};
          if (n8) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo[0], lo[1], hi[0], hi[1]}, c_rs, ob, 0, ST_AUX);
          else {
            __builtin_amdgcn_raw_buffer_store_b64(lo, c_rs, ob, 0, ST_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(hi, c_rs, full ? ob + 8 : OOB, 0, ST_AUX);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[b][0]), __float_as_uint(v[b][1]), __float_as_uint(v[b][2]), __float_as_uint(v[b][3])},
                                                 c_rs, ob, 0, ST_AUX);
        }
      }
    }
    if (q < 6) { STAMP(5 + 2 * q); }
    if (q + 1 < NCHUNK) lds_barrier();
  }
#undef DROP_CHUNK
  }  // epilogue
  STAMP(2);
#ifdef TAD_GEMM_ABLATION
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(3);
  }
  ++stamp_i;
#endif
  if (!has_next) break;
  if (!DIRECT) lds_barrier();  // epilogue reads of the LDS are done before the next tile's DMAs overwrite it
  }  // tile loop
}

// ------------------------------------------------------------------------------------------------------------
// gemm_tn: slab[s][n][k] = sum_{m in split s} P[m][n] * Q[m][k]
struct GemmTN {
  const uint16_t* P;  // [Mr, N]  (dy)
  const uint16_t* Q;  // [Mr, K]  (x)
  float* slab;        // [splits][N][K]
  float* bias_slab;   // [splits * tiles_k][N] partial column sums of P (bias gradient), or null
  int Mr, N, K;
  int rows_per_split;  // multiple of 64
  int debug;           // ablation bits as in GemmNT
  // PAIR (gemm_tn_w4_kernel only; launch_gemm_tn_pair): TWO weight gradients with the same Mr and K in one launch.  Output rows [0, N1) are
  // P^T Q of the first problem (P [Mr, N1]), rows [N1, N) those of the second (P2 [Mr, N - N1], Q2 [Mr, K]); N1 is a multiple of 256 so that no
  // tile straddles the two, and only the first problem has bias column sums.  N1 = 0: one problem
  const uint16_t* P2;
  const uint16_t* Q2;
  int N1;
};

// swizzle of the 16-byte chunk index within a tile row (rows are >= 256 bytes); changes bits 1..3 only
__device__ __forceinline__ int sw_tn(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

// Transposed 16x16x32 fragments from a [64 reduction rows][row bytes] tile: lane (g = lane>>4, li = lane&15) supplies rows
// 32ks + 8g + (li>>2) (+4 for the second read), columns col0 + 4*(li&3); it receives column col0 + li, reduction rows 32ks + 8g + 0..7.

// PDEEP (the 256 x 256 two-stage configuration): the P and Q halves of a stage live in rings of their own, THREE slots for P and two
// for Q (3 x 32 + 2 x 32 KiB = the whole 160 KiB of LDS), and the P half is requested TWO reduction tiles ahead.  The loop is bound by
// the round trip of a stage's DMA, not by its bytes (staging and barriers alone take 75 % of the kernel's time, docs/DESIGN_HISTORY.md
// section 8): with 96 instead of 64 KiB in flight per CU a first-touch miss has half a tile longer to arrive.  Measured (round 4,
// tools/exp_tn_pdeep.py): bit-identical and NOT faster (four dW shapes 716.4 vs 717.6 us) -- the staging is bound by its rate through
// the DMA path (~37 GB/s per CU), not by latency; kept behind tad_linear_tuning("tn_pdeep"), off.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool PDEEP = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_tn_kernel(const GemmTN p) {
  static_assert(!PDEEP || STAGES == 2, "PDEEP extends the two-stage ring");
  constexpr int BKT = BK;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MREP = WTM / 16, NREP = WTN / 16;
  constexpr int PROW = BM * 2, QROW = BN * 2;  // bytes per LDS row
  constexpr int KSTEPS = BKT / 32;             // reduction rows per stage: 64 (two k-steps) or 32 (one, deeper ring)
  constexpr int P_BYTES = BKT * PROW, Q_BYTES = BKT * QROW;
  constexpr int STAGE_BYTES = P_BYTES + Q_BYTES;
  constexpr int P_PIECES = P_BYTES / 1024 / NW, Q_PIECES = Q_BYTES / 1024 / NW;
  constexpr int P_LPR = PROW / 16, Q_LPR = QROW / 16;  // lanes (16-B chunks) per row
  static_assert(P_BYTES % (1024 * NW) == 0 && Q_BYTES % (1024 * NW) == 0, "tile must split into 1-KiB DMA pieces per wave");
  static_assert(P_LPR <= 64 && Q_LPR <= 64 && PROW >= 256 && QROW >= 256, "row length");
  constexpr int LOADS = P_PIECES + Q_PIECES;
  constexpr int LDS_BYTES_TN = PDEEP ? 3 * P_BYTES + 2 * Q_BYTES : STAGES * STAGE_BYTES;
  static_assert(LDS_BYTES_TN <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES_TN];

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
  const int nt = p.rows_per_split / BKT;

  const int p_bytes = (int)((int64_t)p.Mr * p.N * 2), q_bytes = (int)((int64_t)p.Mr * p.K * 2);

  // DMA: piece = 1 KiB = (1024/PROW) rows; lane -> row lane / P_LPR, physical chunk lane % P_LPR
  uint32_t p_off[P_PIECES], q_off[Q_PIECES];
#pragma unroll
  for (int i = 0; i < P_PIECES; ++i) {
    const int piece = i * NW + wave;
    const int row = piece * (64 / P_LPR) + lane / P_LPR;
    const int chunk = (lane % P_LPR) ^ ((DBG_BITS(p) & 16) ? 0 : sw_tn(row));  // (debug 16: unswizzled source, timing experiments only)
    // columns beyond N only feed outputs that are never stored; clamp keeps the address inside the row
    int col = n0 + chunk * 8;
    if (col > p.N - 8) col = p.N - 8;
    p_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(p.N * 2) + (uint32_t)(col * 2);
  }
#pragma unroll
  for (int i = 0; i < Q_PIECES; ++i) {
    const int piece = i * NW + wave;
    const int row = piece * (64 / Q_LPR) + lane / Q_LPR;
    const int chunk = (lane % Q_LPR) ^ ((DBG_BITS(p) & 16) ? 0 : sw_tn(row));
    int col = k0 + chunk * 8;
    if (col > p.K - 8) col = p.K - 8;
    q_off[i] = (uint32_t)(mr0 + row) * (uint32_t)(p.K * 2) + (uint32_t)(col * 2);
  }
#define STAGE_TN(buf, t) \
  stage_tile<P_PIECES, NW>(p.P, p_bytes, lds + (buf) * STAGE_BYTES, p_off, (uint32_t)(t) * BKT * (uint32_t)(p.N * 2), wave); \
  stage_tile<Q_PIECES, NW>(p.Q, q_bytes, lds + (buf) * STAGE_BYTES + P_BYTES, q_off, (uint32_t)(t) * BKT * (uint32_t)(p.K * 2), wave)
  // PDEEP: P ring = slots 0..2 at the bottom of the LDS, Q ring = slots 0..1 behind it
#define STAGE_P(slot, t) stage_tile<P_PIECES, NW>(p.P, p_bytes, lds + (slot) * P_BYTES, p_off, (uint32_t)(t) * BKT * (uint32_t)(p.N * 2), wave)
#define STAGE_Q(slot, t) stage_tile<Q_PIECES, NW>(p.Q, q_bytes, lds + 3 * P_BYTES + (slot) * Q_BYTES, q_off, (uint32_t)(t) * BKT * (uint32_t)(p.K * 2), wave)

  // transposed fragment reads: 16-lane group g = lane>>4 covers reduction rows 8g..8g+7 of a 32-deep k-step;
  // lane i = lane&15 of the group supplies row (i>>2) (+4 for the second read), columns c0 + 4*(i&3) .. +3
  const int g = lane >> 4, li = lane & 15;
  // bias gradient = column sums of P = P^T * ones: one extra MFMA per row fragment against an all-ones B operand.  The work is
  // spread evenly (the grid is one workgroup per CU, so any imbalance is pure idle time): the tiles_k workgroups that share a
  // row panel take turns over the reduction tiles (t % tiles_k == own column-panel index), and inside a workgroup the WAVES_N
  // waves that share the same rows split the row fragments (i % WAVES_N == wn).  Partials: bias_slab[split*tiles_k + tk_][N].
  const bool bias_on = (p.bias_slab != nullptr);
  static_assert(MREP % WAVES_N == 0, "bias fragments split across the waves of a row");
  constexpr int BREP = MREP / WAVES_N;
  op16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (op16_t)1.0f;
  f32x4 bacc[BREP];
#pragma unroll
  for (int i = 0; i < BREP; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment reads are inline asm (lds_tr16_b64, common.h): with the builtin the compiler drained the LDS-DMA of the next stage
  // (s_waitcnt vmcnt(0)) in front of the first read after it had been issued, i.e. staging and matrix work ran one after the other.
  // Byte offset of a fragment's first read inside its tile for k-step 0; the second read is 4 rows further, k-step 1 32 rows
  // further (neither changes the swizzle: sw_tn looks at row bits 0, 1 and 3), both as instruction immediates.
  static_assert(NREP == 4, "the waits below are written for four column fragments per wave");
  uint32_t p_rd[MREP], q_rd[NREP];
  {
    const int r0 = 8 * g + (li >> 2);
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      const int col = wm * WTM + 16 * i + 4 * (li & 3);
      p_rd[i] = (uint32_t)(r0 * PROW + (((col >> 3) ^ sw_tn(r0)) << 4) + (col & 7) * 2);
    }
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int col = wn * WTN + 16 * j + 4 * (li & 3);
      q_rd[j] = (uint32_t)(r0 * QROW + (((col >> 3) ^ sw_tn(r0)) << 4) + (col & 7) * 2);
    }
  }
  const uint32_t lds0 = lds_addr(lds);

  if (PDEEP) {
    if (0 < nt) { STAGE_P(0, 0); STAGE_Q(0, 0); }
    if (1 < nt) { STAGE_P(1, 1); }
  } else {
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
      if (st < nt) { STAGE_TN(st, st); }
  }
  int rd = 0, wr = STAGES - 1;
  int p_rdslot = 0, q_rdslot = 0;  // (PDEEP) ring slots of reduction tile t
  for (int t = 0; t < nt; ++t) {
    if (PDEEP) {
      // outstanding, oldest first: P(t), Q(t), P(t+1) -- the first two have to be there, the pieces of P(t+1) may still be in flight
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      wait_stage<LOADS>(min(STAGES - 2, nt - 1 - t));
    }
    block_barrier();
    const uint32_t st_addr = lds0 + (uint32_t)(rd * STAGE_BYTES);
    const uint32_t p_addr = PDEEP ? lds0 + (uint32_t)(p_rdslot * P_BYTES) : st_addr;
    const uint32_t q_addr = PDEEP ? lds0 + (uint32_t)(3 * P_BYTES + q_rdslot * Q_BYTES) : st_addr;
    constexpr int QIMM = PDEEP ? 0 : P_BYTES;  // where the Q half starts relative to q_addr
    const bool more = PDEEP ? (t + 1 < nt) : (t + STAGES - 1 < nt);
    const bool bias_now = bias_on && (t % tiles_k == tk_);
    const int wr_now = wr, t_next = t + STAGES - 1;
    rd = (rd + 1 == STAGES) ? 0 : rd + 1;
    wr = (wr + 1 == STAGES) ? 0 : wr + 1;
    // (PDEEP) what this tile's DMA point issues: Q(t+1) into the slot Q(t-1) left, then P(t+2) into the slot P(t-1) left -- in that
    // order, so that P(t+2) is the youngest when the next tile waits
    const int q_wrslot = q_rdslot ^ 1, p_wrslot = (p_rdslot == 0) ? 2 : p_rdslot - 1;
    const bool more_p = t + 2 < nt;
    q_rdslot ^= 1;
    p_rdslot = (p_rdslot == 2) ? 0 : p_rdslot + 1;
#define KSTEP_TN(ks) \
  if (!(DBG_BITS(p) & 8)) {                                                                                   \
    s16x4 ql_[NREP], qh_[NREP], pl_[MREP], ph_[MREP];                                                         \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j) {                                                        \
      ql_[j] = lds_tr16_b64<QIMM + (ks) * 32 * QROW>(q_addr + q_rd[j]);                                       \
      qh_[j] = lds_tr16_b64<QIMM + (ks) * 32 * QROW + 4 * QROW>(q_addr + q_rd[j]);                            \
    }                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < MREP; ++i) {                                                        \
      pl_[i] = lds_tr16_b64<(ks) * 32 * PROW>(p_addr + p_rd[i]);                                              \
      ph_[i] = lds_tr16_b64<(ks) * 32 * PROW + 4 * PROW>(p_addr + p_rd[i]);                                   \
    }                                                                                                         \
    /* the four column fragments (8 reads), then one row fragment (2 reads) at a time as its MFMAs come up */ \
    lds_wait<2 * MREP>(ql_[0], qh_[0], ql_[1], qh_[1], ql_[2], qh_[2], ql_[3], qh_[3]);                       \
    op16x8 qf[NREP];                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NREP; ++j) qf[j] = join_tr(ql_[j], qh_[j]);                         \
    static_for<0, MREP>([&](auto ic) {                                                                        \
      constexpr int i = decltype(ic)::value;                                                                  \
      lds_wait<2 * (MREP - 1 - i)>(pl_[i], ph_[i]);                                                           \
      const op16x8 pf = join_tr(pl_[i], ph_[i]);                                                              \
      if (!(DBG_BITS(p) & 2)) {                                                                               \
        _Pragma("unroll") for (int j = 0; j < NREP; ++j)                                                      \
            acc[i][j] = TAD_MFMA_16x16x32(qf[j], pf, acc[i][j]);               \
      } else {                                                                                                \
        asm volatile("" ::"v"(pf));                                                                           \
        _Pragma("unroll") for (int j = 0; j < NREP; ++j) asm volatile("" ::"v"(qf[j]));                       \
      }                                                                                                       \
      if (bias_now && (i % WAVES_N) == wn)                                                                    \
        bacc[i / WAVES_N] = TAD_MFMA_16x16x32(pf, ones, bacc[i / WAVES_N]);    \
    });                                                                                                       \
  }
    const bool late = wave >= NW / 2;  // stagger the DMA issue of the two waves that share a SIMD (see gemm_nt_kernel)
    const bool dma = more && !(DBG_BITS(p) & 1);
    static_assert(KSTEPS == 2, "two 32-deep k-steps per stage");
#define ISSUE_TN()                                             \
  if (PDEEP) {                                                   \
    STAGE_Q(q_wrslot, t + 1);                                    \
    if (more_p) { STAGE_P(p_wrslot, t + 2); }                    \
  } else {                                                       \
    STAGE_TN(wr_now, t_next);                                    \
  }
    if (dma && !late) { ISSUE_TN(); }
    KSTEP_TN(0);
    if (dma && late) { ISSUE_TN(); }
    KSTEP_TN(1);
#undef ISSUE_TN
  }

  if (bias_on && li == 0) {
    float* bo = p.bias_slab + ((int64_t)split * tiles_k + tk_) * p.N;
#pragma unroll
    for (int ib = 0; ib < BREP; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wm * WTM + 16 * (ib * WAVES_N + wn) + 4 * g + r;
        if (n < p.N) bo[n] = bacc[ib][r];
      }
  }
  // The Q (k) fragments feed the MFMA's row operand, so D[row = k][col = n]: lane (li, g) holds k = 4g .. 4g+3 of output row
  // n = li -- one 16-byte store per fragment (16 rows x 64 contiguous bytes per instruction) instead of four 4-byte stores.
  // K % 8 == 0, so a group of four k is either wholly inside or wholly outside.
  float* out = p.slab + (int64_t)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int n = n0 + wm * WTM + 16 * i + li;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int k = k0 + wn * WTN + 16 * j + 4 * g;
      if (n < p.N && k < p.K)
        *reinterpret_cast<float4*>(out + (int64_t)n * p.K + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

#ifndef TAD_GEMM_W4_TU  // (csrc/gemm_w4.hip includes this file for the kernel templates and parameter blocks only)
// ------------------------------------------------------------------------------------------------------------

static int cu_count();

TAD_NAMESPACE_END
// Scheduling knobs and counters of the Linear GEMMs (tad_linear_tuning; initial values from the environment).  One copy for the whole
// library: defined by the bf16 compilation pass, shared by the half pass.
namespace tad { namespace knobs {
#ifndef TAD_OPND_F16
static int env_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
int gemm_debug = env_int("TAD_GEMM_DEBUG");  // ablation bits (GemmNT::debug); only ablation builds look at them
int nt_persist = !env_int("TAD_GEMM_NO_PERSIST");
int nt_direct = getenv("TAD_GEMM_DIRECT_EPI") ? env_int("TAD_GEMM_DIRECT_EPI") : 1;
int nt_split = getenv("TAD_GEMM_SPLIT_TAIL") ? env_int("TAD_GEMM_SPLIT_TAIL") : 1;
int nt_sk_defer = getenv("TAD_GEMM_SPLITK_DEFER") ? env_int("TAD_GEMM_SPLITK_DEFER") : 1;  // 1 = split-K tails leave their partial tiles in front of the whole-round launch and are combined behind it; 0 = combined inside one launch
int nt_splitk = getenv("TAD_GEMM_SPLITK_TAIL") ? env_int("TAD_GEMM_SPLITK_TAIL") : 1;  // 1 = the tail launch of the split plan may split its tiles along K (needs a workspace); 0 = never; 2 = whenever eligible
int nt_variant = env_int("TAD_GEMM_NT_VARIANT");  // 0 = planned per shape (launch_gemm_nt), else the tile configuration for every launch
int nt_group_m_knob = getenv("TAD_GEMM_GROUP_M") ? env_int("TAD_GEMM_GROUP_M") : 0;  // 0 = per-shape choice (nt_group_m)
int tn_variant = env_int("TAD_GEMM_TN_VARIANT");
// > 0: bias-only Linears with K >= this run their whole rounds on the four-wave kernel.  Measured (tools/exp_w4_plain.py, planned launches at M = 50176):
// qkv forward 176.9 / 177.2 us (eight / four waves), dX(proj) 78.0 / 73.9, dX(qkv) 159.0 / 148.4, dX(fc1) 225.4 / 208.5 -- on by default
int nt_w4_plain = getenv("TAD_GEMM_W4_PLAIN") ? env_int("TAD_GEMM_W4_PLAIN") : 640;  // K_min of the Linears whose whole rounds of 256 x 256 tiles run on the four-wave kernel (0 = none): below ~640 its epilogue costs more than its K loop gains (tools/exp_w4_plain.py --D 384 / 512)
// bit mask: which other epilogues' whole rounds take the four-wave kernel (see nt_main_variant).  Measured (tools/exp_gemm_knobs.py --configs
// "w4_epilogues=0;w4_epilogues=14", eight / four waves): proj + residual 96.0 / 96.9 us, fc2 + residual 250.4 / 241.1, fc1 GELU 278.1 / 282.1, dX(fc2) GELU backward
// 291.2 / 338.8 -- the residual epilogue (bit 2) is on by default, the vector-heavy GELU ones stay on eight waves
int nt_w4_epilogues = getenv("TAD_GEMM_W4_EPILOGUES") ? env_int("TAD_GEMM_W4_EPILOGUES") : 4;
int tn_pair = getenv("TAD_GEMM_TN_PAIR") ? env_int("TAD_GEMM_TN_PAIR") : 1;  // 1: tad_linear_bwd_weight_pair runs its two problems as one launch when they fit (launch_gemm_tn_pair); 0: always two launches
int nt_short_k = getenv("TAD_GEMM_SHORT_K") ? env_int("TAD_GEMM_SHORT_K") : 1;  // 1: the short-K plan of launch_gemm_nt (K <= 512: tiles that put two workgroups on a CU)
int nt_tail_192 = getenv("TAD_GEMM_TAIL_192") ? env_int("TAD_GEMM_TAIL_192") : 1;  // 1: tails of the split plan may run as 192 x 128 tiles (nt_tail_variant)
int tn_w4 = getenv("TAD_GEMM_TN_W4") ? env_int("TAD_GEMM_TN_W4") : 1;  // 1: the 256 x 256 weight-gradient GEMM runs as four waves of 128 x 128 (gemm_w4.hip)
int tn_pdeep = env_int("TAD_GEMM_TN_PDEEP");  // 1: gemm_tn 256 x 256 with the P operand two reduction tiles ahead (see PDEEP); measured null (round 4), off
unsigned long long* nt_stamps = nullptr;
long long nt_launches = 0;  // gemm_nt kernel launches so far (tad_linear_kernel_launches)
#else
extern int gemm_debug, nt_persist, nt_direct, nt_split, nt_splitk, nt_variant, nt_group_m_knob, tn_variant, tn_pdeep, nt_sk_defer, tn_w4, nt_w4_plain, nt_w4_epilogues, nt_tail_192, nt_short_k, tn_pair;
extern unsigned long long* nt_stamps;
extern long long nt_launches;
#endif
}}  // namespace tad::knobs
TAD_NAMESPACE_BEGIN
using namespace knobs;

// Row panels per column-panel group of the tile raster.  The ~32 workgroups resident on an XCD (private 4 MiB L2) walk consecutive
// tile ids, so at any time they touch GROUP_M A-panels and ~32 / GROUP_M W-panels; each panel streams K-tile by K-tile, and a
// panel's K-tile is fetched from beyond L2 once per group of tiles that share it while they run together.  Per group of
// GROUP_M x tiles_n tiles that is GROUP_M A-panel loads (algorithmic) + tiles_n W-panel loads (overhead, amortised over GROUP_M).
// Measured (tools/exp_gemm_knobs.py --knob group_m, the eight Linear shapes of a ViT-B block, with the non-temporal output stores in): 2 / 4 / 8 are
// within 0.5 % of each other in sum; the long reductions (K = 2304 / 3072: an A panel of 256 rows is 1.2 - 1.5 MB) are 1 - 4 % faster with 4 (fc2 241
// vs 246 us, dX(fc1) 220 vs 222), the K = 768 shapes with 8 (qkv 166 vs 169).
static int nt_group_m(int tiles_m, int tiles_n, int K) {
  if (nt_group_m_knob > 0) return nt_group_m_knob;
  return K >= 2048 ? 4 : 8;
}


// Tile configurations.  NT: 1 = 256x256 (2x4 waves) 2 stages; 2 = 128x128 (2x2) 2 stages, 2 workgroups/CU;
// 3 = 256x128 (4x2) 3 stages; 4 = 128x64, 5 = 64x64 (2x2 waves, 2 stages: small problems).  0 = auto.  The epilogue kind and output type are compile-time (the epilogue is VALU-bound).
// Variants 1 and 3 run as persistent kernels (one workgroup per CU walks a tile list) once there are more than 1.5 tiles per CU.
template <int EPI, bool OUT_BF16>
static void launch_nt_variant(int v, GemmNT& p, hipStream_t st) {
  auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const int no_persist = !nt_persist;
  if (((EPI == EPI_RESIDUAL && OUT_BF16) || EPI == EPI_RESMOD) && (v == 1 || v == 7)) v = 3;  // (not instantiated: no registers / never needed)
  if (v != 2 && v != 4 && v != 5 && p.rowscale && p.rows_per_scale < 256) v = 2;  // the 256-row tiles take at most two row-scale groups per tile
  const int grid_p = cu_count() & ~7;
  const int bn = v == 1 ? 256 : 128;
  const bool persist = !no_persist && (v == 1 || v == 3) && grid_p >= 8 && tiles(256, bn) > grid_p + grid_p / 2;
  p.group_m = nt_group_m((p.M + 255) / 256, (p.N + bn - 1) / bn, p.K);
#define NT_LAUNCH(BM_, BN_, WM_, WN_, ST_, PER_, DIR_, GRID_, THREADS_)                                                              \
  hipLaunchKernelGGL((gemm_nt_kernel<BM_, BN_, WM_, WN_, ST_, EPI, OUT_BF16, PER_, DIR_>), dim3(GRID_), dim3(THREADS_), 0, st, p)
  // measured per shape (tools/exp_epilogue.py): storing straight from the MFMA layout wins only for the bias-only bf16 epilogue
  // (nothing to fetch, no arithmetic); the others keep the LDS transposition.  nt_direct: 0 = never, 1 = auto, 2 = always.
  const bool direct = nt_direct == 2 || (nt_direct == 1 && EPI == EPI_PLAIN && OUT_BF16);
  switch (v) {
    case 1:
      if constexpr (!(EPI == EPI_RESIDUAL && OUT_BF16) && EPI != EPI_RESMOD) {
        if (persist) { if (direct) NT_LAUNCH(256, 256, 2, 4, 2, true, true, grid_p, 512); else NT_LAUNCH(256, 256, 2, 4, 2, true, false, grid_p, 512); }
        else { if (direct) NT_LAUNCH(256, 256, 2, 4, 2, false, true, tiles(256, 256), 512); else NT_LAUNCH(256, 256, 2, 4, 2, false, false, tiles(256, 256), 512); }
      }
      break;
    case 3:
      if (persist) { if (direct) NT_LAUNCH(256, 128, 4, 2, 3, true, true, grid_p, 512); else NT_LAUNCH(256, 128, 4, 2, 3, true, false, grid_p, 512); }
      else { if (direct) NT_LAUNCH(256, 128, 4, 2, 3, false, true, tiles(256, 128), 512); else NT_LAUNCH(256, 128, 4, 2, 3, false, false, tiles(256, 128), 512); }
      break;
    case 7: {  // the four-wave 256 x 256 kernels (gemm_w4.hip); persistent under the same rule as variant 1
      if (p.K < 2 * BK) { NT_LAUNCH(256, 256, 2, 4, 2, false, false, tiles(256, 256), 512); break; }  // (its K loop is written for >= 2 K-tiles)
      const bool per7 = !no_persist && grid_p >= 8 && tiles(256, 256) > grid_p + grid_p / 2;
      (void)launch_gemm_nt_w4(p, per7 ? grid_p : 0, st);
      break;
    }
    case 8:  // 192 x 128 (4 x 2 waves, 3 stages): tails of the split plan whose 256 x 128 tiles would leave > a third of the CUs idle
      if (direct) NT_LAUNCH(192, 128, 4, 2, 3, false, true, tiles(192, 128), 512); else NT_LAUNCH(192, 128, 4, 2, 3, false, false, tiles(192, 128), 512);
      break;
    case 9:  // 192 x 128 as FOUR waves (2 x 2 of 96 x 64), 2 stages of 40 KiB: two workgroups per CU (the short-K plan of launch_gemm_nt)
      if (direct) NT_LAUNCH(192, 128, 2, 2, 2, false, true, tiles(192, 128), 256); else NT_LAUNCH(192, 128, 2, 2, 2, false, false, tiles(192, 128), 256);
      break;
    case 4:
      if (direct) NT_LAUNCH(128, 64, 2, 2, 2, false, true, tiles(128, 64), 256); else NT_LAUNCH(128, 64, 2, 2, 2, false, false, tiles(128, 64), 256);
      break;
    case 5:
      if (direct) NT_LAUNCH(64, 64, 2, 2, 2, false, true, tiles(64, 64), 256); else NT_LAUNCH(64, 64, 2, 2, 2, false, false, tiles(64, 64), 256);
      break;
    default:
      if (direct) NT_LAUNCH(128, 128, 2, 2, 2, false, true, tiles(128, 128), 256); else NT_LAUNCH(128, 128, 2, 2, 2, false, false, tiles(128, 128), 256);
      break;
  }
#undef NT_LAUNCH
}

// Split-K launch of a (small) problem: tiles x splits workgroups, all resident (the caller checked tiles * splits <= CUs).  Workspace:
// [arrival counters, one per tile | error word][partial tiles].  The counters are zeroed on the stream in front of the launch.
constexpr size_t SK_HEADER_BYTES = 4096;  // counters (<= 1008 tiles)
constexpr size_t SK_TILE_BYTES = 256 * 256 * sizeof(float);
// The in-launch combine (mode 0) waits for the other shares of its tile with a bounded spin; a share that never arrives (the grid was
// not co-resident: a side stream or another process held CUs) leaves stale partial sums in the output.  The kernel then sets this word,
// which lives in pinned, device-mapped host memory so that the host can see it without a synchronisation: every later gemm_nt launch
// checks it first and returns TAD_ELAUNCH (the word used to sit in the workspace, where nobody read it).
static unsigned* sk_error_word(unsigned** dev_ptr) {
  static unsigned* host = nullptr;
  static unsigned* dev = nullptr;
  if (!host) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return nullptr;
    host = (unsigned*)h;
    dev = (unsigned*)d;
    *host = 0u;
  }
  if (dev_ptr) *dev_ptr = dev;
  return host;
}
static bool sk_error_pending_armed = false;  // (the word exists only once a mode-0 launch has been made)
static int sk_check_pending_error() {
  if (!sk_error_pending_armed) return TAD_OK;
  unsigned* host = sk_error_word(nullptr);
  if (host && __atomic_load_n(host, __ATOMIC_RELAXED)) {
    __atomic_store_n(host, 0u, __ATOMIC_RELAXED);
    set_error("gemm_nt: an earlier in-launch split-K combine (tad_linear_tuning(\"splitk_defer\", 0)) gave up waiting for a share of its tile -- "
              "its grid was not co-resident (another stream or process held CUs); the output of that Linear is INVALID.  Use the default "
              "deferred plan (splitk_defer = 1), which has no residency requirement");
    return TAD_ELAUNCH;
  }
  return TAD_OK;
}

static int launch_gemm_nt_splitk(GemmNT p, int splits, void* ws, hipStream_t st, int mode = 0) {
  ++nt_launches;
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  p.sk_splits = splits;
  p.sk_mode = mode;
  p.sk_cnt = (unsigned*)ws;
  p.sk_err = nullptr;
  if (mode == 0) {
    if (!sk_error_word(&p.sk_err)) { set_error("gemm_nt: split-K error word allocation failed"); return TAD_ELAUNCH; }
    sk_error_pending_armed = true;
  }
  p.sk_ws = (float*)((char*)ws + SK_HEADER_BYTES);
  p.group_m = nt_group_m((p.M + 255) / 256, (p.N + 255) / 256, p.K);
  if (mode == 0 && hipMemsetAsync(ws, 0, SK_HEADER_BYTES, st) != hipSuccess) { set_error("gemm_nt: split-K counter reset failed"); return TAD_ELAUNCH; }
  const dim3 grid(tiles * splits), block(512);
  if (p.epi == EPI_PLAIN && p.c_bf16) hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 4, 2, EPI_PLAIN, true, false, false, true>), grid, block, 0, st, p);
  else if (p.epi == EPI_PLAIN) hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 4, 2, EPI_PLAIN, false, false, false, true>), grid, block, 0, st, p);
  else if (p.epi == EPI_RESIDUAL && !p.c_bf16) hipLaunchKernelGGL((gemm_nt_kernel<256, 256, 2, 4, 2, EPI_RESIDUAL, false, false, false, true>), grid, block, 0, st, p);
  else { set_error("gemm_nt: no split-K kernel for epilogue %d", p.epi); return TAD_EINVAL; }
  return check_launch("gemm_nt_splitk");
}
// ---- geometry of the split plan, ONE copy for launch_gemm_nt, nt_splitk_plan and tad_linear_workspace_bytes (the workspace query
// used to restate it, and any drift made the split-K tail drop out without a diagnostic)
// Largest row count one launch may cover (32-bit operand offsets, see launch_gemm_nt); esz = bytes of the widest element the epilogue touches
static int64_t nt_max_rows(int N, int K, int64_t esz) {
  int64_t max_rows = ((1ll << 31) - 1) / ((int64_t)N * esz) - 256;
  const int64_t a_rows = ((1ll << 32) - 1) / ((int64_t)K * 2);
  if (a_rows < max_rows) max_rows = a_rows;
  return max_rows / 256 * 256;
}
// Rows that fill whole rounds of one 256 x 256 tile per CU (0: the split plan does not apply); *tail_tiles = the 256 x 256 tiles left behind them
static int nt_main_rows(int64_t M, int N, int64_t* tail_tiles) {
  const int grid = cu_count() & ~7;
  const int64_t tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256;
  const int64_t rounds = grid > 0 ? tiles_m * tiles_n / grid : 0;
  const int64_t panels = rounds > 0 ? rounds * grid / tiles_n : 0;
  if (!(panels > 0 && panels < tiles_m)) return 0;
  if (tail_tiles) *tail_tiles = (tiles_m - panels) * tiles_n;
  return (int)(panels * 256);
}
// Shares per tile of a split-K tail of `tiles` tiles with nk K-tiles each (0: it does not split): every share needs at least two K-tiles and all
// shares must be resident at once
static int nt_splitk_shares(int64_t tiles, int nk) {
  const int cus = cu_count();
  if (!nt_splitk || tiles <= 0 || tiles > cus / 2 || tiles > 1008) return 0;
  // measured (tools/exp_splitk.py, round 4): the 78-tile tails of ViT-B's N = 768 Linears (30 % of the CUs busy for one K loop) do NOT
  // gain -- three shares of 16 K-tiles + the combine take as long as one 256 x 128 K loop of 48 -- while the 16-tile tails of ViT-L's
  // N = 1024 Linears (6 % of the CUs) do: auto mode takes tails of at most a quarter of the CUs whose K loop is long enough
  if (nt_splitk == 1 && (tiles > cus / 4 || nk < 64)) return 0;  // (K = 3072 at 16 tiles: 289 -> 297 us; K = 4096: 424 -> 392, 387 -> 376)
  int64_t s = cus / tiles;
  if (s > nk / 2) s = nk / 2;
  if (s > 8) s = 8;
  return s < 2 ? 0 : (int)s;
}
static size_t nt_splitk_ws_bytes(int64_t tiles, int shares) { return SK_HEADER_BYTES + (size_t)(tiles * shares) * SK_TILE_BYTES; }

// splits for a tail of `tiles` 256 x 256 tiles with nk K-tiles each, or 0: the combine pass costs ~14 us (partial tiles written and read back,
// arrival wait), so short reductions gain nothing
static int nt_splitk_plan(const GemmNT& t, size_t ws_bytes, double* cost_us) {
  const int tiles = ((t.M + 255) / 256) * ((t.N + 255) / 256), nk = t.K / BK;
  // Deferred plan (nt_sk_defer, launch_gemm_nt): the partial tiles are left by a launch IN FRONT of the whole-round launch and combined by a
  // launch BEHIND it, so their way through memory (58 MB each way for 76 tiles x 3 shares) runs beside two rounds of matrix work instead of
  // at the end of a 26 us kernel.  Measured (kernel trace, ViT-B dX(fc1), 76 tiles x 3 shares): partial-tile launch 54 us with write-through
  // stores, whole rounds 190, combine 15 -- against 190 + 47 for the 256 x 128 tail: the 58 MB of partial tiles cost what the balanced
  // K loop saves, so the rule for WHICH tails split stays what it was; the deferred form is the better way to run those that do
  // (ViT-L fc2 421 -> 388 us against 392 combined in the launch, dX(fc1) 387 -> 364 against 376).
  if (!((t.epi == EPI_PLAIN) || (t.epi == EPI_RESIDUAL && !t.c_bf16 && t.res_mod <= 0))) return 0;
  if (t.rowscale && t.rows_per_scale < 256) return 0;
  if (t.colscale_cols > 0) return 0;
  const int s = nt_splitk_shares(tiles, nk);
  if (!s) return 0;
  if (ws_bytes < nt_splitk_ws_bytes(tiles, s)) {
    // (a workspace was handed in but is smaller than tad_linear_workspace_bytes says for this shape under the current knobs: say so once instead
    //  of silently running the unsplit tail)
    static bool warned = false;
    if (!warned) { warned = true; fprintf(stderr, "[tad] gemm_nt: split-K workspace of %zu bytes < the %zu the tail of M=%d N=%d K=%d needs; running it unsplit\n", ws_bytes, nt_splitk_ws_bytes(tiles, s), t.M, t.N, t.K); }
    return 0;
  }
  *cost_us = nt_sk_defer ? (double)((nk + s - 1) / s) * 1.45 + 10.0 + 16.0 : (double)((nk + s - 1) / s) * 1.65 + 14.0 + 3.0;
  return s;
}

static int launch_gemm_nt_one(GemmNT p, int v, hipStream_t st) {
  ++nt_launches;
  if (p.epi == EPI_RESIDUAL && p.residual && p.res_mod > 0) p.epi = EPI_RESMOD;
#define NT_CASE(E)                                                       \
  case E:                                                                \
    if (p.c_bf16) launch_nt_variant<E, true>(v, p, st);                  \
    else launch_nt_variant<E, false>(v, p, st);                          \
    break;
  switch (p.epi) {
    NT_CASE(EPI_PLAIN)
    NT_CASE(EPI_GELU)
    NT_CASE(EPI_RESIDUAL)
    NT_CASE(EPI_DGELU)
    NT_CASE(EPI_RESMOD)
    default: set_error("gemm_nt: bad epilogue %d", p.epi); return TAD_EINVAL;
  }
#undef NT_CASE
  return check_launch("gemm_nt");
}

// rows [r0, r0 + rows) of the problem as a problem of its own
static GemmNT row_range(const GemmNT& p, int r0, int rows) {
  GemmNT q = p;
  const size_t esz = p.c_bf16 ? 2 : 4;
  q.M = rows;
  q.row_base = p.row_base + r0;
  q.A = p.A + (size_t)r0 * p.K;
  q.C = (char*)p.C + (size_t)r0 * p.N * esz;
  if (p.residual && p.res_mod <= 0) q.residual = p.residual + (size_t)r0 * p.N;
  if (p.preact) q.preact = p.preact + (size_t)r0 * p.N;
  if (p.dgelu_h) q.dgelu_h = p.dgelu_h + (size_t)r0 * p.N;
  return q;
}

// Rough cost (us) of one launch on MI355X, from the in-kernel timelines (tools/exp_timeline.py): K-tile of a 256 x 256 tile
// 1.65 us, of a 256 x 128 tile 1.06 us; epilogue per tile by kind.  Only used to rank plans.
static double nt_cost(int v, int epi, int c_bf16, int M, int N, int K) {
  const int cus = cu_count();
  const double nk = K / 64;
  if (v == 1) {
    const double epi_us = (epi == EPI_GELU || epi == EPI_DGELU) ? 9.0 : (epi == EPI_RESIDUAL ? 12.5 : 4.5);
    const int64_t tiles = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    return (double)((tiles + cus - 1) / cus) * (nk * 1.65 + epi_us) + 3.0;
  }
  if (v == 3) {
    const double epi_us = (epi == EPI_GELU || epi == EPI_DGELU) ? 5.5 : (epi == EPI_RESIDUAL ? 6.5 : 3.0);
    const int64_t tiles = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
    return (double)((tiles + cus - 1) / cus) * (nk * 1.06 + epi_us) + 3.0;
  }
  if (v == 8) {  // 192 x 128: 7 / 8 of the 256 x 128 tile's time per K-tile for 3 / 4 of its rows (tools/exp_tail_tile.py)
    const double epi_us = (epi == EPI_GELU || epi == EPI_DGELU) ? 4.5 : (epi == EPI_RESIDUAL ? 5.5 : 2.5);
    const int64_t tiles = (int64_t)((M + 191) / 192) * ((N + 127) / 128);
    return (double)((tiles + cus - 1) / cus) * (nk * 0.93 + epi_us) + 3.0;
  }
  const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  return (double)((tiles + 2 * cus - 1) / (2 * cus)) * (nk * 1.3 + 5.0) + 3.0;
}
// The tile of a tail launch of the split plan (a problem of at least 2048 rows that fills less than a round of 256 x 256 tiles): 256 x 256,
// 256 x 128, or -- tad_linear_tuning("tail_192", 1), default -- 192 x 128 where that puts more CUs to work (ViT-B's N = 768 Linears:
// 6656 rows = 156 tiles of 256 x 128 on 256 CUs, or 210 of 192 x 128: the five tails of a block 146.4 -> 132.9 us, bit-identical)
static int nt_tail_variant(const GemmNT& t) {
  const double c1 = nt_cost(1, t.epi, t.c_bf16, t.M, t.N, t.K), c3 = nt_cost(3, t.epi, t.c_bf16, t.M, t.N, t.K);
  const double c8 = nt_tail_192 ? nt_cost(8, t.epi, t.c_bf16, t.M, t.N, t.K) : 1e300;
  const bool v1_ok = !(t.epi == EPI_RESIDUAL && (t.c_bf16 || t.res_mod > 0));
  if (c8 < c3 && (c8 < c1 || !v1_ok)) return 8;
  return (c1 < c3 && v1_ok) ? 1 : 3;
}

// Which 256 x 256 kernel runs the whole rounds of a planned launch: the eight-wave one (1), or -- tad_linear_tuning("w4_plain", K_min): for
// bias-only epilogues whose reduction is at least K_min long -- the four-wave one (7), whose faster K loop outweighs its slower epilogue only
// on long reductions (csrc/gemm_w4.hip; bit-identical either way)
static int nt_main_variant(const GemmNT& p) {
  if (p.K < 2 * BK) return 1;
  if (p.epi == EPI_PLAIN) return (nt_w4_plain > 0 && p.K >= nt_w4_plain) ? 7 : 1;
  // the other epilogues: tad_linear_tuning("w4_epilogues", mask) -- bit 1 GELU, 2 residual (f32 output), 3 GELU backward
  const int bit = p.epi == EPI_GELU ? 1 : (p.epi == EPI_RESIDUAL && !p.c_bf16 && p.res_mod <= 0) ? 2 : p.epi == EPI_DGELU ? 3 : -1;
  return (bit > 0 && ((nt_w4_epilogues >> bit) & 1) && nt_w4_plain > 0 && p.K >= nt_w4_plain) ? 7 : 1;
}

int launch_gemm_nt(const GemmNT& p_in, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0) {
  GemmNT p = p_in;
  if (const int rc = sk_check_pending_error()) return rc;
  if (!(p.M > 0 && p.N > 0 && p.K > 0)) { set_error("gemm_nt: empty problem"); return TAD_EINVAL; }
  if (p.K % BK) { set_error("gemm_nt: K=%d must be a multiple of %d", p.K, BK); return TAD_EINVAL; }
  if (p.N % 4) { set_error("gemm_nt: N=%d must be a multiple of 4", p.N); return TAD_EINVAL; }
  if ((int64_t)p.N * p.K * 2 >= (1ll << 32)) { set_error("gemm_nt: weight operand exceeds 4 GiB"); return TAD_EINVAL; }
  // The kernel addresses A through a buffer descriptor with 32-bit byte offsets (< 4 GiB), and C -- with the residual / pre-activation
  // operands, which have C's shape -- through descriptors whose out-of-range lanes get offset 0x80000000, so every real offset,
  // including the rows of the last, partly filled tile, must stay below 2^31: (M + 256) * N * element size < 2 GiB with the element
  // size of the widest operand the epilogue touches (4 for an f32 output or an f32 residual, else 2).  Rows are independent, so a
  // taller problem runs as row ranges that fit (ViT-B fc1, N = 3072 bf16: 222 clips of 1568 tokens per range; the f32 cap used to
  // apply to bf16 outputs too and refused B > 111).
  {
    const int64_t max_rows = nt_max_rows(p.N, p.K, (p.c_bf16 && !p.residual) ? 2 : 4);
    if (max_rows <= 0) { set_error("gemm_nt: N=%d / K=%d too wide for the 32-bit operand offsets", p.N, p.K); return TAD_EINVAL; }
    if (p.M > max_rows) {
      for (int64_t r0 = 0; r0 < p.M; r0 += max_rows) {
        const int rows = (int)((p.M - r0) < max_rows ? (p.M - r0) : max_rows);
        const int rc = launch_gemm_nt(row_range(p, (int)r0, rows), st, ws, ws_bytes);
        if (rc) return rc;
      }
      return TAD_OK;
    }
  }
  const int forced = nt_variant;
  p.debug = gemm_debug;
  p.stamps = nt_stamps;
  if (forced) return launch_gemm_nt_one(p, forced, st);
  if (p.M < 2048 || p.N < 128) {
    // small problems (batch-1 inference: 1568 or 784 rows): 128 x 128 tiles, or 128 x 64 when those would leave more than half of the
    // CUs without a workgroup (ViT-B proj / fc2 at 1568 rows: 78 tiles of 128 x 128)
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const int64_t t64 = (int64_t)((p.M + 127) / 128) * ((p.N + 63) / 64);
    if (t128 * 2 > cu_count() || p.N < 64) return launch_gemm_nt_one(p, 2, st);
    return launch_gemm_nt_one(p, (t64 * 4 <= 3 * cu_count() && p.M >= 64) ? 5 : 4, st);  // 64 x 64 while even 128 x 64 fills < 3/4 of the CUs
  }
  // Short reductions (ViT-S: K = 384; the MAE decoder: K = 512): the epilogue is as long as the K loop, and tiles that put TWO workgroups on a CU
  // let one's epilogue run beside the other's K loop: 192 x 128 as four waves of 96 x 64 (variant 9: 80 KiB of LDS, <= 210 registers) or
  // 128 x 128 (variant 2).  Measured (tools/exp_gemm_knobs.py --D 384 / 512 --configs "variant=0;variant=9;variant=2"), us, planned 256 x 256 /
  // variant 9 / variant 2 -- K = 512: fc1 + GELU 171 / 152 / 161, dX(fc2) GELU' 165 / 151 / 159, the bias-only and residual shapes 1-7 % slower on
  // either; K = 384: qkv 60.0 / 58.5 / 62, fc1 + GELU 102 / 98.6 / 100, dX(fc2) 99 / 97 / 100.5, dX(proj) 28.5 / 25.0 / 24.8, proj + residual
  // 41.0 / 38 / 36 (N = 384 is 1.5 tiles of 256 columns).  From K = 768 the 256 x 256 tile wins everywhere (fc1 274 / 277 / 290)
  if (nt_short_k && p.K <= 512 && !(p.rowscale && p.rows_per_scale < 256)) {
    const bool act = p.epi == EPI_GELU || p.epi == EPI_DGELU;
    if (p.epi == EPI_RESIDUAL && p.K < 512) return launch_gemm_nt_one(p, 2, st);
    if (act || (p.epi == EPI_PLAIN && p.K < 512)) return launch_gemm_nt_one(p, 9, st);
  }
  const bool v1_ok = !(p.epi == EPI_RESIDUAL && (p.c_bf16 || p.res_mod > 0));  // (those instantiations do not exist)
  // Plans: (a) 256 x 128 tiles, (b) 256 x 256 tiles, (c) 256 x 256 tiles for as many row panels as fill whole rounds of one
  // workgroup per CU, the remaining rows as a second launch with whatever suits that smaller problem.  (c) is what lets the
  // N = 768 Linears of ViT-B use the (27 % faster per flop) 256 x 256 tile: 196 x 3 tiles are 2.3 rounds of 256 workgroups.
  const double cost_a = nt_cost(3, p.epi, p.c_bf16, p.M, p.N, p.K);
  const double cost_b = v1_ok ? nt_cost(1, p.epi, p.c_bf16, p.M, p.N, p.K) : 1e300;
  double cost_c = 1e300;
  int main_rows = 0, tail_splits = 0;
  if (v1_ok && nt_split) {
    main_rows = nt_main_rows(p.M, p.N, nullptr);
    if (main_rows > 0) {
      const int tail = p.M - main_rows;
      double tail_cost = tail < 2048 ? nt_cost(2, p.epi, p.c_bf16, tail, p.N, p.K)
                                     : nt_cost(nt_tail_variant(row_range(p, main_rows, tail)), p.epi, p.c_bf16, tail, p.N, p.K);
      // (d) the tail's tiles split along K over all CUs (SPLITK kernels): 78 tiles of the N = 768 Linears of ViT-B run as 3 shares
      // each on 234 CUs instead of one K loop on 156
      if (ws) {
        double sk_cost = 1e300;
        const int s = nt_splitk_plan(row_range(p, main_rows, tail), ws_bytes, &sk_cost);
        if (s && (sk_cost < tail_cost || nt_splitk == 2)) { tail_splits = s; tail_cost = sk_cost; }
      }
      cost_c = nt_cost(1, p.epi, p.c_bf16, main_rows, p.N, p.K) + tail_cost;
    }
  }
  if (nt_split == 2 && main_rows > 0) cost_c = 0.0;  // forced (experiments)
  if (cost_c < cost_a && cost_c < cost_b) {
    const GemmNT t = row_range(p, main_rows, p.M - main_rows);
    if (tail_splits && nt_sk_defer) {  // partial tiles of the tail | whole rounds | combine + epilogue of the tail
      int rc = launch_gemm_nt_splitk(t, tail_splits, ws, st, 1);
      if (rc) return rc;
      rc = launch_gemm_nt_one(row_range(p, 0, main_rows), nt_main_variant(p), st);
      if (rc) return rc;
      return launch_gemm_nt_splitk(t, tail_splits, ws, st, 2);
    }
    int rc = launch_gemm_nt_one(row_range(p, 0, main_rows), nt_main_variant(p), st);
    if (rc) return rc;
    if (tail_splits) return launch_gemm_nt_splitk(t, tail_splits, ws, st);
    if (t.M < 2048) return launch_gemm_nt_one(t, 2, st);
    return launch_gemm_nt_one(t, nt_tail_variant(t), st);
  }
  return launch_gemm_nt_one(p, cost_b < cost_a ? nt_main_variant(p) : 3, st);
}

// TN: 1 = 256x256 (2x4) 2 stages; 3 = 256x128 (4x2) 3 stages.  Splits over the reduction dim target ~1 workgroup per CU.
// TAD_GEMM_TN_VARIANT forces one; by default the 128-wide tile serves shapes whose K the 256-wide one would pad by a third or more
// (K = 384 = 1.5 tiles: ViT-S -- the weight gradients of its qkv / proj / fc1 Linears computed 2 x 256 columns for 384)
static int tn_variant(int K = 0) {
  if (knobs::tn_variant) return knobs::tn_variant;
  const int pad256 = (K + 255) / 256 * 256, pad128 = (K + 127) / 128 * 128;
  return (K > 0 && pad256 * 3 >= pad128 * 4) ? 3 : 1;
}
static int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
// One workgroup per CU is resident (128 KiB of LDS), so the grid runs in rounds of cu_count() workgroups and a grid of
// cu_count() + 1 costs two full rounds.  Pick the split count that minimises rounds x (reduction tiles per split + epilogue)
// plus the slab-reduce pass, all in units of one reduction tile (~2 us on MI355X).
static int tn_plan(int64_t Mr, int N, int K, int* splits, int* rows_per_split) {
  const int bn = tn_variant(K) != 3 ? 256 : 128;
  const int tiles = ((N + 255) / 256) * ((K + bn - 1) / bn);
  const int64_t ktiles = (Mr + BK - 1) / BK;
  const int cus = cu_count();
  const double epi_units = 6.0;                                            // accumulator store of one workgroup
  const double reduce_units = (double)N * (double)K * 8.0 / 4.0e12 / 2.0e-6;  // slab write + read per split
  int best = 1;
  double best_cost = 1e300;
  for (int s = 1; s <= 64 && s <= ktiles; ++s) {
    const int64_t per = (ktiles + s - 1) / s;
    const int s_eff = (int)((ktiles + per - 1) / per);
    if (s_eff != s) continue;
    const int64_t rounds = ((int64_t)tiles * s + cus - 1) / cus;
    const double cost = (double)rounds * ((double)per + epi_units) + (s > 1 ? s * reduce_units : 0.0);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  const int64_t per = (ktiles + best - 1) / best;
  *splits = best;
  *rows_per_split = (int)(per * BK);
  return tiles;
}

size_t gemm_tn_workspace_bytes(int64_t Mr, int N, int K) {
  int s, r;
  tn_plan(Mr, N, K, &s, &r);
  const int bn = tn_variant(K) != 3 ? 256 : 128;
  return (size_t)s * ((size_t)N * (size_t)K + (size_t)((K + bn - 1) / bn) * (size_t)N) * sizeof(float);
}

// bias_out2 != null: the column sums of the first third of the columns go to bias_out, those of the last third to bias_out2 and the
// middle third is dropped (the qkv Linear: q_bias / no k bias / v_bias, modeling_finetune.py:69-76, 89-92)
int launch_gemm_tn(const uint16_t* P, const uint16_t* Q, float* out, float* bias_out, float* bias_out2, int accumulate, void* ws,
                   size_t ws_bytes, int64_t Mr, int N, int K, hipStream_t st) {
  if (const int rc = sk_check_pending_error()) return rc;  // (a failed in-launch split-K combine is reported by the NEXT tad_linear_* call, whichever it is)
  if (!(Mr > 0 && N > 0 && K > 0)) { set_error("gemm_tn: empty problem"); return TAD_EINVAL; }
  if (N % 8 || K % 8) { set_error("gemm_tn: N=%d and K=%d must be multiples of 8", N, K); return TAD_EINVAL; }
  if (Mr * (int64_t)N * 2 >= (1ll << 32) || Mr * (int64_t)K * 2 >= (1ll << 32)) { set_error("gemm_tn: operand exceeds 4 GiB"); return TAD_EINVAL; }
  GemmTN p{};
  p.P = P; p.Q = Q; p.slab = (float*)ws; p.Mr = (int)Mr; p.N = N; p.K = K;
  p.debug = gemm_debug;
  int splits;
  const int tiles = tn_plan(Mr, N, K, &splits, &p.rows_per_split);
  const int tiles_k = (K + (tn_variant(K) != 3 ? 256 : 128) - 1) / (tn_variant(K) != 3 ? 256 : 128);
  if (ws_bytes < (size_t)splits * ((size_t)N * K + (size_t)tiles_k * N) * sizeof(float)) { set_error("gemm_tn: workspace too small"); return TAD_ENOSPACE; }
  p.bias_slab = bias_out ? p.slab + (size_t)splits * N * K : nullptr;
  // 256 x 256 tiles: the four-wave kernel (gemm_w4.hip; bit-identical results) unless switched off (tad_linear_tuning("tn_w4", 0)); its loop
  // is written for at least two reduction tiles per workgroup
  if (tn_variant(K) == 1 && knobs::tn_w4 && !knobs::tn_pdeep && p.rows_per_split >= 2 * BK && !p.debug) {
    const int rc = launch_gemm_tn_w4(p, tiles * splits, st);
    if (rc) return rc;
  } else if (tn_variant(K) == 1 && knobs::tn_pdeep)
    hipLaunchKernelGGL((gemm_tn_kernel<256, 256, 2, 4, 2, true>), dim3(tiles * splits), dim3(512), 0, st, p);
  else if (tn_variant(K) == 1)
    hipLaunchKernelGGL((gemm_tn_kernel<256, 256, 2, 4, 2>), dim3(tiles * splits), dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<256, 128, 4, 2, 3>), dim3(tiles * splits), dim3(512), 0, st, p);
  int rc = check_launch("gemm_tn");
  if (rc) return rc;
  if (!bias_out) return launch_reduce_partials(p.slab, out, splits, (int64_t)N * K, accumulate, st);
  // slab reduction and bias column sums in one launch (N * K % 4 == 0 and N % 4 == 0 hold: N, K are multiples of 8)
  if (bias_out2) return launch_reduce_dw(p.slab, out, splits, (int64_t)N * K, accumulate, p.bias_slab, splits * tiles_k, N, bias_out, bias_out2, N / 3, 2 * (N / 3), st);
  return launch_reduce_dw(p.slab, out, splits, (int64_t)N * K, accumulate, p.bias_slab, splits * tiles_k, N, bias_out, nullptr, N, 0, st);
}

// Two weight gradients over the same rows with the same K as ONE launch of the four-wave kernel (GemmTN::N1): out1 [N1, K] = P1^T Q1 with its
// bias column sums (bias_out1, or split into bias_out1 / bias_out1b as for the qkv Linear), out2 [N2, K] = P2^T Q2.  A small problem pays for
// filling the chip with many splits -- the 768 x 768 proj gradient of ViT-B runs 9 tiles x 28 splits: 28 slabs to write and sum, an epilogue per
// 28 reduction tiles -- where the pair (36 tiles x 7 splits: the shape of the fc1 gradient) pays once: qkv + proj 160.8 + 71.7 us apart, ~197 as a
// pair.  Falls back to two launches when the pair does not fit the kernel (K not on 256-wide tiles, N1 not a multiple of 256, four-wave kernel off).
// The summation order over the rows differs from the single launches' (another split count), as between any two batch sizes.
int launch_gemm_tn_pair(const uint16_t* P1, const uint16_t* Q1, float* out1, float* bias_out1, float* bias_out1b, int N1, const uint16_t* P2,
                        const uint16_t* Q2, float* out2, int N2, int accumulate, void* ws, size_t ws_bytes, int64_t Mr, int K, hipStream_t st) {
  if (const int rc = sk_check_pending_error()) return rc;
  if (!(Mr > 0 && N1 > 0 && N2 > 0 && K > 0)) { set_error("gemm_tn_pair: empty problem"); return TAD_EINVAL; }
  if (N1 % 8 || N2 % 8 || K % 8) { set_error("gemm_tn_pair: N1=%d, N2=%d and K=%d must be multiples of 8", N1, N2, K); return TAD_EINVAL; }
  const int N = N1 + N2;
  int splits = 0, rows_per_split = 0;
  const int tiles = tn_plan(Mr, N, K, &splits, &rows_per_split);
  const int tiles_k = (K + 255) / 256;
  const bool fits = tn_pair && tn_variant(K) == 1 && knobs::tn_w4 && !knobs::tn_pdeep && !gemm_debug && N1 % 256 == 0 && rows_per_split >= 2 * BK &&
                    Mr * (int64_t)N * 2 < (1ll << 32) && Mr * (int64_t)K * 2 < (1ll << 32) &&
                    ws_bytes >= (size_t)splits * ((size_t)N * K + (size_t)tiles_k * N) * sizeof(float);
  if (!fits) {
    const int rc = launch_gemm_tn(P1, Q1, out1, bias_out1, bias_out1b, accumulate, ws, ws_bytes, Mr, N1, K, st);
    if (rc) return rc;
    return launch_gemm_tn(P2, Q2, out2, nullptr, nullptr, accumulate, ws, ws_bytes, Mr, N2, K, st);
  }
  GemmTN p{};
  p.P = P1; p.Q = Q1; p.P2 = P2; p.Q2 = Q2; p.N1 = N1;
  p.slab = (float*)ws; p.Mr = (int)Mr; p.N = N; p.K = K; p.rows_per_split = rows_per_split;
  p.bias_slab = bias_out1 ? p.slab + (size_t)splits * N * K : nullptr;
  int rc = launch_gemm_tn_w4(p, tiles * splits, st);
  if (rc) return rc;
  // slabs [splits][N][K]: rows [0, N1) -> out1, rows [N1, N) -> out2, both in one launch (with the bias column sums of the first problem)
  const int64_t n1 = (int64_t)N1 * K, n = (int64_t)N * K;
  if (!bias_out1) return launch_reduce_dw_pair(p.slab, out1, out2, n1, splits, n, accumulate, nullptr, 0, 0, nullptr, nullptr, 0, 0, st);
  if (bias_out1b) return launch_reduce_dw_pair(p.slab, out1, out2, n1, splits, n, accumulate, p.bias_slab, splits * tiles_k, N, bias_out1, bias_out1b, N1 / 3, 2 * (N1 / 3), st);
  return launch_reduce_dw_pair(p.slab, out1, out2, n1, splits, n, accumulate, p.bias_slab, splits * tiles_k, N, bias_out1, nullptr, N1, 0, st);
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" {

int tad_linear_fwd(const uint16_t* x, const uint16_t* w, const float* bias, void* y, int y_dtype, int epilogue, uint16_t* preact,
                   const float* residual, const float* gamma, const float* rowscale, int rows_per_scale, void* ws, size_t ws_bytes,
                   int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(x && w && y, "linear_fwd: null pointer");
  TAD_REQUIRE(y_dtype == TAD_F32 || y_dtype == TAD_OP16, "linear_fwd: bad y_dtype %d", y_dtype);
  TAD_REQUIRE(epilogue >= TAD_EPI_BIAS && epilogue <= TAD_EPI_BIAS_RESIDUAL, "linear_fwd: bad epilogue %d", epilogue);
  TAD_REQUIRE(!rowscale || rows_per_scale > 0, "linear_fwd: rows_per_scale must be positive");
  TAD_REQUIRE(M > 0 && M < (1ll << 31), "linear_fwd: bad M");
  GemmNT p{};
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.c_bf16 = (y_dtype == TAD_OP16);
  p.M = (int)M; p.N = N; p.K = K;
  p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
  if (epilogue == TAD_EPI_BIAS_GELU) { p.epi = EPI_GELU; p.preact = preact; }
  else if (epilogue == TAD_EPI_BIAS_RESIDUAL) { p.epi = EPI_RESIDUAL; p.residual = residual; p.gamma = gamma; p.rowscale = rowscale; }
  else p.epi = EPI_PLAIN;
  return launch_gemm_nt(p, (hipStream_t)stream, ws, ws ? ws_bytes : 0);
}

#ifndef TAD_OPND_F16
size_t tad_linear_workspace_bytes(int64_t M, int N, int K) {
  // What launch_gemm_nt's split plan hands to nt_splitk_plan for this shape (same helpers: nt_max_rows, nt_main_rows, nt_splitk_shares; the
  // epilogue kind is not known here: assumed eligible, and both element sizes of the row-range rule are covered), or 0 when no tail of this
  // shape splits along K -- every Linear of ViT-B under the default knobs (64 MB used to stay resident per stream for nothing)
  if (M < 4096 || N < 128 || K <= 0 || K % BK || !nt_split || !nt_splitk) return 0;
  size_t need = 0;
  for (int64_t esz = 2; esz <= 4; esz += 2) {
    const int64_t max_rows = nt_max_rows(N, K, esz);
    if (max_rows <= 0) continue;
    for (int64_t r0 = 0; r0 < M; r0 += max_rows) {  // the row ranges launch_gemm_nt cuts a taller problem into (two distinct sizes at most)
      const int64_t rows = (M - r0) < max_rows ? (M - r0) : max_rows;
      if (r0 > 0 && rows == max_rows) continue;
      int64_t tiles = 0;
      if (rows < 2048 || !nt_main_rows(rows, N, &tiles)) continue;
      const int s = nt_splitk_shares(tiles, K / BK);
      if (s && nt_splitk_ws_bytes(tiles, s) > need) need = nt_splitk_ws_bytes(tiles, s);
    }
  }
  return need;
}
#endif

int tad_linear_fwd_qkv(const uint16_t* x, const uint16_t* w, const float* q_bias, const float* v_bias, void* y, int y_dtype, float q_prescale,
                       int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(x && w && y, "linear_fwd_qkv: null pointer");
  TAD_REQUIRE((q_bias == nullptr) == (v_bias == nullptr), "linear_fwd_qkv: q_bias and v_bias come together");
  TAD_REQUIRE(y_dtype == TAD_F32 || y_dtype == TAD_OP16, "linear_fwd_qkv: bad y_dtype %d", y_dtype);
  TAD_REQUIRE(N > 0 && N % 12 == 0, "linear_fwd_qkv: N=%d must be 3 x a multiple of 4", N);
  TAD_REQUIRE(M > 0 && M < (1ll << 31), "linear_fwd_qkv: bad M");
  TAD_REQUIRE(q_prescale > 0.f, "linear_fwd_qkv: q_prescale must be positive (1 = plain)");
  TAD_REQUIRE(q_prescale == 1.f || N % 24 == 0, "linear_fwd_qkv: q_prescale needs N/3 = %d to be a multiple of 8", N / 3);
  GemmNT p{};
  p.A = x; p.B = w; p.C = y; p.c_bf16 = (y_dtype == TAD_OP16);
  p.bias = q_bias; p.bias2 = v_bias; p.bias_seg = q_bias ? N / 3 : 0;
  if (q_prescale != 1.f) { p.colscale = q_prescale; p.colscale_cols = N / 3; }
  p.M = (int)M; p.N = N; p.K = K;
  p.rows_per_scale = 1;
  p.epi = EPI_PLAIN;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

int tad_linear_bwd_weight_qkv(const uint16_t* dy, const uint16_t* x, float* dW, float* dq_bias, float* dv_bias, int accumulate, void* ws,
                              size_t ws_bytes, int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(dy && x && dW && ws && dq_bias && dv_bias, "linear_bwd_weight_qkv: null pointer");
  TAD_REQUIRE(N > 0 && N % 12 == 0, "linear_bwd_weight_qkv: N=%d must be 3 x a multiple of 4", N);
  return launch_gemm_tn(dy, x, dW, dq_bias, dv_bias, accumulate, ws, ws_bytes, M, N, K, (hipStream_t)stream);
}

int tad_linear_bwd_weight_pair(const uint16_t* dy1, const uint16_t* x1, float* dW1, float* db1, float* db1b, int N1, const uint16_t* dy2,
                               const uint16_t* x2, float* dW2, int N2, int accumulate, void* ws, size_t ws_bytes, int64_t M, int K, tad_stream_t stream) {
  TAD_REQUIRE(dy1 && x1 && dW1 && dy2 && x2 && dW2 && ws, "linear_bwd_weight_pair: null pointer");
  TAD_REQUIRE(!db1b || db1, "linear_bwd_weight_pair: db1b (the v_bias third) comes with db1 (the q_bias third)");
  TAD_REQUIRE(!db1b || (N1 > 0 && N1 % 12 == 0), "linear_bwd_weight_pair: N1=%d must be 3 x a multiple of 4 for the split bias sums", N1);
  // the slab reduction reads and writes float4s through every output (dW2 is usually a view into the flat gradient buffer)
  TAD_REQUIRE(((uintptr_t)dW1 | (uintptr_t)dW2 | (uintptr_t)db1 | (uintptr_t)db1b) % 16 == 0, "linear_bwd_weight_pair: dW1, dW2, db1 and db1b must be 16-byte aligned");
  return launch_gemm_tn_pair(dy1, x1, dW1, db1, db1b, N1, dy2, x2, dW2, N2, accumulate, ws, ws_bytes, M, K, (hipStream_t)stream);
}

#ifndef TAD_OPND_F16  // process-wide knobs / counters: one copy for the library (bf16 pass)
int tad_linear_tuning(const char* key, int value) {
  TAD_REQUIRE(key, "linear_tuning: null key");
  const std::string k(key);
  if (k == "splitk_tail") { TAD_REQUIRE(value >= 0 && value <= 2, "linear_tuning: splitk_tail=%d not in {0, 1, 2}", value); nt_splitk = value; return TAD_OK; }
  if (k == "persistent") nt_persist = value != 0;
  else if (k == "direct_epilogue") { TAD_REQUIRE(value >= 0 && value <= 2, "linear_tuning: direct_epilogue=%d not in 0..2", value); nt_direct = value; }
  else if (k == "debug") gemm_debug = value;  // ablation bits (timing experiments; ignored by production builds)
  else if (k == "group_m") { TAD_REQUIRE(value >= 0 && value <= 1024, "linear_tuning: group_m=%d out of range", value); nt_group_m_knob = value; }
  else if (k == "variant") { TAD_REQUIRE(value >= 0 && value <= 9 && value != 6, "linear_tuning: variant=%d not one of 0..5, 7, 8, 9", value); nt_variant = value; }
  else if (k == "splitk_defer") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: splitk_defer=%d not in {0, 1}", value); nt_sk_defer = value; }
  else if (k == "tn_pair") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: tn_pair=%d not in {0, 1}", value); tn_pair = value; }
  else if (k == "short_k") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: short_k=%d not in {0, 1}", value); nt_short_k = value; }
  else if (k == "tail_192") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: tail_192=%d not in {0, 1}", value); nt_tail_192 = value; }
  else if (k == "w4_epilogues") { TAD_REQUIRE(value >= 0 && value < 16, "linear_tuning: w4_epilogues=%d not a mask of bits 1..3", value); nt_w4_epilogues = value; }
  else if (k == "w4_plain") { TAD_REQUIRE(value >= 0, "linear_tuning: w4_plain=%d must be >= 0", value); nt_w4_plain = value; }
  else if (k == "tn_w4") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: tn_w4=%d not in {0, 1}", value); tn_w4 = value; }
  else if (k == "tn_pdeep") { TAD_REQUIRE(value == 0 || value == 1, "linear_tuning: tn_pdeep=%d not in {0, 1}", value); tn_pdeep = value; }
  else if (k == "split_tail") { TAD_REQUIRE(value >= 0 && value <= 2, "linear_tuning: split_tail=%d not in 0..2", value); nt_split = value; }
  else { set_error("linear_tuning: unknown key '%s'", key); return TAD_EINVAL; }
  return TAD_OK;
}

int tad_linear_tuning_get(const char* key, int* value) {
  TAD_REQUIRE(key && value, "linear_tuning_get: null pointer");
  const std::string k(key);
  const struct { const char* name; const int* v; } table[] = {
      {"splitk_tail", &nt_splitk}, {"persistent", &nt_persist}, {"direct_epilogue", &nt_direct}, {"debug", &gemm_debug}, {"group_m", &nt_group_m_knob},
      {"variant", &nt_variant}, {"splitk_defer", &nt_sk_defer}, {"tn_pair", &tn_pair}, {"short_k", &nt_short_k}, {"tail_192", &nt_tail_192},
      {"w4_epilogues", &nt_w4_epilogues}, {"w4_plain", &nt_w4_plain}, {"tn_w4", &tn_w4}, {"tn_pdeep", &tn_pdeep}, {"split_tail", &nt_split}};
  for (const auto& e : table)
    if (k == e.name) { *value = *e.v; return TAD_OK; }
  set_error("linear_tuning_get: unknown key '%s'", key);
  return TAD_EINVAL;
}

long long tad_linear_kernel_launches(void) { return nt_launches; }

int tad_linear_debug_stamps(void* buf) {
#ifndef TAD_GEMM_ABLATION
  if (buf) { set_error("linear_debug_stamps: timeline stamps need an ablation build (TAD_BUILD_ABLATION=1 python -m simple_tad_amd.build --force)"); return TAD_EINVAL; }
#endif
  nt_stamps = (unsigned long long*)buf;
  return TAD_OK;
}
#endif

int tad_linear_bwd_input(const uint16_t* dy, const uint16_t* wT, void* dx, int dx_dtype, const uint16_t* gelu_preact, void* ws, size_t ws_bytes,
                         int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(dy && wT && dx, "linear_bwd_input: null pointer");
  TAD_REQUIRE(dx_dtype == TAD_F32 || dx_dtype == TAD_OP16, "linear_bwd_input: bad dx_dtype %d", dx_dtype);
  TAD_REQUIRE(M > 0 && M < (1ll << 31), "linear_bwd_input: bad M");
  GemmNT p{};
  p.A = dy; p.B = wT; p.C = dx; p.c_bf16 = (dx_dtype == TAD_OP16);
  p.M = (int)M; p.N = K; p.K = N;  // dx[M,K] = dy[M,N] * (wT[K,N])^T
  p.rows_per_scale = 1;
  p.epi = gelu_preact ? EPI_DGELU : EPI_PLAIN;
  p.dgelu_h = gelu_preact;
  return launch_gemm_nt(p, (hipStream_t)stream, ws, ws ? ws_bytes : 0);
}

#ifndef TAD_OPND_F16
size_t tad_linear_bwd_weight_workspace_bytes(int64_t M, int N, int K) {
  return gemm_tn_workspace_bytes(M, N, K);
}
#endif

int tad_linear_bwd_weight(const uint16_t* dy, const uint16_t* x, float* dW, float* db, int accumulate, void* ws, size_t ws_bytes,
                          int64_t M, int N, int K, tad_stream_t stream) {
  TAD_REQUIRE(dy && x && dW && ws, "linear_bwd_weight: null pointer");
  return launch_gemm_tn(dy, x, dW, db, nullptr, accumulate, ws, ws_bytes, M, N, K, (hipStream_t)stream);
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
  p.M = B * ntok; p.N = D; p.K = tad_patch_embed_ldk(C, tubelet, patch);  // cols / w_bf16 row stride (= K unless the patch is /14-like)
  p.rows_per_scale = 1;
  p.epi = EPI_RESIDUAL;
  p.residual = pos;
  p.res_mod = ntok;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

int tad_patch_embed_gemm(const uint16_t* cols, const uint16_t* w_bf16, const float* bias, const float* pos, float* out, int64_t M, int ntok,
                         int D, int K, tad_stream_t stream) {
  TAD_REQUIRE(cols && w_bf16 && out, "patch_embed_gemm: null pointer");
  TAD_REQUIRE(M > 0 && M < (1ll << 31) && ntok > 0 && M % ntok == 0, "patch_embed_gemm: M=%lld must be a positive multiple of ntok=%d",
              (long long)M, ntok);
  GemmNT p{};
  p.A = cols; p.B = w_bf16; p.C = out; p.bias = bias; p.c_bf16 = 0;
  p.M = (int)M; p.N = D; p.K = K;
  p.rows_per_scale = 1;
  p.epi = EPI_RESIDUAL;
  p.residual = pos;
  p.res_mod = ntok;
  return launch_gemm_nt(p, (hipStream_t)stream);
}

#ifndef TAD_OPND_F16
size_t tad_patch_embed_bwd_workspace_bytes(int64_t M, int D, int K) { return tad_linear_bwd_weight_workspace_bytes(M, D, K); }
#endif

int tad_patch_embed_bwd(const uint16_t* dy_bf16, const uint16_t* cols, float* dW, float* db, void* ws, size_t ws_bytes, int64_t M, int D,
                        int K, tad_stream_t stream) {
  return tad_linear_bwd_weight(dy_bf16, cols, dW, db, 0, ws, ws_bytes, M, D, K, stream);
}

}  // extern "C"
#endif  // TAD_GEMM_W4_TU
