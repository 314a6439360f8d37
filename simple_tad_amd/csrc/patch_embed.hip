// PatchEmbed as an IMPLICIT GEMM (modeling_finetune.py:169-191, 312-313): out[m, d] = sum_k A[m, k] W[d, k] + bias[d] + pos[m % ntok, d]
// with A[m, k] = x[b, c, tub t' + kt, 16 h' + kh, 16 w' + kw] read straight from the f32 clip -- no patch matrix in memory.
//
// When it is used: forwards that keep nothing for a backward pass (eval / no_grad / inference: ops.PatchEmbedFn).  The training step keeps the
// explicit form (tad_patch_embed_fwd: im2col + gemm_nt): its weight gradient reads the 16-bit patch matrix again, so the matrix has to exist anyway,
// and an implicit A operand costs f32 reads (twice the bytes of the 16-bit matrix) once per column tile of the output.
//
// Structure (tile 128 tokens x 128 outputs, K-tile 64 = four image rows kh of one (channel, frame) pair x the patch's 16 columns kw; 256 threads =
// 2 x 2 waves of 64 x 64; two workgroups per CU): the x operand is REGISTER-staged -- each thread loads eight 16-byte pieces of the next K-tile
// (consecutive lanes walk the 64-byte patch rows of horizontally adjacent tokens: 896 contiguous bytes per image row of a 14-token row), rounds
// them to the 16-bit operand format exactly as tad_im2col_tubelets does, and drops them into the LDS image gemm_nt's fragment reads expect
// (128-byte rows, XOR swizzle on the 16-byte chunk: sw_nt); the weight operand arrives by LDS-DMA as in gemm_nt.  Two LDS slots, one barrier
// per K-tile.  Same matrix instruction, same k order inside and across K-tiles, bias as the initial accumulator and the position row added
// last as in gemm_nt's "+pos_embed" epilogue: results are BIT-IDENTICAL to tad_patch_embed_fwd (tests/test_kernels_gpu.py).
// Patch size 16 only (the reference's only one: modeling_finetune.py:338-398); other sizes take the explicit route.
#include "common.h"

TAD_NAMESPACE_BEGIN

__device__ __forceinline__ int pe_sw(int row) { return ((row >> 1) & 7) ^ (((row >> 4) & 3) << 1); }  // = gemm.hip's sw_nt

struct PatchEmbedImplicit {
  const float* x;       // [B, C, T, H, W]
  const uint16_t* w;    // [D, K] 16-bit
  const float* bias;    // [D] or null
  const float* pos;     // [ntok, D] or null
  float* out;           // [M, D]
  int M, D, K;          // K = C * tub * 256
  int C, T, H, W, tub;
  int Hp, Wp, ntok;     // tokens per column / row / clip
};

__global__ __launch_bounds__(256, 2) void patch_embed_implicit_kernel(const PatchEmbedImplicit p) {
  constexpr int BM = 128, BN = 128, BKT = 64, ROWB = 128, A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES, NW = 4;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.D + BN - 1) / BN;
  // consecutive logical ids = the column tiles of one row tile, kept on one XCD (they re-read the same x pieces through its L2)
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;
  const int nk = p.K / BKT;

  // ---- x operand: piece q = it * 256 + tid of a K-tile: image row khl = it >> 1 of the K-tile's four, token (it & 1) * 64 + tid / 4, 16-byte piece tid & 3
  const int64_t plane = (int64_t)p.H * p.W;
  uint32_t tok_off[2];  // byte offset of the token's patch origin (b, c = 0, t = tub t', y = 16 h', x = 16 w'), or out of range
  const uint32_t x_bytes = (uint32_t)((int64_t)(p.M / p.ntok) * p.C * p.T * plane * 4);
  const auto x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)x_bytes, 0x00020000);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + j * 64 + (tid >> 2);
    if (m < p.M) {
      const int b = m / p.ntok, n = m - b * p.ntok;
      const int tp = n / (p.Hp * p.Wp), r = n - tp * (p.Hp * p.Wp);
      const int hp = r / p.Wp, wp = r - hp * p.Wp;
      tok_off[j] = (uint32_t)((((int64_t)b * p.C * p.T + (int64_t)tp * p.tub) * plane + (int64_t)hp * 16 * p.W + wp * 16) * 4) + (uint32_t)((tid & 3) * 16);
    } else {
      tok_off[j] = 0x80000000u;  // rows past M: the buffer's bounds check returns zeros
    }
  }
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
  u32x4 xa[8];
  // K-tile kt = (c, frame kt_, rows 4 q4 .. + 3): k = ((c tub + kt_) 16 + kh) 16 + kw
  auto load_x = [&](int kt) {
    const int ck = kt >> 2, q4 = kt & 3;                                   // ck = c * tub + kt_
    const int c = ck / p.tub, kf = ck - c * p.tub;
    const uint32_t base = (uint32_t)((((int64_t)c * p.T + kf) * plane + (int64_t)(4 * q4) * p.W) * 4);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const uint32_t o = tok_off[it & 1];
      xa[it] = __builtin_amdgcn_raw_buffer_load_b128(x_rs, o == 0x80000000u ? o : o + base + (uint32_t)((it >> 1) * p.W * 4), 0, 0);
    }
  };
  auto store_x = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = (it & 1) * 64 + (tid >> 2), c8 = (it >> 1) * 2 + ((tid & 3) >> 1);
      const u32x2 v = {pack_op16x2(__uint_as_float(xa[it][0]), __uint_as_float(xa[it][1])), pack_op16x2(__uint_as_float(xa[it][2]), __uint_as_float(xa[it][3]))};
      *reinterpret_cast<u32x2*>(lds + buf * STAGE + row * ROWB + ((c8 ^ pe_sw(row)) << 4) + (tid & 1) * 8) = v;
    }
  };
  // ---- weight operand by LDS-DMA: a 1-KiB piece = 8 rows; wave w moves pieces w, w + 4, w + 8, w + 12
  const auto w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, (int)((int64_t)p.D * p.K * 2), 0x00020000);
  uint32_t w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    w_off[i] = (uint32_t)(n0 + row) * (uint32_t)(p.K * 2) + (uint32_t)(((lane & 7) ^ pe_sw(row)) << 4);  // rows >= D land past the descriptor: zeros
  }
  auto dma_w = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, LDS_PTR(lds + buf * STAGE + A_BYTES + (i * NW + wave) * 1024), 16, w_off[i], (uint32_t)kt * ROWB, 0, 0);
  };
  // ---- fragments (gemm_nt's addressing): lane (c = lane & 15, kq = lane >> 4)
  const int c = lane & 15, kq = lane >> 4;
  uint32_t a_rd[4], b_rd[4];
  int a_sw[4], b_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + c, rb = wn * 64 + i * 16 + c;
    a_rd[i] = ra * ROWB; a_sw[i] = pe_sw(ra);
    b_rd[i] = A_BYTES + rb * ROWB; b_sw[i] = pe_sw(rb);
  }
  // accumulators start from the bias (gemm_nt): lane holds out[m = 16 i + c][n = 16 j + 4 kq .. + 3]
  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nc = n0 + wn * 64 + 16 * j + 4 * kq;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nc < p.D) {
      const float4 t = *reinterpret_cast<const float4*>(p.bias + nc);
      b4 = f32x4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][j] = b4;
  }

  load_x(0);
  dma_w(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    store_x(cur);  // (slot `cur` was last read two K-tiles ago: every wave has passed the barrier of K-tile kt - 1 since)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this K-tile's weight pieces have landed, the x pieces are written
    __syncthreads();
    if (kt + 1 < nk) {  // behind the barrier: nobody reads slot cur ^ 1 any more
      load_x(kt + 1);
      dma_w(cur ^ 1, kt + 1);
    }
    const char* s = lds + cur * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      op16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const op16x8*>(s + a_rd[i] + (((4 * ks + kq) ^ a_sw[i]) << 4));
        bf[i] = *reinterpret_cast<const op16x8*>(s + b_rd[i] + (((4 * ks + kq) ^ b_sw[i]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = TAD_MFMA_16x16x32(bf[j], af[i], acc[i][j]);
    }
  }
  // ---- epilogue: + pos[m % ntok], 16 bytes per lane (four lanes cover 64 contiguous bytes of a row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + 16 * i + c;
    if (m >= p.M) continue;
    const float* prow = p.pos ? p.pos + (int64_t)(m % p.ntok) * p.D : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + 16 * j + 4 * kq;
      if (n >= p.D) continue;
      f32x4 v = acc[i][j];
      if (prow) {
        const float4 t = *reinterpret_cast<const float4*>(prow + n);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
      *reinterpret_cast<float4*>(p.out + (int64_t)m * p.D + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" int tad_patch_embed_fwd_implicit(const float* x, const uint16_t* w_bf16, const float* bias, const float* pos, float* out, int B, int C,
                                            int T, int H, int W, int tubelet, int patch, int D, tad_stream_t stream) {
  TAD_REQUIRE(x && w_bf16 && out, "patch_embed_fwd_implicit: null pointer");
  TAD_REQUIRE(patch == 16, "patch_embed_fwd_implicit: patch size %d (the implicit kernel is written for 16; use tad_patch_embed_fwd)", patch);
  TAD_REQUIRE(B > 0 && C > 0 && tubelet > 0 && T % tubelet == 0 && H % 16 == 0 && W % 16 == 0, "patch_embed_fwd_implicit: T/H/W must be multiples of tubelet/16 (got T=%d H=%d W=%d tub=%d)", T, H, W, tubelet);
  TAD_REQUIRE(D > 0 && D % 4 == 0, "patch_embed_fwd_implicit: D=%d must be a multiple of 4", D);
  TAD_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)w_bf16) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (((uintptr_t)bias) & 15) == 0 && (((uintptr_t)pos) & 15) == 0,
              "patch_embed_fwd_implicit: buffers must be 16-byte aligned");
  PatchEmbedImplicit p{};
  p.x = x; p.w = w_bf16; p.bias = bias; p.pos = pos; p.out = out;
  p.C = C; p.T = T; p.H = H; p.W = W; p.tub = tubelet;
  p.Hp = H / 16; p.Wp = W / 16;
  p.ntok = (T / tubelet) * p.Hp * p.Wp;
  const int64_t M = (int64_t)B * p.ntok;
  p.K = C * tubelet * 256;
  p.D = D;
  TAD_REQUIRE(M < (1ll << 31) && (int64_t)B * C * T * H * W * 4 < (1ll << 31) && (int64_t)D * p.K * 2 < (1ll << 31) && M * D * 4 < (1ll << 40),
              "patch_embed_fwd_implicit: clip of %lld bytes / weight beyond the 2 GiB buffer offsets", (long long)B * C * T * H * W * 4);
  p.M = (int)M;
  const int64_t tiles = ((M + 127) / 128) * ((D + 127) / 128);
  TAD_REQUIRE(tiles < (1ll << 31), "patch_embed_fwd_implicit: grid too large");
  hipLaunchKernelGGL(patch_embed_implicit_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("patch_embed_fwd_implicit");
}
