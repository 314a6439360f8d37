// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libtad_mi355x.so.
#pragma once
// 16-bit operand format of this compilation pass.  Every source that touches 16-bit GEMM / attention operands is compiled twice
// (simple_tad_amd/build.py): once for bfloat16 (the default: entry points tad_*) and once with -DTAD_OPND_F16 for IEEE half
// (entry points tad_*_f16, see opnd_f16_names.h).  Same kernels, same schedules, same MFMA rate; only the conversion instructions
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, shift / v_cvt_f32_f16) and the MFMA opcode differ.  Device code lives in an inline
// namespace per pass so that equally named template instantiations of the two passes never meet at link time.
#ifdef TAD_OPND_F16
#include "opnd_f16_names.h"
#define TAD_NS op_f16
#else
#define TAD_NS op_bf16
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
#include "../../include/tad_mi355x.h"

#define TAD_NAMESPACE_BEGIN namespace tad { inline namespace TAD_NS {
#define TAD_NAMESPACE_END }}

namespace tad {
// shared by both passes (defined once, capi.hip)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
}  // namespace tad

TAD_NAMESPACE_BEGIN

#ifdef TAD_OPND_F16
typedef _Float16 op16_t;
#define TAD_OP16 TAD_F16
#define TAD_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define TAD_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 op16_t;
#define TAD_OP16 TAD_BF16
#define TAD_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define TAD_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

typedef __attribute__((ext_vector_type(8))) op16_t op16x8;
typedef __attribute__((ext_vector_type(4))) op16_t op16x4;
typedef __attribute__((ext_vector_type(2))) op16_t op16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int WAVE = 64;

#define TAD_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      tad::set_error(__VA_ARGS__);            \
      return TAD_EINVAL;                      \
    }                                         \
  } while (0)

// 16-bit operand <-> f32.  bf16: the upper half of the f32 pattern; f16: v_cvt_f32_f16.  Round-to-nearest-even on the way down;
// the plain casts lower to v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950 and keep a NaN a NaN.
__device__ __forceinline__ float op16_to_f32(uint16_t v) {
#ifdef TAD_OPND_F16
  return (float)__builtin_bit_cast(_Float16, v);
#else
  return __uint_as_float(((uint32_t)v) << 16);
#endif
}
// low / high element of a packed pair
__device__ __forceinline__ float op16_lo_f32(uint32_t w) {
#ifdef TAD_OPND_F16
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
#else
  return __uint_as_float(w << 16);
#endif
}
__device__ __forceinline__ float op16_hi_f32(uint32_t w) {
#ifdef TAD_OPND_F16
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
#else
  return __uint_as_float(w & 0xffff0000u);
#endif
}
__device__ __forceinline__ uint16_t f32_to_op16(float f) {
  op16_t b = (op16_t)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi) {
  op16x2 v;
  v[0] = (op16_t)lo;
  v[1] = (op16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 64 lanes without LDS traffic and without waits: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after them every lane of a row holds the row's sum), then the two row
// exchanges of gfx950 (v_permlane16_swap, v_permlane32_swap).  `__shfl_xor` lowers to ds_bpermute_b32 + s_waitcnt lgkmcnt per step
// (12 LDS round trips for the two sums of a LayerNorm-backward row).  The result is the same in every lane; the association of the
// additions differs from wave_sum's butterfly, so the two are not bitwise interchangeable.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm 1,0,3,2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm 2,3,0,1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- transposed LDS reads (ds_read_b64_tr_b16) as inline asm
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the BUILTIN form of this read whenever an LDS-DMA (buffer_load ... lds) is
// in flight -- it cannot tell that the read does not touch the DMA's destination -- so the prefetch of the next tile was drained
// right after it had been issued and never overlapped the matrix work (plain ds_read_b128 reads do not get that wait).  The asm form
// is invisible to the compiler's waitcnt pass: the DMA stays in flight, but the result registers must be waited for BY HAND with
// lds_wait<N>(regs...), which emits s_waitcnt lgkmcnt(N) and names the registers as in/out operands so that no consumer (and no
// register copy) can be scheduled ahead of it.  N = number of LDS instructions issued after the ones being waited for (LDS operations
// complete in order; reads the compiler issues by itself in between only make the wait more conservative, never wrong; the counter
// has 4 bits, so N saturates at 15).
// EXEC must be all ones (the gather crosses lanes).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)LDS_PTR(p); }
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16_b64(uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <typename T, int OFF>
__device__ __forceinline__ T lds_read_b128(uint32_t addr) {  // plain 16-byte read (T = op16x8, f32x4 ...), same contract as lds_tr16_b64
  static_assert(OFF >= 0 && OFF < 65536 && sizeof(T) == 16, "ds offset field is 16 bits; 16-byte result");
  T r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
#define TAD_LGKM(N) "n"((N) > 15 ? 15 : (N))
template <int N, typename A>
__device__ __forceinline__ void lds_wait(A& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : TAD_LGKM(N));
}
template <int N, typename A, typename B>
__device__ __forceinline__ void lds_wait(A& a, B& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : TAD_LGKM(N));
}
template <int N, typename A, typename B, typename C, typename D>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c, D& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : TAD_LGKM(N));
}
template <int N, typename A, typename B, typename C>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : TAD_LGKM(N));
}
template <int N, typename A, typename B, typename C, typename D, typename E>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c, D& d, E& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : TAD_LGKM(N));
}
template <int N, typename A, typename B, typename C, typename D, typename E, typename F>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c, D& d, E& e, F& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : TAD_LGKM(N));
}
template <int N, typename A, typename B, typename C, typename D, typename E, typename F, typename G, typename H>
__device__ __forceinline__ void lds_wait(A& a, B& b, C& c, D& d, E& e, F& f, G& g, H& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : TAD_LGKM(N));
}
__device__ __forceinline__ op16x8 join_tr(const s16x4& lo, const s16x4& hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(op16x8, v);
}
// A 32 x 64 tile of 16-bit values held the way two 32 x 32 accumulator tiles leave it -- lane = (row lane & 31, half h5 = lane >> 5),
// pk[dt][r4] = the four columns dt*32 + 8*r4 + 4*h5 .. +3 of that row -- stored as WHOLE ROWS: through a per-wave LDS patch (32 rows x
// 144 B: 128 + 16, so the 16-byte row reads stay aligned and the 8-byte writes of consecutive rows spread over the banks), then 16 bytes per
// lane, eight lanes per 128-byte row.  Stored directly, every instruction writes 8 bytes to each of 32 rows (16 instructions per tile
// against 4 here); measured on the attention backward: the scattered form cost up to 3.4 % of the kernel pair.  `dst` points at column 0 of the
// tile's first row, `row_stride` in elements, rows >= rows_valid are not stored.  One wave, no barrier: LDS operations of a wave
// complete in order; consecutive calls may reuse the patch.
constexpr int ROW_PATCH_BYTES = 32 * 144;
__device__ __forceinline__ void store_rows_via_lds(char* patch, const uint2 (&pk)[2][4], uint16_t* dst, int64_t row_stride, int rows_valid,
                                                   int lane) {
  const int rl = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) *reinterpret_cast<uint2*>(patch + rl * 144 + (dt * 32 + 8 * r4 + 4 * h5) * 2) = pk[dt][r4];
  // The rows are read back by OTHER lanes through a differently typed pointer: make the order explicit instead of relying on the
  // compiler keeping differently typed LDS accesses in program order -- a workgroup-scope fence (no instruction beyond
  // the s_waitcnt lgkmcnt(0) the read-back needs anyway; the LDS executes one wave's operations in order).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3), ch = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(patch + row * 144 + ch * 16);
    if (row < rows_valid) *reinterpret_cast<uint4*>(dst + row * row_stride + ch * 8) = v;
  }
  // (a following call reuses the patch: its writes must not pass these reads)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Attention dropout (nn.Dropout on the softmax matrix, modeling_finetune.py:99-101; flash_attention_class.py:59-61 passes dropout_p in
// training): element (b, h, query, key) is kept iff hash(row, key, seed) >= p * 2^32, row = (b H + h) N + query, and kept
// probabilities are scaled by 1 / (1 - p).  A counter-based hash instead of a stored N x N mask: every attention kernel (16-bit and
// f32 operands, forward and the two backward kernels) regenerates the same bits, and so does the oracle (oracle/vit_oracle.py:
// attention_dropout_keep) -- parity by injected mask, as for drop-path.
struct Drop {
  uint32_t thr, seed;
  float inv_keep;
};
__device__ __forceinline__ bool drop_keep(const Drop& d, uint32_t row, uint32_t key) {
  uint32_t x = row * 0x9E3779B1u + key * 0x85EBCA77u + d.seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x >= d.thr;
}
inline bool make_drop(float p, uint32_t seed, Drop* d) {
  if (!(p >= 0.f && p < 1.f)) return false;
  d->thr = (uint32_t)((double)p * 4294967296.0);
  d->seed = seed;
  d->inv_keep = 1.f / (1.f - p);
  return true;
}
// accumulator register r of lane half h holds row (r & 3) + 8 (r >> 2) + 4 h of a 32 x 32 tile
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// The 16 extra head dims of a head_dim-80 tile (dims 64..79), held the way ONE 32 x 32 accumulator tile leaves them with only its rows
// 0..15 meaningful: lane (row lane & 31, half h5) has registers 0..7 = dims (r & 3) + 8 (r >> 2) + 4 h5, i.e. two groups of four
// consecutive dims.  pk[g] = the packed group g (dims 8 g + 4 h5 .. +3); stored as 8-byte pieces (32 bytes per row in all).
__device__ __forceinline__ void store_side16(const uint2 (&pk)[2], uint16_t* dst, int64_t row_stride, int rows_valid, int lane) {
  const int rl = lane & 31, h5 = lane >> 5;
  if (rl < rows_valid) {
#pragma unroll
    for (int g = 0; g < 2; ++g) *reinterpret_cast<uint2*>(dst + rl * row_stride + 8 * g + 4 * h5) = pk[g];
  }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [BEGIN, END)
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (BEGIN < END) {
    f(std::integral_constant<int, BEGIN>{});
    static_for<BEGIN + 1, END>(f);
  }
}

// raw v_exp_f32 (2^x); denormal results flush to zero, which is what the softmax wants
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf-form GELU (nn.GELU() default, modeling_finetune.py:38) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, i.e. f32 rounding level) with one v_rcp and one v_exp: libm's erff costs ~40 VALU instructions
// per element, which made the GELU epilogue as long as the whole K loop of the fc1 GEMM.
__device__ __forceinline__ void erf_parts(float x, float& half_erfc_abs, float& e) {
  // for z = |x|/sqrt(2): returns 0.5*(1 - erf(z)) = 0.5*poly(t)*exp(-z^2) and e = exp(-x^2/2)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  e = fast_exp2(-0.72134752044448170368f * x * x);  // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  half_erfc_abs = 0.5f * poly * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float h, e;
  erf_parts(x, h, e);
  const float cdf = x >= 0.f ? 1.0f - h : h;  // Phi(x) = 0.5*(1 + erf(x/sqrt 2))
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float h, e;
  erf_parts(x, h, e);
  const float cdf = x >= 0.f ? 1.0f - h : h;
  return fmaf(x * 0.39894228040143267794f, e, cdf);  // Phi(x) + x * phi(x)
}

// ---- fast-mode GELU for the GEMM epilogues: no transcendental, packed f32 math (v_pk_fma_f32 does two elements per issue).
// The epilogues of the fc1 / fc2-dX GEMMs are VALU-bound (128 outputs per lane).  Measured on gfx950 (tools/micro/valu_rate.hip):
// a DEPENDENT v_pk_fma_f32 (or v_fma_f32) issues every 9 cycles per wave, independent ones every ~5.4, at one or two waves per
// SIMD alike -- so the Horner recurrences of the 4 element pairs of a row are evaluated side by side (coefficient-major order),
// never one chain after the other.  Coefficients: tools/fit_gelu_poly.py (Chebyshev fit of (f(x) - 0.5)/x in x^2 on |x| <= XMAX,
// Horner in t = 2x^2/XMAX^2 - 1).  Errors (f32 evaluation): |Phi| <= 2.4e-5, gelu <= 2.4e-5 * max(1, |x|), gelu' <= 1.1e-4 absolute --
// 20 to 100 times below the bf16 rounding (2^-9 relative) of the values these epilogues read and write.  The precise mode keeps
// the A-S form above.
typedef __attribute__((ext_vector_type(2))) float f32x2;
#ifdef TAD_OPND_F16
// half pass: the values these epilogues read and write are rounded at 2^-12, so the polynomials are carried four / five degrees
// further (|Phi error| <= 3.9e-7, gelu <= 1.9e-6 * max(1, |x|), |gelu' error| <= 6.0e-6: >= 40x below the half rounding again)
constexpr float PHI_XMAX = 5.0f;
constexpr float PHI_C[13] = {1.413638185e-01f, -7.029590887e-02f, 5.151792974e-02f, -4.045128240e-02f, 3.147675865e-02f, -2.321312828e-02f, 1.623608981e-02f, -1.130712491e-02f, 6.766527505e-03f, -2.526916729e-03f, 1.374596151e-03f, -1.676730979e-03f, 7.353763888e-04f};
constexpr float DGELU_XMAX = 5.0f;
constexpr float DGELU_C[14] = {1.421342312e-01f, -7.511106477e-02f, 6.653327323e-02f, -7.173111262e-02f, 8.085778139e-02f, -8.495648970e-02f, 7.754241580e-02f, -6.598634567e-02f, 5.801962140e-02f, -3.821696020e-02f, 1.123988696e-02f, -7.203916123e-03f, 1.261547586e-02f, -5.735615125e-03f};
#else
constexpr float PHI_XMAX = 4.0f;
constexpr float PHI_C[9] = {1.759501642e-01f, -8.430131655e-02f, 5.591522291e-02f, -3.713346277e-02f, 2.266773498e-02f, -1.154668287e-02f, 5.828486749e-03f, -3.991263335e-03f, 1.605170928e-03f};
constexpr float DGELU_XMAX = 4.5f;
constexpr float DGELU_C[10] = {1.594574418e-01f, -9.003904569e-02f, 8.571837549e-02f, -9.350841452e-02f, 1.078448091e-01f, -9.557466143e-02f, 4.249439465e-02f, -3.356380937e-02f, 5.896066889e-02f, -3.068672777e-02f};
#endif

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
// out[i] = 0.5 + xc[i] * P(t[i]) for W independent element pairs, coefficient-major so that the W recurrences interleave
template <int N, int W>
__device__ __forceinline__ void half_plus_x_poly2(const f32x2 (&x)[W], f32x2 (&out)[W], const float (&c)[N], float xmax) {
  f32x2 xc[W], t[W], r[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    xc[i] = f32x2{__builtin_amdgcn_fmed3f(x[i][0], -xmax, xmax), __builtin_amdgcn_fmed3f(x[i][1], -xmax, xmax)};
    t[i] = __builtin_elementwise_fma(xc[i] * xc[i], splat2(2.0f / (xmax * xmax)), splat2(-1.0f));
    r[i] = splat2(c[N - 1]);
  }
#pragma unroll
  for (int k = N - 2; k >= 0; --k) {
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = __builtin_elementwise_fma(r[i], t[i], splat2(c[k]));
  }
#pragma unroll
  for (int i = 0; i < W; ++i) out[i] = __builtin_elementwise_fma(xc[i], r[i], splat2(0.5f));
}
// v[0..2W) -> gelu(v) in place
template <int W>
__device__ __forceinline__ void gelu_fast_row(float* v) {
  f32x2 x[W], ph[W];
#pragma unroll
  for (int i = 0; i < W; ++i) x[i] = f32x2{v[2 * i], v[2 * i + 1]};
  half_plus_x_poly2(x, ph, PHI_C, PHI_XMAX);
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const f32x2 y = x[i] * ph[i];
    v[2 * i] = y[0];
    v[2 * i + 1] = y[1];
  }
}
// v[0..2W) *= gelu'(h), h given as W packed 16-bit operand pairs
template <int W>
__device__ __forceinline__ void gelu_grad_fast_row(float* v, const uint32_t* hw) {
  f32x2 x[W], g[W];
#pragma unroll
  for (int i = 0; i < W; ++i) x[i] = f32x2{op16_lo_f32(hw[i]), op16_hi_f32(hw[i])};
  half_plus_x_poly2(x, g, DGELU_C, DGELU_XMAX);
#pragma unroll
  for (int i = 0; i < W; ++i) {
    v[2 * i] *= g[i][0];
    v[2 * i + 1] *= g[i][1];
  }
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 xs[1] = {x};
  f32x2 ph[1];
  half_plus_x_poly2(xs, ph, PHI_C, PHI_XMAX);
  return x * ph[0];
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
  const f32x2 xs[1] = {x};
  f32x2 g[1];
  half_plus_x_poly2(xs, g, DGELU_C, DGELU_XMAX);
  return g[0];
}

// Bijective XCD-aware block remap (8 XCDs, blocks dealt round-robin): consecutive logical ids land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}


// global accesses with a compile-time cache policy: NT = non-temporal ("streamed": the line is the first to leave the caches)
template <bool NT>
__device__ __forceinline__ float4 ldg_f4(const float4* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4*>(p)) : *reinterpret_cast<const v4*>(p);
  return make_float4(t[0], t[1], t[2], t[3]);
}
template <bool NT>
__device__ __forceinline__ uint2 ldg_u2(const uint2* p) {
  typedef unsigned v2 __attribute__((ext_vector_type(2)));
  const v2 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2*>(p)) : *reinterpret_cast<const v2*>(p);
  return make_uint2(t[0], t[1]);
}
template <bool NT>
__device__ __forceinline__ void stg_f4(float4* p, const float4& v) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 t = {v.x, v.y, v.z, v.w};
  if (NT) __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
  else *reinterpret_cast<v4*>(p) = t;
}
template <bool NT>
__device__ __forceinline__ void stg_u2(uint2* p, const uint2& v) {
  typedef unsigned v2 __attribute__((ext_vector_type(2)));
  const v2 t = {v.x, v.y};
  if (NT) __builtin_nontemporal_store(t, reinterpret_cast<v2*>(p));
  else *reinterpret_cast<v2*>(p) = t;
}

TAD_NAMESPACE_END
