// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libtad_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tad_mi355x.h"

namespace tad {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int WAVE = 64;

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define TAD_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      tad::set_error(__VA_ARGS__);            \
      return TAD_EINVAL;                      \
    }                                         \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
// The epilogues of the fc1 / fc2-dX GEMMs are VALU-bound (128 outputs per lane; the A-S form above costs ~76 issue cycles per
// element = as long as half the K loop); these forms cost ~38.  Coefficients: tools/fit_gelu_poly.py (Chebyshev fit of
// (f(x) - 0.5)/x in x^2 on |x| <= XMAX, Horner in t = 2x^2/XMAX^2 - 1).  Errors (f32 evaluation): |Phi| <= 3.9e-7, gelu <= 1.9e-6,
// gelu' <= 7e-6 absolute -- three orders below the bf16 rounding of the values these epilogues read and write.  The precise mode
// keeps the A-S form.
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr float PHI_XMAX = 5.0f;
constexpr float PHI_C[13] = {1.413638185e-01f, -7.029590887e-02f, 5.151792974e-02f, -4.045128240e-02f, 3.147675865e-02f, -2.321312828e-02f, 1.623608981e-02f, -1.130712491e-02f, 6.766527505e-03f, -2.526916729e-03f, 1.374596151e-03f, -1.676730979e-03f, 7.353763888e-04f};
constexpr float DGELU_XMAX = 5.5f;
constexpr float DGELU_C[14] = {1.287606817e-01f, -6.574511122e-02f, 5.352302286e-02f, -5.354277321e-02f, 6.254520933e-02f, -7.080669140e-02f, 6.123300602e-02f, -6.376653697e-02f, 1.008177666e-01f, -7.514079910e-02f, -9.145315790e-03f, 6.300958162e-04f, 4.666087374e-02f, -2.511548108e-02f};

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
template <int N>
__device__ __forceinline__ f32x2 half_plus_x_poly2(f32x2 x, const float (&c)[N], float xmax) {
  const f32x2 xc = f32x2{__builtin_amdgcn_fmed3f(x[0], -xmax, xmax), __builtin_amdgcn_fmed3f(x[1], -xmax, xmax)};
  const f32x2 t = __builtin_elementwise_fma(xc * xc, splat2(2.0f / (xmax * xmax)), splat2(-1.0f));
  f32x2 r = splat2(c[N - 1]);
#pragma unroll
  for (int i = N - 2; i >= 0; --i) r = __builtin_elementwise_fma(r, t, splat2(c[i]));
  return __builtin_elementwise_fma(xc, r, splat2(0.5f));
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) { return x * half_plus_x_poly2(x, PHI_C, PHI_XMAX); }
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) { return half_plus_x_poly2(x, DGELU_C, DGELU_XMAX); }

// Bijective XCD-aware block remap (8 XCDs, blocks dealt round-robin): consecutive logical ids land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace tad
