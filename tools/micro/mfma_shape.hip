// Which bf16 MFMA shape sustains more FLOP/s in an attention-like loop on random data -- v_mfma_f32_32x32x16_bf16 (what the attention
// kernels use) or v_mfma_f32_16x16x32_bf16 (what the GEMMs use)?  MI355X_MICROARCH.md (DVFS give-back, item 7) reports 1.12-1.15x for
// bare loops of the 16x16x32 shape at equal cycles per FLOP (the chip holds a higher clock).  This probe runs the attention forward's
// instruction mix per "tile" at its occupancy (4 waves per SIMD, 256-thread workgroups, <= 128 VGPRs): 8 units of score MFMAs (K operand
// rotating over 8 random fragments, Q fixed), VALU (one v_sub + v_exp per score, v_cvt_pk per pair; VALU = 0 leaves them out), 8 units of
// P V MFMAs whose B operand is the packed P.  One unit = 32768 FLOP... per 64 lanes: 32x32x16 = 1 instruction, 16x16x32 = 2.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int SHAPE, int VALU>
__global__ __launch_bounds__(256, 4) void k(const bf16x8* __restrict__ rnd, float* __restrict__ sink, unsigned long long* clk, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 kf[8], qf[4], vf[2];
  for (int i = 0; i < 8; ++i) kf[i] = rnd[(tid * 16 + i) & 0xfffff];
  for (int i = 0; i < 4; ++i) qf[i] = rnd[(tid * 16 + 8 + i) & 0xfffff];
  for (int i = 0; i < 2; ++i) vf[i] = rnd[(tid * 16 + 12 + i) & 0xfffff];
  float o[32];
  for (int i = 0; i < 32; ++i) o[i] = 0.f;
  float m = 0.25f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    float s[32];
    if (SHAPE == 0) {
      f32x16 a0, a1;
      for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[4 + ks], qf[ks], a1, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) { s[r] = a0[r]; s[16 + r] = a1[r]; }
    } else {
      f32x4 a[8];
      for (int i = 0; i < 8; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[2 * ks + (i & 1)], a[i], 0, 0, 0);
      for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) s[4 * i + r] = a[i][r];
    }
    bf16x8 pf[4];
    if (VALU) {
#pragma unroll
      for (int i = 0; i < 32; ++i) pf[i >> 3][i & 7] = (__bf16)__builtin_amdgcn_exp2f(s[i] * 1e-3f - m);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // keep the scores live without vector work: 4 of them stand in for the packed P
        typedef __attribute__((ext_vector_type(4))) float v4;
        v4 t = {s[8 * i], s[8 * i + 1], s[8 * i + 2], s[8 * i + 3]};
        pf[i] = __builtin_bit_cast(bf16x8, t);
      }
    }
    if (SHAPE == 0) {
      f32x16 o0, o1;
      for (int r = 0; r < 16; ++r) { o0[r] = o[r]; o1[r] = o[16 + r]; }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf[g], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf[g], o1, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) { o[r] = o0[r] * 0.5f; o[16 + r] = o1[r] * 0.5f; }
    } else {
      f32x4 oo[8];
      for (int i = 0; i < 8; ++i) oo[i] = f32x4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) oo[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[i & 1], pf[2 * g + (i >> 2 & 1)], oo[i], 0, 0, 0);
      for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 4; ++r) o[4 * i + r] = oo[i][r] * 0.5f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int i = 0; i < 32; ++i) acc += o[i];
  if (acc == 12345.678f) sink[0] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int VALU>
static double run(const char* name, const bf16x8* rnd, float* sink, unsigned long long* clk) {
  const int iters = 6000, nwg = 256 * 4 * 4;  // four rounds of full occupancy
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(nwg), dim3(256), 0, 0, rnd, sink, clk, iters);  // ~ 1 s of load first
  hipEventRecord(a);
  hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(nwg), dim3(256), 0, 0, rnd, sink, clk, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * nwg);
  hipMemcpy(h.data(), clk, 2 * nwg * 8, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int i = 0; i < nwg; ++i) mhz.push_back(100.0 * (double)h[2 * i] / (double)h[2 * i + 1]);
  std::sort(mhz.begin(), mhz.end());
  const double flop = (double)nwg * 4 /*waves*/ * iters * 16.0 * 32768.0;
  printf("%-58s %8.2f ms  %7.1f TFLOP/s  clock %6.0f MHz (median)\n", name, ms, flop / ms / 1e9, mhz[nwg / 2]);
  return ms;
}

int main() {
  const size_t n = 1 << 20;
  std::vector<unsigned short> h(n * 8);
  srand(1);
  for (auto& v : h) {  // random bf16 in (-2, 2), full-range mantissas and signs
    const float f = (rand() / (float)RAND_MAX * 4.f - 2.f);
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  bf16x8* rnd; float* sink; unsigned long long* clk;
  hipMalloc(&rnd, n * 16); hipMalloc(&sink, 4); hipMalloc(&clk, 2 * 4096 * 8);
  hipMemcpy(rnd, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    const double a = run<0, 1>("32x32x16, score + P V products with the softmax VALU", rnd, sink, clk);
    const double b = run<1, 1>("16x16x32, score + P V products with the softmax VALU", rnd, sink, clk);
    const double c = run<0, 0>("32x32x16, matrix instructions only", rnd, sink, clk);
    const double d = run<1, 0>("16x16x32, matrix instructions only", rnd, sink, clk);
    printf("   16x16x32 / 32x32x16 time: %.3f with VALU, %.3f matrix only\n", b / a, d / c);
  }
  return 0;
}
