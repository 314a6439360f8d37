// Does vector work issue in the shadow of a matrix instruction of the SAME wave?  One wave per SIMD (launch_bounds(256, 1), 256 CUs),
// a loop of { 1 MFMA 32x32x16 bf16, F independent VALU fillers }, cycles per iteration from s_memtime, for
//   dest = accumulation registers (AGPR)  vs  arch VGPRs,   chain = one accumulator (dependent) vs four rotating ones,
//   fillers = v_fma_f32 or v_exp_f32.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int DEST_V, int NACC, int FILL, int EXP, int SRCB_A = 0, int PRENOP = 0>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = 0.5f + 0.01f * threadIdx.x + i;
  if (SRCB_A) asm volatile("" : "+a"(b));
  if (DEST_V) {
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
  } else {
    asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (PRENOP) asm volatile("s_nop 1");
      if (DEST_V && SRCB_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "a"(b));
      else if (DEST_V) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
      else if (SRCB_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "a"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < FILL; ++j) {
        if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[j & 7]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j & 7]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 12345.678f) sink[0] = s;
}

template <int DEST_V, int NACC, int FILL, int EXP, int SRCB_A = 0, int PRENOP = 0>
static void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int iters = 4000, nwg = 256;
  hipLaunchKernelGGL((k<DEST_V, NACC, FILL, EXP, SRCB_A, PRENOP>), dim3(nwg), dim3(256), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((k<DEST_V, NACC, FILL, EXP, SRCB_A, PRENOP>), dim3(nwg), dim3(256), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg);
  hipMemcpy(h.data(), d_out, nwg * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-44s %7.1f cycles per MFMA (+%d fillers)\n", name, sum / nwg / (4.0 * iters), FILL);
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 256 * 8);
  hipMalloc(&d_sink, 4);
  run<0, 4, 0, 0>("dest AGPR, 4 accumulators, no fillers", d_out, d_sink);
  run<1, 4, 0, 0>("dest VGPR, 4 accumulators, no fillers", d_out, d_sink);
  run<0, 1, 0, 0>("dest AGPR, 1 accumulator (chain), no fillers", d_out, d_sink);
  run<1, 1, 0, 0>("dest VGPR, 1 accumulator (chain), no fillers", d_out, d_sink);
  run<0, 2, 0, 0>("dest AGPR, 2 accumulators, no fillers", d_out, d_sink);
  run<1, 2, 0, 0>("dest VGPR, 2 accumulators, no fillers", d_out, d_sink);
  run<0, 4, 5, 0>("dest AGPR, 4 acc, 5 v_fma", d_out, d_sink);
  run<1, 4, 5, 0>("dest VGPR, 4 acc, 5 v_fma", d_out, d_sink);
  run<0, 2, 5, 0>("dest AGPR, 2 acc, 5 v_fma", d_out, d_sink);
  run<1, 2, 5, 0>("dest VGPR, 2 acc, 5 v_fma", d_out, d_sink);
  run<0, 4, 8, 0>("dest AGPR, 4 acc, 8 v_fma", d_out, d_sink);
  run<1, 4, 8, 0>("dest VGPR, 4 acc, 8 v_fma", d_out, d_sink);
  run<0, 4, 3, 1>("dest AGPR, 4 acc, 3 v_exp", d_out, d_sink);
  run<1, 4, 3, 1>("dest VGPR, 4 acc, 3 v_exp", d_out, d_sink);
  run<1, 2, 0, 0, 1>("dest VGPR, src B AGPR, 2 acc, no fillers", d_out, d_sink);
  run<1, 2, 5, 0, 1>("dest VGPR, src B AGPR, 2 acc, 5 v_fma", d_out, d_sink);
  run<0, 4, 5, 0, 1>("dest AGPR, src B AGPR, 4 acc, 5 v_fma", d_out, d_sink);
  run<1, 2, 5, 0, 0, 1>("dest VGPR, 2 acc, 5 v_fma, s_nop 1 before MFMA", d_out, d_sink);
  run<1, 2, 4, 0, 1, 1>("dest VGPR, src B AGPR, 2 acc, 4 v_fma, s_nop 1", d_out, d_sink);
  return 0;
}
