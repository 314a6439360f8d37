// Micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 (dependent chain vs independent chains), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float t = 0.999f, c = 1e-3f;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const f32x2 t2 = {t, t}, c2 = {c, c};
  const long long s = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(t), "v"(c)); }
      if (MODE == 1) {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t), "v"(c));
      }
      if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(t2), "v"(c2)); }
      if (MODE == 3) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(t2), "v"(c2));
      }
      if (MODE == 4) {  // 8 independent scalar chains
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(t), "v"(c));
      }
    }
  }
  const long long e = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = e - s;
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 2000;
  const char* names[] = {"v_fma_f32 dependent", "v_fma_f32 4 chains", "v_pk_fma_f32 dependent", "v_pk_fma_f32 4 chains", "v_fma_f32 8 chains"};
  const int per_iter[] = {16, 64, 16, 64, 128};
  for (int threads : {256, 512}) {
    for (int m = 0; m < 5; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
      }
      long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double per = (double)h / iters / per_iter[m];
      printf("%d waves/SIMD  %-26s %.2f cycles per instruction per wave  (%.2f per FMA-lane-element)\n", threads / 256, names[m], per,
             per / ((m == 2 || m == 3) ? 2 : 1));
    }
  }
  return 0;
}
