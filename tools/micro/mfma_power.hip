// What does the bf16 matrix pipe deliver at the board's power cap?  A bare MFMA loop -- operands in registers (random bf16, full-range mantissas
// and signs: zero or trivial operands draw far less power, MI355X_MICROARCH.md "DVFS give-back"), no LDS, no memory, no vector work -- on every
// SIMD of the chip, launched back to back for `seconds`.  Run under tools/power_probe.py --cmd so that power, cap and clock are sampled beside it:
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_power tools/micro/mfma_power.hip
//   python tools/power_probe.py --cmd "tools/micro/mfma_power 16 5"      (shape 16 = v_mfma_f32_16x16x32_bf16, 32 = 32x32x16; seconds)
// Prints TFLOP/s per launch batch.  This is the ceiling a kernel made of nothing but matrix instructions reaches on this board.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// one wave per SIMD (256 threads per workgroup, one workgroup per CU): 8 x 8 independent 16x16 accumulators = the 128 x 128 wave tile of the
// four-wave GEMM kernels; 64 matrix instructions per iteration
__global__ __launch_bounds__(256, 1) void k16(const bf16x8* __restrict__ rnd, float* __restrict__ sink, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd[(tid * 16 + i) & 0xfffff]; b[i] = rnd[(tid * 16 + 8 + i) & 0xfffff]; }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.678f) *sink = s;
}
// four waves per SIMD (the attention kernels' occupancy): 2 x 2 independent 32x32 accumulators per wave; 4 matrix instructions per iteration step
__global__ __launch_bounds__(256, 4) void k32(const bf16x8* __restrict__ rnd, float* __restrict__ sink, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd[(tid * 16 + i) & 0xfffff]; b[i] = rnd[(tid * 16 + 8 + i) & 0xfffff]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) *sink = s;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 16;
  const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
  const size_t n = 1 << 20;
  std::vector<unsigned short> h(n * 8);
  srand(1);
  for (auto& v : h) {
    const float f = (rand() / (float)RAND_MAX * 4.f - 2.f);
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  bf16x8* rnd; float* sink;
  if (hipMalloc(&rnd, n * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemcpy(rnd, h.data(), n * 16, hipMemcpyHostToDevice);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = shape == 16 ? 20000 : 80000;
  // flops per launch: k16: grid cus x 4 waves x iters x 64 MFMAs x 16384; k32: grid 4 cus x 4 waves x iters x 16 MFMAs x 32768
  const double fl = shape == 16 ? (double)cus * 4 * iters * 64 * 16384.0 : (double)cus * 4 * 4 * iters * 16 * 32768.0;
  const auto t_begin = std::chrono::steady_clock::now();
  int batch = 0;
  for (;;) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    const int reps = 8;
    for (int r = 0; r < reps; ++r) {
      if (shape == 16) hipLaunchKernelGGL(k16, dim3(cus), dim3(256), 0, 0, rnd, sink, iters);
      else hipLaunchKernelGGL(k32, dim3(4 * cus), dim3(256), 0, 0, rnd, sink, iters);
    }
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s  batch %d: %.1f TFLOP/s (%.2f ms per launch)\n", shape == 16 ? "16x16x32, one wave per SIMD" : "32x32x16, four waves per SIMD", batch++, fl * reps / (ms * 1e-3) / 1e12, ms / reps);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > seconds) break;
  }
  return 0;
}
