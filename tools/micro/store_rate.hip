// Micro-benchmark: how fast can ONE workgroup (512 threads, one per CU) store a 256 x 256 bf16 tile (rows of 512 B, row stride
// `stride` bytes) with 16-byte stores, as a function of how many CUs do it at once?   hipcc --offload-arch=gfx950 -O3 store_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void store_tiles(char* out, int stride, int tiles_per_wg, int ntiles_n, int rows_per_instr_mode,
                                                   long long* cycles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint4 v = make_uint4(tid, blockIdx.x, 3, 4);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int tm = tile / ntiles_n, tn = tile % ntiles_n;
    char* base = out + (size_t)tm * 256 * stride + (size_t)tn * 512;
    // 256 rows x 512 B; a wave instruction covers 2 rows (32 lanes x 16 B each); 16 instructions per wave
    if (rows_per_instr_mode == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i * 8 + wave) * 2 + (lane >> 5);
        *reinterpret_cast<uint4*>(base + (size_t)row * stride + (lane & 31) * 16) = v;
      }
    } else if (rows_per_instr_mode == 1) {
      // MFMA layout: wave (wm = wave>>2, wn = wave&3) owns rows 128wm.., cols 64wn.. (bf16: 128 B); lane (c, kq): 16 rows x 64 B per instr
      const int c = lane & 15, kq = lane >> 4, wm = wave >> 2, wn = wave & 3;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          *reinterpret_cast<uint4*>(base + (size_t)(wm * 128 + 16 * i + c) * stride + (wn * 64 + 32 * jj + 8 * kq) * 2) = v;
    } else {
      // 8 rows x 128 B per instr
      const int c = lane & 7, g = lane >> 3, wm = wave >> 2, wn = wave & 3;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *reinterpret_cast<uint4*>(base + (size_t)(wm * 128 + 8 * i + c) * stride + (wn * 64 + 8 * g) * 2) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int stride = 3072 * 2, M = 50176, ntn = 12;
  char* out;
  long long* cyc;
  hipMalloc(&out, (size_t)M * stride);
  hipMalloc(&cyc, 4096 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode : {0, 1, 2})
  for (int grid : {32, 256}) {
    for (int tpw : {1, 4}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(store_tiles, dim3(grid), dim3(512), 0, 0, out, stride, tpw, ntn, mode, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(grid);
      hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double kb = 128.0 * tpw;
      printf("mode %d grid %4d tiles/wg %d: kernel %7.1f us | in-kernel cycles median %8lld max %8lld -> %.1f B/clk/CU (median), chip %.2f TB/s\n", mode, grid, tpw,
             ms * 1000, h[grid / 2], h[grid - 1], kb * 1024 / h[grid / 2], kb * 1024.0 * grid / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
