// Do matrix instructions of ONE wave and vector instructions of ANOTHER wave on the same SIMD hide each other?  Workgroups of WPS * 256 threads
// (WPS waves per SIMD): "role M" waves run a loop of v_mfma_f32_32x32x16_bf16 on four rotating accumulators, "role V" waves a loop of
// independent vector instructions (v_fma_f32, or v_exp_f32), sized so that each loop alone takes about the same time.  Cycles (s_memtime) of the
// M waves and the V waves, alone and together: together == max(alone) means the two streams overlap across waves, together == sum that they
// do not.   hipcc --offload-arch=gfx950 -O3 -o cross_wave_overlap cross_wave_overlap.hip && ./cross_wave_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// MODE: 0 = the M waves work, the V waves exit at once; 1 = only the V waves work; 2 = both.  roles alternate by wave index: with WPS waves per
// SIMD, wave w sits on SIMD w % 4 (dispatch order), so waves w and w + 4 share a SIMD: role = (w / 4) & 1
template <int WPS, int MODE, int EXP>
__global__ __launch_bounds__(WPS * 256, 1) void k(unsigned long long* out, float* sink, int iters_m, int iters_v) {
  const int wave = threadIdx.x >> 6;
  const bool role_v = ((wave >> 2) & 1) != 0;
  unsigned long long t0 = 0, t1 = 0;
  float s = 0.f;
  if (!role_v) {
    if (MODE == 1) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    if (MODE == 0) return;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5f + 0.01f * threadIdx.x + i;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        if (EXP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
        else if (EXP == 2 && (u & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u & 7]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[u & 7]));
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) s += f[i];
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (WPS * 4) + wave] = t1 - t0;
  if (s == 12345.678f) sink[0] = s;
}

template <int WPS, int MODE, int EXP>
static void run(unsigned long long* d_out, float* d_sink, int im, int iv, double* tm, double* tv) {
  const int nwg = 256, nw = WPS * 4;
  hipMemset(d_out, 0, nwg * nw * 8);
  hipLaunchKernelGGL((k<WPS, MODE, EXP>), dim3(nwg), dim3(WPS * 256), 0, 0, d_out, d_sink, im, iv);
  hipMemset(d_out, 0, nwg * nw * 8);
  hipLaunchKernelGGL((k<WPS, MODE, EXP>), dim3(nwg), dim3(WPS * 256), 0, 0, d_out, d_sink, im, iv);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg * nw);
  hipMemcpy(h.data(), d_out, nwg * nw * 8, hipMemcpyDeviceToHost);
  double sm = 0, sv = 0; int nm = 0, nv = 0;
  for (int b = 0; b < nwg; ++b)
    for (int w = 0; w < nw; ++w) {
      const double v = (double)h[b * nw + w];
      if (v == 0) continue;
      if ((w >> 2) & 1) { sv += v; ++nv; } else { sm += v; ++nm; }
    }
  *tm = nm ? sm / nm : 0; *tv = nv ? sv / nv : 0;
}

template <int WPS, int EXP>
static void trio(const char* name, unsigned long long* d_out, float* d_sink, int im, int iv) {
  double m0, v0, m1, v1, m2, v2;
  run<WPS, 0, EXP>(d_out, d_sink, im, iv, &m0, &v0);
  run<WPS, 1, EXP>(d_out, d_sink, im, iv, &m1, &v1);
  run<WPS, 2, EXP>(d_out, d_sink, im, iv, &m2, &v2);
  printf("%-52s M alone %9.0f  V alone %9.0f  together: M %9.0f  V %9.0f   (memtime ticks; max %.0f, sum %.0f)\n", name, m0, v1, m2, v2,
         m0 > v1 ? m0 : v1, m0 + v1);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 256 * 16 * 8); hipMalloc(&d_sink, 4);
  // 8 MFMAs of 32 cycles = 256 cycles per M iteration; 32 v_fma of 4 cycles = 128 per V iteration (v_exp: 8 or 16 each)
  trio<2, 0>("2 waves/SIMD: 1 M + 1 V (v_fma)", d_out, d_sink, 4000, 8000);
  trio<2, 1>("2 waves/SIMD: 1 M + 1 V (v_exp)", d_out, d_sink, 4000, 4000);
  trio<2, 2>("2 waves/SIMD: 1 M + 1 V (1 v_exp : 3 v_fma)", d_out, d_sink, 4000, 6000);
  trio<4, 0>("4 waves/SIMD: 2 M + 2 V (v_fma)", d_out, d_sink, 2000, 4000);
  trio<4, 2>("4 waves/SIMD: 2 M + 2 V (1 v_exp : 3 v_fma)", d_out, d_sink, 2000, 3000);
  return 0;
}
