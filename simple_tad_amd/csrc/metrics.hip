// Evaluation path: integer bookkeeping behind the thresholded metrics of engine_for_frame_finetuning.calculate_metrics
// (:593-636) and anaysis/metrics.calculate_MORE_metrics (:127-208) -- both evaluate `pred >= t` for the 101 thresholds
// np.arange(0, 1.001, 0.01).  For ascending thresholds the whole family of confusion matrices follows from ONE histogram per label
// of k(p) = #{t : p >= t}: a sample is predicted positive at threshold index i exactly when i < k(p).  Counts are exact integers.
#include "common.h"

TAD_NAMESPACE_BEGIN

constexpr int MAX_THR = 255;

__global__ __launch_bounds__(256) void threshold_hist_kernel(const float* __restrict__ probs, const int32_t* __restrict__ labels,
                                                             const float* __restrict__ thr, int T, int64_t n,
                                                             unsigned long long* __restrict__ hist /*[2][T+1]*/) {
  __shared__ float sthr[MAX_THR];
  __shared__ unsigned int sh[2 * (MAX_THR + 1)];
  for (int i = threadIdx.x; i < T; i += 256) sthr[i] = thr[i];
  for (int i = threadIdx.x; i < 2 * (T + 1); i += 256) sh[i] = 0u;
  __syncthreads();
  // a block handles at most 2^20 samples, so the 32-bit LDS counters cannot overflow
  const int64_t chunk = 1 << 20;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float p = probs[i];
    int a = 0, b = T;  // first index with thr > p (NaN compares false everywhere: k = 0)
    while (a < b) {
      const int m = (a + b) >> 1;
      if (p >= sthr[m]) a = m + 1; else b = m;
    }
    atomicAdd(&sh[(labels[i] != 0 ? (T + 1) : 0) + a], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (T + 1); i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" int tad_threshold_histogram(const float* probs, const int32_t* labels, const float* thresholds, int n_thresholds, int64_t n,
                                       int64_t* hist, tad_stream_t stream) {
  TAD_REQUIRE(probs && labels && thresholds && hist, "threshold_histogram: null pointer");
  TAD_REQUIRE(n > 0 && n_thresholds > 0 && n_thresholds <= MAX_THR, "threshold_histogram: need n > 0 and 1..%d thresholds", MAX_THR);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(hist, 0, sizeof(int64_t) * 2 * (n_thresholds + 1), st) != hipSuccess) { set_error("threshold_histogram: memset failed"); return TAD_EINVAL; }
  const int64_t blocks = (n + (1 << 20) - 1) >> 20;
  hipLaunchKernelGGL(threshold_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, st, probs, labels, thresholds, n_thresholds, n,
                     reinterpret_cast<unsigned long long*>(hist));
  return check_launch("threshold_histogram");
}
