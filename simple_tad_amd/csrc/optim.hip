// Fused multi-tensor AdamW over flat buffers: ONE launch updates every parameter of the model, applies the per-group
// learning rate / weight decay of the reference's layer-decay parameter groups (optim_factory.py:49-88, 126-127; per-step
// assignment engine_for_finetuning.py:49-54), optionally scales the gradient by a device-side clip coefficient, emits the bf16
// operand copy the next forward's GEMMs read, and leaves per-chunk partial sums of g^2 for the gradient norm
// (utils.get_grad_norm_, utils.py:415-427) -- so the optimizer tail of a step is one HBM pass (30 B / parameter) instead of
// ~150 casts + a 50-group multi-tensor apply + a separate norm pass.
//
// Update rule = torch.optim.AdamW (single-tensor form, amsgrad off, maximize off), in its operation order:
//   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g g;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include "common.h"
#include <math.h>

TAD_NAMESPACE_BEGIN

struct AdamGroups {
  float decay[TAD_ADAMW_MAX_GROUPS];      // 1 - lr*wd
  float step_size[TAD_ADAMW_MAX_GROUPS];  // lr / (1 - beta1^t)
  float bc2_sqrt[TAD_ADAMW_MAX_GROUPS];   // sqrt(1 - beta2^t)
};

// One workgroup per TAD_ADAMW_CHUNK (4096) elements: 256 threads x 4 iterations x float4; every tensor of the flat layout starts
// on a chunk boundary, so a chunk belongs to exactly one parameter group (chunk_group[c]; 255 = untouched this step).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ pb,
                                                    const uint8_t* __restrict__ chunk_group, const AdamGroups hp, float b1, float b2,
                                                    float eps, const float* __restrict__ grad_scale,
                                                    float* __restrict__ sumsq_partial, int64_t n) {
  __shared__ float red[4];
  const int grp = chunk_group[blockIdx.x];
  float sq = 0.f;
  const float gs = grad_scale ? *grad_scale : 1.f;
  // *grad_scale == 0 (or not finite) skips the update: what GradScaler.step does when the scaled gradients overflowed (utils.py:386-412;
  // the host folds "found inf" into the scale: engine.NativeScalerWithGradNormCount), decided on the device without a host sync
  const bool skip = grad_scale && !(fabsf(gs) > 0.f && fabsf(gs) < INFINITY);
  if (grp != 255 && !skip) {
    const float decay = hp.decay[grp], ss = hp.step_size[grp], bc2_sqrt = hp.bc2_sqrt[grp];
    const int64_t base = (int64_t)blockIdx.x * TAD_ADAMW_CHUNK;
#pragma unroll
    for (int it = 0; it < TAD_ADAMW_CHUNK / 1024; ++it) {
      const int64_t i = base + (int64_t)(it * 256 + threadIdx.x) * 4;
      if (i >= n) break;
      float4 pv = *reinterpret_cast<const float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<const float4*>(m + i);
      float4 vv = *reinterpret_cast<const float4*>(v + i);
      float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w},
            va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sq = fmaf(ga[e], ga[e], sq);
        const float gg = ga[e] * gs;
        ma[e] = ma[e] + (1.f - b1) * (gg - ma[e]);
        va[e] = b2 * va[e] + (1.f - b2) * gg * gg;
        const float denom = sqrtf(va[e]) / bc2_sqrt + eps;
        pa[e] = pa[e] * decay - ss * (ma[e] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
      *reinterpret_cast<float4*>(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
      *reinterpret_cast<float4*>(v + i) = make_float4(va[0], va[1], va[2], va[3]);
      if (pb) *reinterpret_cast<uint2*>(pb + i) = make_uint2(pack_op16x2(pa[0], pa[1]), pack_op16x2(pa[2], pa[3]));
    }
  }
  if (sumsq_partial) {  // fixed reduction order: deterministic norm
    if (grp != 255 && skip) {  // a skipped step still reports the norm of what it skipped
      const int64_t base = (int64_t)blockIdx.x * TAD_ADAMW_CHUNK;
#pragma unroll
      for (int it = 0; it < TAD_ADAMW_CHUNK / 1024; ++it) {
        const int64_t i = base + (int64_t)(it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        sq = fmaf(gv.x, gv.x, fmaf(gv.y, gv.y, fmaf(gv.z, gv.z, fmaf(gv.w, gv.w, sq))));
      }
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) sumsq_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

TAD_NAMESPACE_END

using namespace tad;

extern "C" int tad_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint16_t* param_bf16,
                              const uint8_t* chunk_group, int64_t n, const float* group_lr, const float* group_wd, int n_groups,
                              const int32_t* group_step, float beta1, float beta2, float eps, const float* grad_scale,
                              float* sumsq_partials, tad_stream_t stream) {
  TAD_REQUIRE(param && grad && exp_avg && exp_avg_sq && chunk_group && group_lr && group_wd && group_step, "adamw_step: null pointer");
  TAD_REQUIRE(n > 0 && (n & 3) == 0, "adamw_step: n=%lld must be a positive multiple of 4", (long long)n);
  TAD_REQUIRE(n_groups > 0 && n_groups <= TAD_ADAMW_MAX_GROUPS, "adamw_step: n_groups=%d out of range (1..%d)", n_groups,
              TAD_ADAMW_MAX_GROUPS);
  TAD_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adamw_step: bad betas / eps");
  TAD_REQUIRE(((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) == 0 &&
                  (((uintptr_t)param_bf16) & 7) == 0,
              "adamw_step: buffers must be 16-byte aligned");
  AdamGroups hp;
  for (int i = 0; i < n_groups; ++i) {
    TAD_REQUIRE(group_lr[i] >= 0.f && group_wd[i] >= 0.f, "adamw_step: negative lr / weight decay in group %d", i);
    TAD_REQUIRE(group_step[i] >= 1, "adamw_step: step counts from 1 (group %d has %d)", i, (int)group_step[i]);
    const double bc1 = 1.0 - pow((double)beta1, (double)group_step[i]), bc2 = 1.0 - pow((double)beta2, (double)group_step[i]);
    hp.decay[i] = (float)(1.0 - (double)group_lr[i] * (double)group_wd[i]);
    hp.step_size[i] = (float)((double)group_lr[i] / bc1);
    hp.bc2_sqrt[i] = (float)sqrt(bc2);
  }
  for (int i = n_groups; i < TAD_ADAMW_MAX_GROUPS; ++i) { hp.decay[i] = 1.f; hp.step_size[i] = 0.f; hp.bc2_sqrt[i] = 1.f; }
  const int64_t chunks = (n + TAD_ADAMW_CHUNK - 1) / TAD_ADAMW_CHUNK;
  TAD_REQUIRE(chunks < (1ll << 31), "adamw_step: too many chunks");
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, param_bf16,
                     chunk_group, hp, beta1, beta2, eps, grad_scale, sumsq_partials, n);
  return check_launch("adamw_step");
}
