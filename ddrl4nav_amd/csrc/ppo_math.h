// Wave-level helpers and the per-sample arithmetic of the PPO loss block shared by the categorical (heads.hip) and Gaussian
// (gheads.hip) head kernels.  Reference: USTC_lab/nn/ppo.py:82-108 and the autograd backward of
// torch.min / torch.max / torch.clamp / torch.where (ties split evenly, clamp passes gradient on
// the closed interval).
#pragma once
#include "common.h"

namespace ddrl {

// ---- one wavefront per sample: 512 features = 8 per lane -----------------------------------------
__device__ __forceinline__ float wave_sum(float v) {  // wave64 xor-shuffle butterfly, result in every lane
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void load8(const float* p, float* o) {
  const float4 x = ((const float4*)p)[0], y = ((const float4*)p)[1];
  o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
  o[4] = y.x; o[5] = y.y; o[6] = y.z; o[7] = y.w;
}
__device__ __forceinline__ void store8(float* p, const float* o) {
  ((float4*)p)[0] = make_float4(o[0], o[1], o[2], o[3]);
  ((float4*)p)[1] = make_float4(o[4], o[5], o[6], o[7]);
}

struct SurrogateTerm {
  float term;    // where(adv > 0, m, max(m, dual_clip * adv)),  m = min(ratio * adv, clamp(ratio) * adv)
  float g_logp;  // d(actor_loss)/d(log pi(a|s)), actor_loss = -mean(term): already scaled by inv_b
};

__device__ __forceinline__ SurrogateTerm ppo_surrogate(float logp, float old_logp, float adv, const ddrl_config& cfg,
                                                       float inv_b) {
  const float lo = 1.0f - cfg.ppo_clip, hi = 1.0f + cfg.ppo_clip;
  const float ratio = expf(logp - old_logp);
  const float surr1 = ratio * adv;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float surr2 = rc * adv;
  const float mn = fminf(surr1, surr2);
  const float dual = cfg.dual_clip * adv;
  SurrogateTerm o;
  o.term = (adv > 0.0f) ? mn : fmaxf(mn, dual);
  const float g_term = -inv_b;
  float g_mn;
  if (adv > 0.0f) g_mn = g_term;
  else g_mn = (mn > dual) ? g_term : ((mn == dual) ? 0.5f * g_term : 0.0f);
  const float g_s1 = (surr1 < surr2) ? g_mn : ((surr1 == surr2) ? 0.5f * g_mn : 0.0f);
  const float g_s2 = (surr2 < surr1) ? g_mn : ((surr1 == surr2) ? 0.5f * g_mn : 0.0f);
  const float inrange = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
  const float g_ratio = g_s1 * adv + g_s2 * adv * inrange;
  o.g_logp = g_ratio * ratio;
  return o;
}

// Value-loss element of one sample: returns d(element)/d(v) (before the 1/B) and adds the element to
// `sum`.  mean((ret - v)^2) / 2 accumulates err^2 (the 1/2 is applied by the reduce kernels), or
// F.smooth_l1_loss(ret, v) with beta = 1 (ppo.py:53-57).
__device__ __forceinline__ float value_loss_element(float err, const ddrl_config& cfg, double& sum) {
  if (cfg.smooth_l1_loss) {
    const float ae = fabsf(err);
    sum += (ae < 1.0f) ? 0.5 * (double)err * (double)err : (double)ae - 0.5;
    return (err < -1.0f) ? 1.0f : ((err > 1.0f) ? -1.0f : -err);
  }
  sum += (double)err * (double)err;
  return -err;
}

}  // namespace ddrl
