// Gaussian actor head + critic head for the continuous-action nets (robot navigation), with the
// same PPO dual-clip loss block as heads.hip.  One wavefront per sample, D <= 8 action dims.
//
// Reference arithmetic replaced:
//   GaussionActor._distribution / _log_prob_from_distribution   USTC_lab/nn/actor.py:43-70
//     mu = actor_linear(h); std = exp(log_std); Normal(mu, std); log_prob(act).sum(-1)
//   torch.distributions.Normal.log_prob / entropy / sample
//     -((a-mu)^2)/(2 var) - log(std) - log(sqrt(2 pi));  0.5 + 0.5 log(2 pi) + log(std);  mu + std * eps
//   Critic.forward                                             USTC_lab/nn/critic.py:14-21
//   PPO.learn loss block + autograd                            USTC_lab/nn/ppo.py:82-129
#include "kernels.h"
#include "ops.h"
#include "ppo_math.h"

namespace ddrl {

constexpr int MAXD = 8;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;  // math.log(math.sqrt(2 * math.pi))


struct GHeadRegs {
  float wa[MAXD][8], wc[8], ba[MAXD], std[MAXD], var[MAXD], log_scale[MAXD], bc;
};

__device__ __forceinline__ void gload_weights(GHeadRegs& R, const float* params, const GaussLayout& L, int lane) {
#pragma unroll
  for (int d = 0; d < MAXD; ++d) {
    const int dd = min(d, L.D - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) R.wa[d][i] = (d < L.D) ? params[L.actor_w + (int64_t)dd * FEAT + lane * 8 + i] : 0.0f;
    R.ba[d] = (d < L.D) ? params[L.actor_b + dd] : 0.0f;
    const float s = expf(params[L.log_std + dd]);  // std = torch.exp(self.log_std)
    R.std[d] = s;
    R.var[d] = s * s;          // Normal.log_prob: var = scale ** 2
    R.log_scale[d] = logf(s);  // log_scale = scale.log()
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) R.wc[i] = params[L.critic_w + lane * 8 + i];
  R.bc = params[L.critic_b];
}

// standard normal from two counter-based uniforms (Box-Muller); u1 in (0,1]
__device__ __forceinline__ float hash_normal(uint64_t seed, uint64_t stream, uint64_t idx) {
  const float u1 = 1.0f - hash_uniform(seed, stream, 2 * idx);
  const float u2 = hash_uniform(seed, stream, 2 * idx + 1);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(256) void gauss_act_kernel(const float* __restrict__ h_actor, const float* __restrict__ h_critic,
                                                        const float* __restrict__ params, GaussLayout L, int n,
                                                        const float* __restrict__ act_in, uint64_t seed, uint64_t stream_id,
                                                        float* __restrict__ mu_out, float* __restrict__ value,
                                                        float* __restrict__ action_out, float* __restrict__ logp_out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  GHeadRegs R;
  gload_weights(R, params, L, lane);
  for (int b = gw; b < n; b += nw) {
    float ha[8], hc[8];
    load8(h_actor + (int64_t)b * FEAT + lane * 8, ha);
    load8(h_critic + (int64_t)b * FEAT + lane * 8, hc);
    float sv = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sv = __builtin_fmaf(hc[i], R.wc[i], sv);
    const float v = wave_sum(sv) + R.bc;
    float logp = 0.0f;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s = __builtin_fmaf(ha[i], R.wa[d][i], s);
      const float mu = wave_sum(s) + R.ba[d];
      if (d < L.D) {
        float a;
        if (act_in != nullptr) a = act_in[(int64_t)b * L.D + d];
        else a = mu + R.std[d] * hash_normal(seed, stream_id, (uint64_t)b * L.D + d);
        const float diff = a - mu;
        logp += -(diff * diff) / (2.0f * R.var[d]) - R.log_scale[d] - LOG_SQRT_2PI;
        if (lane == 0) {
          if (mu_out) mu_out[(int64_t)b * L.D + d] = mu;
          if (action_out) action_out[(int64_t)b * L.D + d] = a;
        }
      }
    }
    if (lane == 0) {
      value[b] = v;
      if (logp_out) logp_out[b] = logp;
    }
  }
}

// hpart per workgroup: [D*512 dWa][512 dwc][D dba][1 dbc][D dlog_std][actor_sum, v_sum, ent_sum]
__global__ __launch_bounds__(256) void gauss_loss_kernel(
    const float* __restrict__ h_actor, const float* __restrict__ h_critic, const float* __restrict__ params, GaussLayout L,
    ddrl_config cfg, int n, const float* __restrict__ actions, const float* __restrict__ old_logps,
    const float* __restrict__ advs, const float* __restrict__ rets, float inv_b, float* __restrict__ dh_actor,
    float* __restrict__ dh_critic, float* __restrict__ dmu_out, float* __restrict__ dvalue, float* __restrict__ hpart,
    int64_t hstride) {
  __shared__ float red[(MAXD + 1) * FEAT + 3 * MAXD + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int D = L.D;
  const bool shared = L.shared != 0;
  GHeadRegs R;
  gload_weights(R, params, L, lane);
  float gwa[MAXD][8], gwc[8], gba[MAXD], gls[MAXD], gbc = 0.0f;
#pragma unroll
  for (int d = 0; d < MAXD; ++d) {
    gba[d] = gls[d] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) gwa[d][i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) gwc[i] = 0.0f;
  double s_actor = 0.0, s_v = 0.0, s_ent = 0.0;
  float ent_b = 0.0f;  // sum_d entropy_d / D : identical for every sample
#pragma unroll
  for (int d = 0; d < MAXD; ++d)
    if (d < D) ent_b += 0.5f + LOG_SQRT_2PI + R.log_scale[d];  // 0.5 + 0.5*log(2 pi) + log(std)
  ent_b = ent_b / (float)D;

  for (int b = gw; b < n; b += nw) {
    float ha[8], hc[8];
    load8(h_actor + (int64_t)b * FEAT + lane * 8, ha);
    load8(h_critic + (int64_t)b * FEAT + lane * 8, hc);
    float sv = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sv = __builtin_fmaf(hc[i], R.wc[i], sv);
    const float v = wave_sum(sv) + R.bc;
    float diff[MAXD], logp = 0.0f;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s = __builtin_fmaf(ha[i], R.wa[d][i], s);
      const float mu = wave_sum(s) + R.ba[d];
      diff[d] = (d < D) ? actions[(int64_t)b * D + min(d, D - 1)] - mu : 0.0f;
      if (d < D) logp += -(diff[d] * diff[d]) / (2.0f * R.var[d]) - R.log_scale[d] - LOG_SQRT_2PI;
    }
    const float adv = advs[b];
    const SurrogateTerm sg = ppo_surrogate(logp, old_logps[b], adv, cfg, inv_b);
    s_actor += (double)sg.term;
    const float err = rets[b] - v;
    const float gv_unit = value_loss_element(err, cfg, s_v);
    s_ent += (double)ent_b;
    const float g_logp = sg.g_logp;
    const float gv = shared ? gv_unit * inv_b * cfg.v_loss_theta : gv_unit * inv_b;
    float dmu[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      dmu[d] = (d < D) ? g_logp * (diff[d] / R.var[d]) : 0.0f;
      // d logp / d log_std = diff^2 / var - 1 ; entropy term only when total_loss is differentiated
      float gl = (d < D) ? g_logp * ((diff[d] * diff[d]) / R.var[d] - 1.0f) : 0.0f;
      if (shared && d < D) gl += -cfg.ent_loss_theta * inv_b / (float)D;
      gls[d] += gl;
      gba[d] += dmu[d];
    }
    float da[8], dc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < MAXD; ++d) s = __builtin_fmaf(dmu[d], R.wa[d][i], s);
      da[i] = s;
      dc[i] = gv * R.wc[i];
      gwc[i] = __builtin_fmaf(gv, hc[i], gwc[i]);
    }
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
#pragma unroll
      for (int i = 0; i < 8; ++i) gwa[d][i] = __builtin_fmaf(dmu[d], ha[i], gwa[d][i]);
    gbc += gv;
    if (shared) {
#pragma unroll
      for (int i = 0; i < 8; ++i) da[i] += dc[i];
      store8(dh_actor + (int64_t)b * FEAT + lane * 8, da);
    } else {
      store8(dh_actor + (int64_t)b * FEAT + lane * 8, da);
      store8(dh_critic + (int64_t)b * FEAT + lane * 8, dc);
    }
    if (lane < D) {
      float x = dmu[0];
#pragma unroll
      for (int d = 1; d < MAXD; ++d) x = (lane == d) ? dmu[d] : x;
      dmu_out[(int64_t)b * D + lane] = x;
    }
    if (lane == 0) dvalue[b] = gv;
  }

  constexpr int SCAL = (MAXD + 1) * FEAT;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      const bool first = (w == 0);
#pragma unroll
      for (int d = 0; d < MAXD; ++d)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int idx = d * FEAT + lane * 8 + i;
          red[idx] = first ? gwa[d][i] : red[idx] + gwa[d][i];
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = MAXD * FEAT + lane * 8 + i;
        red[idx] = first ? gwc[i] : red[idx] + gwc[i];
      }
      if (lane == 0) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) {
          red[SCAL + d] = first ? gba[d] : red[SCAL + d] + gba[d];
          red[SCAL + MAXD + d] = first ? gls[d] : red[SCAL + MAXD + d] + gls[d];
        }
        red[SCAL + 2 * MAXD] = first ? gbc : red[SCAL + 2 * MAXD] + gbc;
        red[SCAL + 2 * MAXD + 1] = first ? (float)s_actor : red[SCAL + 2 * MAXD + 1] + (float)s_actor;
        red[SCAL + 2 * MAXD + 2] = first ? (float)s_v : red[SCAL + 2 * MAXD + 2] + (float)s_v;
        red[SCAL + 2 * MAXD + 3] = first ? (float)s_ent : red[SCAL + 2 * MAXD + 3] + (float)s_ent;
      }
    }
    __syncthreads();
  }
  float* out = hpart + (int64_t)blockIdx.x * hstride;
  for (int i = threadIdx.x; i < D * FEAT; i += 256) out[i] = red[i];
  for (int i = threadIdx.x; i < FEAT; i += 256) out[D * FEAT + i] = red[MAXD * FEAT + i];
  const int o = (D + 1) * FEAT;
  if (threadIdx.x < D) {
    out[o + threadIdx.x] = red[SCAL + threadIdx.x];
    out[o + D + 1 + threadIdx.x] = red[SCAL + MAXD + threadIdx.x];
  }
  if (threadIdx.x == 0) out[o + D] = red[SCAL + 2 * MAXD];
  if (threadIdx.x < 3) out[o + 2 * D + 1 + threadIdx.x] = red[SCAL + 2 * MAXD + 1 + threadIdx.x];
}

__global__ __launch_bounds__(256) void gauss_reduce_kernel(const float* __restrict__ hpart, int64_t hstride, int nwg,
                                                           GaussLayout L, ddrl_config cfg, float inv_b,
                                                           float* __restrict__ grads) {
  __shared__ double sh[8][RED_OUT];
  const int D = L.D;
  const int o = (D + 1) * FEAT;
  const int nsum = o + 2 * D + 1;
  if (blockIdx.x == gridDim.x - 1) {  // the three loss sums, one wave each
    const int k = threadIdx.x >> 6;
    if (k >= 3) return;
    const double s = wave_sum_partials(hpart, hstride, nwg, nsum + k);
    double r;
    if (k == 0) r = -s * (double)inv_b;
    else if (k == 1) r = s * (double)inv_b * (cfg.smooth_l1_loss ? 1.0 : 0.5);
    else r = s * (double)inv_b;
    if ((threadIdx.x & 63) == 0) grads[L.n_params + k] = (float)r;
    return;
  }
  // summed in double, rounded once, in the fixed order of ppo_math.h sum_partials8 (the arithmetic of the one-thread-per-element form)
  const int i = blockIdx.x * RED_OUT + (threadIdx.x & (RED_OUT - 1));
  const float s = sum_partials8(hpart, hstride, nwg, min(i, nsum - 1), sh);
  if (threadIdx.x >= RED_OUT || i >= nsum) return;
  int64_t dst;
  if (i < D * FEAT) dst = L.actor_w + i;
  else if (i < o) dst = L.critic_w + (i - D * FEAT);
  else if (i < o + D) dst = L.actor_b + (i - o);
  else if (i == o + D) dst = L.critic_b;
  else dst = L.log_std + (i - (o + D + 1));
  grads[dst] = s;
}

int64_t gauss_hpart_stride(int D) { return align_up((int64_t)(D + 1) * FEAT + 2 * D + 1 + 3, 64); }

void launch_gauss_act(const GaussLayout& L, const float* params, const float* h_actor, const float* h_critic, int n,
                      const float* act_in, uint64_t seed, uint64_t stream_id, float* mu_out, float* value, float* action_out,
                      float* logp_out, hipStream_t st) {
  int wgs = (n + 3) / 4;
  if (wgs > 1024) wgs = 1024;
  hipLaunchKernelGGL(gauss_act_kernel, dim3(wgs), dim3(256), 0, st, h_actor, h_critic, params, L, n, act_in, seed, stream_id,
                     mu_out, value, action_out, logp_out);
}

void launch_gauss_loss(const GaussLayout& L, const ddrl_config& cfg, const float* params, const float* h_actor,
                       const float* h_critic, int n, const float* actions, const float* old_logps, const float* advs,
                       const float* rets, float inv_b, float* dh_actor, float* dh_critic, float* dmu, float* dvalue,
                       float* hpart, float* grads, hipStream_t st) {
  const int64_t hs = gauss_hpart_stride(L.D);
  hipLaunchKernelGGL(gauss_loss_kernel, dim3(HEAD_WG), dim3(256), 0, st, h_actor, h_critic, params, L, cfg, n, actions,
                     old_logps, advs, rets, inv_b, dh_actor, dh_critic, dmu, dvalue, hpart, hs);
  const int nsum = (L.D + 1) * FEAT + 2 * L.D + 1;  // gradient elements; + one workgroup for the three loss sums
  hipLaunchKernelGGL(gauss_reduce_kernel, dim3((nsum + RED_OUT - 1) / RED_OUT + 1), dim3(256), 0, st, hpart, hs, HEAD_WG, L, cfg, inv_b,
                     grads);
}

}  // namespace ddrl
