// Wave-level helpers and the per-sample arithmetic of the PPO loss block shared by the categorical (heads.hip) and Gaussian
// (gheads.hip) head kernels.  Reference: USTC_lab/nn/ppo.py:82-108 and the autograd backward of
// torch.min / torch.max / torch.clamp / torch.where (ties split evenly, clamp passes gradient on
// the closed interval).
#pragma once
#include "common.h"

namespace ddrl {

// ---- one wavefront per sample: 512 features = 8 per lane -----------------------------------------
// The xor butterfly 32, 16, 8, 4, 2, 1 without the LDS crossbar (ds_bpermute: an address register, a round trip and an s_waitcnt per step, and
// with one wave per SIMD nothing hides it): gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / neighbouring rows of two
// registers in the vector ALU, and DPP row rotations / shifts / quad permutes reach lane ^ 8, ^ 4, ^ 2, ^ 1.  Same partners in the same order as
// `v += __shfl_xor(v, off)`, so the sums are bit-identical to the shuffle form (a + b == b + a in IEEE arithmetic).
template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_lane(float old, float x) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(x), CTRL, 0xF, BANK_MASK, false));
}
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
  {  // ^ 32: r[0] = {v[0..31], v[0..31]}, r[1] = {v[32..63], v[32..63]}
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {  // ^ 16: r[0] = rows {0, 0, 2, 2}, r[1] = rows {1, 1, 3, 3}
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  v = op(v, dpp_lane<0x128>(v, v));  // row_ror:8 = lane ^ 8
  {
    float t = dpp_lane<0x104, 0x5>(v, v);  // row_shl:4: lanes with bit 2 clear (banks 0, 2) read lane + 4
    t = dpp_lane<0x114, 0xA>(t, v);        // row_shr:4: lanes with bit 2 set (banks 1, 3) read lane - 4
    v = op(v, t);
  }
  v = op(v, dpp_lane<0x4E>(v, v));   // quad_perm [2,3,0,1] = lane ^ 2
  v = op(v, dpp_lane<0xB1>(v, v));   // quad_perm [1,0,3,2] = lane ^ 1
  return v;
}
// ---- the same butterfly over MANY values at once ("transposing" reduction) -------------------------------------------------------
// Level by level two registers of partial sums become one: half of the lanes (those whose bit `off` is clear) continue with the first
// value, the other half with the second.  After the levels 32, 16, 8, 4, 2 lane l holds the wave total of value
//   idx = bit5(l) + 2 bit4(l) + 4 bit3(l) + 8 bit2(l) + 16 bit1(l)
// (both lanes of a pair after the plain level 1).  Every value is still summed by ITS xor butterfly 32, 16, 8, 4, 2, 1 -- bit-identical to
// wave_sum -- but 32 values cost 16 + 8 + 4 + 2 + 1 + 1 exchanges instead of 32 x 6.
__device__ __forceinline__ float swap_add32(float a, float b) {  // lanes < 32: a[l] + a[l + 32]; lanes >= 32: b[l - 32] + b[l]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {  // rows 0, 2: a's row pair sums; rows 1, 3: b's
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_add8(float c, float d, bool bit3) {  // bit3 = lane & 8
  const float x = bit3 ? d : c, y = bit3 ? c : d;
  return x + dpp_lane<0x128>(y, y);
}
__device__ __forceinline__ float fold_add4(float c, float d, bool bit2) {
  const float x = bit2 ? d : c, y = bit2 ? c : d;
  float t = dpp_lane<0x104, 0x5>(y, y);  // row_shl:4 / row_shr:4 as in wave_butterfly
  t = dpp_lane<0x114, 0xA>(t, y);
  return x + t;
}
__device__ __forceinline__ float fold_add2(float c, float d, bool bit1) {
  const float x = bit1 ? d : c, y = bit1 ? c : d;
  return x + dpp_lane<0x4E>(y, y);
}
__device__ __forceinline__ float wave_sum(float v) {  // wave64 xor butterfly, result in every lane
  return wave_butterfly(v, [](float a, float b) { return a + b; });
}
// maximum of NON-NEGATIVE magnitudes (|x| maxima of the plane scales): on their bit patterns an unsigned integer maximum is the same
// number, a NaN (0x7FC00000 > every finite pattern) wins and stays visible, and the compiler does not have to canonicalise what comes out of
// a lane exchange before every v_max_f32
__device__ __forceinline__ float wave_max(float v) {
  return wave_butterfly(v, [](float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); });
}
// ---- reduction of per-workgroup partials [nwg][stride] (heads.hip, gheads.hip, gail.hip) ---------------------------------------------
// Element i: eight partial sums (workgroups w = q mod 8, ascending, in double; a ragged tail nwg % 8 goes to partial sum 0), combined in
// a fixed order and rounded once (as the split-K slabs in optim.hip).  A 256-thread workgroup takes RED_OUT = 32 elements and gives each
// of the eight partial sums a thread of its own (thread = 32 q + element): 32 loads per thread, all in flight together (round 6; one
// thread per element walked its 256 loads eight at a time: 43 us for the 3,591 elements of the Pong heads, now 5).  Every thread of the
// workgroup calls; the total comes back in the threads q == 0.
constexpr int RED_OUT = 32;
__device__ __forceinline__ float sum_partials8(const float* __restrict__ part, int64_t stride, int nwg, int i, double (*sh)[RED_OUT]) {
  const int o = threadIdx.x & (RED_OUT - 1), q = threadIdx.x / RED_OUT;
  const int full = nwg - nwg % 8;
  double ps = 0.0;
  int w = q;
  for (; w + 56 < full; w += 64) {  // eight of this thread's workgroups per round: the loads do not wait for the adds
    float x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = part[(int64_t)(w + 8 * t) * stride + i];
#pragma unroll
    for (int t = 0; t < 8; ++t) ps += (double)x[t];
  }
  for (; w < full; w += 8) ps += (double)part[(int64_t)w * stride + i];
  if (q == 0)
    for (int t = full; t < nwg; ++t) ps += (double)part[(int64_t)t * stride + i];
  sh[q][o] = ps;
  __syncthreads();
  if (q != 0) return 0.0f;
  return (float)(((sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o])) + ((sh[4][o] + sh[5][o]) + (sh[6][o] + sh[7][o])));
}
// one scalar (a loss sum) over the workgroups by one wave: four workgroups per lane and a butterfly, in double; the total in every lane
__device__ __forceinline__ double wave_sum_partials(const float* __restrict__ part, int64_t stride, int nwg, int idx) {
  double s = 0.0;
  for (int w = threadIdx.x & 63; w < nwg; w += 64) s += (double)part[(int64_t)w * stride + idx];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
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
