// Launch sequencing of the encoder forward / backward on the v2 engine kernels
// (conv2.hip, wgrad2.hip, fc2.hip).  Reference call sites being replaced: PPO.forward
// (USTC_lab/nn/ppo.py:72-75) and the autograd backward of PPO.learn (ppo.py:122-123).
#include "engine2.h"
#include "kernels.h"

namespace ddrl {

// Per-sample normalisation of the backward (common.h Workspace::gsc).  dh comes from heads_loss, from the GAIL critic's value head on
// top of it, or from the caller (ddrl_encoder_backward).  One wave per (sample, encoder): g_s = 2^floor(log2 max_k |dh[s][k]|)
// (clamped to 2^+-60; 2^-60 for an all-zero row), dh[s][:] *= 1 / g_s in place (exact), gsc[e][s] = g_s.  The whole data-gradient
// chain is linear per sample, so dz3 / dz2 / dz1 come out normalised by the same g_s and a sample whose advantage is 10^6 x below
// the batch's largest keeps the same 22 bits as the largest; only the weight gradients, which SUM over samples, multiply g_s back
// in (wgrad2.hip, fc2.hip).  Slots: AMAX_DH = largest normalised |dh| (in [1, 2) unless clamped), AMAX_GMAX = largest g_s.
__global__ __launch_bounds__(256) void dh_normalise_kernel(float* __restrict__ dh, int64_t dh_es, int n, float* __restrict__ gsc, int64_t gsc_es,
                                                           float* __restrict__ amax) {
  const int e = blockIdx.y, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n) return;  // wave-uniform
  f4* row = (f4*)(dh + e * dh_es + (int64_t)b * FEAT) + lane * 2;
  f4 v0 = row[0], v1 = row[1];
  float m = fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                  fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  int ex = GSC_EXP_MIN;  // floor(log2 m) from the bit pattern (subnormal m clamps to the lower bound)
  if (m > 0.0f && m < 3.0e38f) ex = min(max((int)((__float_as_uint(m) >> 23) & 0xFFu) - 127, GSC_EXP_MIN), GSC_EXP_MAX);
  const float g = __uint_as_float((unsigned)(ex + 127) << 23), gi = __uint_as_float((unsigned)(127 - ex) << 23);
  row[0] = v0 * gi;
  row[1] = v1 * gi;
  if (lane == 0) {
    gsc[e * gsc_es + b] = g;
    const unsigned gb = __float_as_uint(g), mb = __float_as_uint(m * gi);  // non-negative floats order like unsigned integers
    unsigned* sg = (unsigned*)(amax + amax_idx(AMAX_GMAX, e));
    unsigned* sm = (unsigned*)(amax + amax_idx(AMAX_DH, e));
    // look first: after the first few waves almost no atomic is sent (engine2.h amax_update)
    if (gb > __hip_atomic_load(sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sg, gb);
    if (mb > __hip_atomic_load(sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sm, mb);
  }
}
void launch_backward_amax_reset(const EncCall& c, hipStream_t st) {
  static_assert(AMAX_DZ3 == AMAX_DH + 1 && AMAX_DZ2 == AMAX_DH + 2 && AMAX_DZ1 == AMAX_DH + 3 && AMAX_GMAX == AMAX_DH + 4 &&
                    AMAX_SLOTS == AMAX_DH + 5,
                "gradient slots are the last five");
  (void)hipMemsetAsync(c.ws->amax + amax_idx(AMAX_DH, 0), 0, 5 * 2 * sizeof(float), st);  // dz3 / dz2 / dz1 are raised by their producers
}
static void launch_backward_amax(const EncCall& c, hipStream_t st) {
  const Workspace& w = *c.ws;
  launch_backward_amax_reset(c, st);
  ProfRange pr(c.prof, "dh_normalise", st);
  hipLaunchKernelGGL(dh_normalise_kernel, dim3((c.n + 3) / 4, c.L->NE), dim3(256), 0, st, w.dh, c.max_batch * FEAT, c.n, w.gsc, c.max_batch,
                     w.amax);
}

void launch_encoder_forward(const EncCall& c, bool acting, hipStream_t st) {
  if (acting && c.n <= DDRL_ACT_FUSED_MAX) {  // latency-bound: one launch for the three convolutions (act.hip), then the batched dense layer
    launch_act_convs(c, st);
    launch_fc_forward2(c, true, st, true);
    return;
  }
  launch_conv_forward2(c, acting, st);
  launch_fc_forward2(c, acting, st);
}

// Backward of both encoders given dh[e][n][512] (written by heads_loss).  Every weight-gradient
// kernel leaves split-K partial slabs that reduce_partials sums into the grad arena.
// Order: the data-gradient chain first (dense, conv3, conv2: kernels on the bf16 matrix pipe, then conv1's weight
// gradient, also on it), the three fp32-MFMA weight gradients last.  Every buffer a weight gradient reads (dh, dz3, dz2
// and the activations) is still intact then.  The layer-by-layer order (-DDDRL_BWD_LAYERWISE) alternates bf16-pipe and
// fp32-pipe kernels, and each fp32 kernel that follows a bf16 one starts at the lower clock the denser pipe leaves
// behind: 42.07 vs 41.84 ms per PPO iteration on one box.
void launch_encoder_backward(const EncCall& c, float* grads, hipStream_t st, bool dh_normalised) {
  if (!dh_normalised) launch_backward_amax(c, st);  // dh from heads_loss arrives normalised (heads.hip); anyone else's is normalised here
#ifndef DDRL_BWD_LAYERWISE
  launch_fc_backward2(c, grads, st, 1);
  launch_conv_dgrad3_2(c, st);
  launch_conv_dgrad2_2(c, st);
  launch_conv_wgrad1_2(c, grads, st);  // no data gradient for conv1: the frames are leaves
  bucket_done(c, BUCKET_CONV1, st);
  launch_fc_backward2(c, grads, st, 2);
  bucket_done(c, BUCKET_FC, st);       // 95 % of the arena's bytes: its all-reduce runs under the two conv weight gradients below
  launch_conv_wgrad3_2(c, grads, st);
  bucket_done(c, BUCKET_CONV3, st);
  launch_conv_wgrad2_2(c, grads, st);
  bucket_done(c, BUCKET_CONV2, st);
#else
  launch_fc_backward2(c, grads, st);
  bucket_done(c, BUCKET_FC, st);
  launch_conv_wgrad3_2(c, grads, st);
  bucket_done(c, BUCKET_CONV3, st);
  launch_conv_dgrad3_2(c, st);
  launch_conv_wgrad2_2(c, grads, st);
  bucket_done(c, BUCKET_CONV2, st);
  launch_conv_dgrad2_2(c, st);
  launch_conv_wgrad1_2(c, grads, st);
  bucket_done(c, BUCKET_CONV1, st);
#endif
}

}  // namespace ddrl
