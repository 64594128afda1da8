// Launch sequencing of the encoder forward / backward on the v2 engine kernels
// (conv2.hip, wgrad2.hip, fc2.hip).  Reference call sites being replaced: PPO.forward
// (USTC_lab/nn/ppo.py:72-75) and the autograd backward of PPO.learn (ppo.py:122-123).
#include "engine2.h"
#include "kernels.h"

namespace ddrl {

// largest |dh| per encoder -> Workspace::amax (scale of dh's fp16 planes, engine2.h plane scheme).  dh comes from
// heads_loss, from the GAIL critic's value head on top of it, or from the caller (ddrl_encoder_backward): measured here.
__global__ __launch_bounds__(256) void dh_amax_kernel(const float* __restrict__ dh, int64_t dh_es, int64_t count, float* __restrict__ amax) {
  const int e = blockIdx.y;
  const f4* src = (const f4*)(dh + e * dh_es);
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count / 4; i += (int64_t)gridDim.x * 256) {
    const f4 v = src[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  amax_update(m, amax + amax_idx(AMAX_DH, e));
}
static void launch_backward_amax(const EncCall& c, hipStream_t st) {
  static_assert(AMAX_DZ3 == AMAX_DH + 1 && AMAX_DZ2 == AMAX_DH + 2 && AMAX_DZ1 == AMAX_DH + 3 && AMAX_SLOTS == AMAX_DH + 4,
                "gradient slots are the last four");
  const Workspace& w = *c.ws;
  (void)hipMemsetAsync(w.amax + amax_idx(AMAX_DH, 0), 0, 4 * 2 * sizeof(float), st);  // dz3 / dz2 / dz1 are raised by their producers
  const int64_t count = (int64_t)c.n * FEAT;
  int wgs = (int)((count / 4 + 255) / 256);
  if (wgs > 512) wgs = 512;
  hipLaunchKernelGGL(dh_amax_kernel, dim3(wgs, c.L->NE), dim3(256), 0, st, w.dh, c.max_batch * FEAT, count, w.amax);
}

void launch_encoder_forward(const EncCall& c, bool acting, hipStream_t st) {
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
void launch_encoder_backward(const EncCall& c, float* grads, hipStream_t st) {
  launch_backward_amax(c, st);
#ifndef DDRL_BWD_LAYERWISE
  launch_fc_backward2(c, grads, st, 1);
  launch_conv_dgrad3_2(c, st);
  launch_conv_dgrad2_2(c, st);
  launch_conv_wgrad1_2(c, grads, st);  // no data gradient for conv1: the frames are leaves
  launch_fc_backward2(c, grads, st, 2);
  launch_conv_wgrad3_2(c, grads, st);
  launch_conv_wgrad2_2(c, grads, st);
#else
  launch_fc_backward2(c, grads, st);
  launch_conv_wgrad3_2(c, grads, st);
  launch_conv_dgrad3_2(c, st);
  launch_conv_wgrad2_2(c, grads, st);
  launch_conv_dgrad2_2(c, st);
  launch_conv_wgrad1_2(c, grads, st);
#endif
}

}  // namespace ddrl
