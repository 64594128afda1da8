// Launch sequencing of the encoder forward / backward on the v2 engine kernels
// (conv2.hip, wgrad2.hip, fc2.hip).  Reference call sites being replaced: PPO.forward
// (USTC_lab/nn/ppo.py:72-75) and the autograd backward of PPO.learn (ppo.py:122-123).
#include "kernels.h"

namespace ddrl {

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
