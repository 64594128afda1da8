// Internal launcher interface of the generic operators (gconv.hip, glinear.hip, gheads.hip).
#pragma once
#include "common.h"

namespace ddrl {

struct ConvGeom {
  int n, cin, h, w, cout, kh, kw, stride, lgs, pad_h, pad_w, oh, ow;
  int64_t in_sn, out_sn;  // sample strides (floats) of the input / output tensors ([c][h][w] dense inside)
};

// gconv.hip
bool conv_geom_fill(ConvGeom& g);  // derives lgs, oh, ow from the rest; false when unsupported
void conv_pack_sizes(const ConvGeom& g, int64_t out[5]);
void launch_conv_pack(const ConvGeom& g, const float* w, float* wpf, int2* ktf, float* wpd, int2* ktd, int* ptab, hipStream_t st);
void launch_conv_fwd(const ConvGeom& g, const float* in, const float* wpf, const int2* ktf, const float* bias, int act,
                     float* out, hipStream_t st);
void launch_conv_dgrad(const ConvGeom& g, const float* dz, const float* wpd, const int2* ktd, float* din, hipStream_t st);
int conv_wgrad_splits(const ConvGeom& g);
void launch_conv_wgrad(const ConvGeom& g, const float* in, const float* dz, const int* ptab, float* part, float* dw, float* db,
                       hipStream_t st);
void launch_maxpool2_fwd(const float* in, int64_t planes, int H, int W, float* out, hipStream_t st);
void launch_maxpool2_relu_bwd(const float* a, const float* dpool, int64_t planes, int H, int W, float* dz, hipStream_t st);
void launch_maxpool2_fwd_idx(const float* in, int64_t planes, int H, int W, float* out, uint8_t* code, hipStream_t st);
void launch_maxpool2_bwd_idx(const float* dpool, const uint8_t* code, int64_t planes, int H, int W, float* dz, hipStream_t st);

// pconv.hip: forward / data gradient / weight gradient of the heavy nav layers on the 16-bit matrix pipe (two scaled fp16 planes per operand, per-sample
// input scales from the samples' largest magnitudes: engine2.h scale_of_amax); `scales` = n floats of scratch for a pre-pass over the
// input; `given*` = the magnitudes the caller already holds (a producer's out_amax, or launch_sample_amax), nullptr = run the pre-pass;
// `out_amax` (may be null) = n floats the epilogue RAISES to the largest |output| of every sample (zeroed by the caller)
bool conv_has_planes(const ConvGeom& g);
int64_t conv_planes_pack_floats(const ConvGeom& g);  // floats of ONE region (forward or data gradient)
void launch_conv_planes_pack(const ConvGeom& g, const float* w, float* wpf, float* wpd, hipStream_t st);
void launch_conv_planes_fwd(const ConvGeom& g, const float* in, const float* wpf, float* scales, const float* bias, int act, float* out,
                            hipStream_t st);
void launch_conv_planes_dgrad(const ConvGeom& g, const float* dz, const float* wpd, float* scales, float* din, hipStream_t st);
bool conv_planes_has_pool(const ConvGeom& g);     // forward with ReLU + max_pool2d(2) in the epilogue
void launch_sample_amax(const float* x, int64_t sn, int elems, int n, float* amax, hipStream_t st, int accumulate = 0);   // amax[b] = (max with) max |x[b][:]|
void launch_conv_planes_fwd_pool(const ConvGeom& g, const float* in, const float* wpf, float* scales, const float* given, const float* bias,
                                 float* pooled, uint8_t* code, float* out_amax, hipStream_t st);
// gradients of such a layer from d(pooled) + decision bytes (scales / part as launch_conv_planes_dgrad / _wgrad)
void launch_conv_planes_dgrad_pooled(const ConvGeom& g, const float* dpool, const uint8_t* ucode, const float* wpd, float* scales, const float* given,
                                     float* din, float* out_amax, hipStream_t st);
void launch_conv_planes_wgrad_pooled(const ConvGeom& g, const float* in, const float* dpool, const uint8_t* ucode, const float* given_in,
                                     const float* given_dp, float* part, float* dw, float* db, hipStream_t st);
int conv_planes_wgrad_splits(const ConvGeom& g);  // 0 when the layer has no plane kernels
// part: conv_planes_wgrad_splits slabs of cout * cin * kh * kw + cout floats, then 2 n floats (per-sample scales of this launch)
void launch_conv_planes_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st);

// fconv.hip: the 3-channel 7x7 first layer of NavPreNet1D as fp16 plane products (forward + weight gradient; per-sample / per-stage
// scales are found inside the kernels: no scratch)
bool conv_has_first(const ConvGeom& g);
int64_t conv_first_pack_floats(const ConvGeom& g);
void launch_conv_first_pack(const ConvGeom& g, const float* w, float* region, hipStream_t st);
void launch_conv_first_fwd(const ConvGeom& g, const float* in, const float* region, const float* bias, int act, float* out, hipStream_t st);
void launch_conv_first_fwd_pool(const ConvGeom& g, const float* in, const float* region, const float* bias, float* pooled, uint8_t* code,
                                float* out_amax, hipStream_t st);  // conv + ReLU + max_pool2d(2) in one launch; out_amax as pconv.hip's
void launch_conv_first_wgrad_pooled(const ConvGeom& g, const float* in, const float* dpool, const uint8_t* ucode, float* part, float* dw, float* db,
                                    hipStream_t st);  // weight gradient from d(pooled) + decision bytes
int conv_first_wgrad_splits(const ConvGeom& g);  // 0 when the layer is not this one
void launch_conv_first_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st);

// c1d.hip: the two stride-2 Conv1d layers of NavPreNet1D's laser branch as fp32 vector kernels (weights through the scalar cache)
bool conv_has_c1d(const ConvGeom& g);           // forward
bool conv_has_c1d_backward(const ConvGeom& g);  // data + weight gradient (the 32 -> 32 layer)
int64_t conv_c1d_pack_floats(const ConvGeom& g);  // floats of EACH of the two regions
void launch_conv_c1d_pack(const ConvGeom& g, const float* w, float* wt, float* wd, hipStream_t st);
void launch_conv_c1d_fwd(const ConvGeom& g, const float* in, const float* wt, const float* bias, int act, float* out, float* out_amax, hipStream_t st);
void launch_conv_c1d_dgrad(const ConvGeom& g, const float* dz, const float* wd, float* din, hipStream_t st);
int conv_c1d_wgrad_splits(const ConvGeom& g);  // 0 when the layer has no such kernel
void launch_conv_c1d_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st);

// glinear.hip
void launch_reduce_slabs(const float* part, int nsplit, int64_t slab_stride, int64_t count, float* dst, hipStream_t st);
void launch_reduce_slabs2(const float* part, int nsplit, int64_t slab_stride, int64_t c0, float* dst0, int64_t c1, float* dst1, hipStream_t st);
void launch_relu_mask(float* d, int64_t ld_d, const float* act, int64_t ld_act, int64_t n, int width, hipStream_t st);
void launch_accumulate(float* dst, const float* src, int64_t count, hipStream_t st);
void launch_linear_pack(const float* w, int K, int N, float* wt, float* wn, hipStream_t st);
int linear_fwd_splits(int n, int K, int N);
void launch_linear_fwd(const float* in, int64_t ld_in, const float* wt, const float* bias, float* out, int64_t ld_out, int n,
                       int K, int N, int act, float* part, const float* in_scales, hipStream_t st);
void launch_linear_dgrad(const float* dout, int64_t ld_dout, const float* wn, const float* mask_src, int64_t ld_mask,
                         float* din, int64_t ld_din, int n, int K, int N, float* ws, const float* dout_scales, float* din_amax, int amax_lo,
                         int amax_hi, hipStream_t st);
int linear_wgrad_splits(int n, int K, int N);
void launch_linear_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* part, int n, int K, int N,
                         float* dw, float* db, const float* in_scales, const float* dout_scales, hipStream_t st);

void launch_linear_finish(const float* part, int nsplit, int n, int N, const float* bias, int act, float* out, int64_t ld_out, hipStream_t st);

// plin.hip: the same three operators on the 16-bit matrix pipe (two scaled fp16 planes per operand, per-row activation scales) for layers
// of K >= 128, N >= 64 and launches of n >= 128 rows; packed plane regions follow the f32 layouts inside wt / wn
bool linear_has_planes(int K, int N);
bool linear_uses_planes(int n, int K, int N);
int64_t linear_planes_fwd_floats(int K, int N);
int64_t linear_planes_dgrad_floats(int K, int N);
void launch_linear_planes_pack(const float* w, int K, int N, float* pf, float* pd, hipStream_t st);
int linear_planes_fwd_splits(int n, int K, int N);
// ws: n floats rounded up to 64 (row magnitudes of a pre-pass), then linear_planes_fwd_splits * n * N partials
void launch_row_amax(const float* x, int64_t ld, int width, int n, float* amax, int accumulate, hipStream_t st);   // amax[b] = (max with) max |x[b][:width]|
// given*: per-row magnitudes the caller already holds for that tensor (a producer's out_amax, launch_row_amax), or nullptr for a pre-pass into the scratch
void launch_linear_planes_fwd(const float* in, int64_t ld_in, const float* pf, const float* bias, float* out, int64_t ld_out, int n, int K,
                              int N, int act, float* ws, const float* given, hipStream_t st);
// ws: n floats
// din_amax (may be null): n floats RAISED to the largest |din| of every row over the columns [amax_lo, amax_hi) (zeroed by the caller)
void launch_linear_planes_dgrad(const float* dout, int64_t ld_dout, const float* pd, const float* mask_src, int64_t ld_mask, float* din,
                                int64_t ld_din, int n, int K, int N, float* ws, const float* given, float* din_amax, int amax_lo, int amax_hi,
                                hipStream_t st);
int linear_planes_wgrad_splits(int n, int K, int N);
// part: linear_planes_wgrad_splits slabs of N * K + N floats, then 2 n floats
void launch_linear_planes_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* part, int n, int K, int N,
                                float* dw, float* db, const float* given_in, const float* given_dout, hipStream_t st);

// gheads.hip: Gaussian actor + critic heads; offsets into the caller's flat parameter arena
struct GaussLayout {
  int D;       // action dims (<= 8)
  int shared;  // 1: total_loss is differentiated (shared prenet), both heads feed dh_actor
  int64_t actor_w, actor_b, log_std, critic_w, critic_b, n_params;
};
int64_t gauss_hpart_stride(int D);
void launch_gauss_act(const GaussLayout& L, const float* params, const float* h_actor, const float* h_critic, int n,
                      const float* act_in, uint64_t seed, uint64_t stream_id, float* mu_out, float* value, float* action_out,
                      float* logp_out, hipStream_t st);
void launch_gauss_loss(const GaussLayout& L, const ddrl_config& cfg, const float* params, const float* h_actor,
                       const float* h_critic, int n, const float* actions, const float* old_logps, const float* advs,
                       const float* rets, float inv_b, float* dh_actor, float* dh_critic, float* dmu, float* dvalue,
                       float* hpart, float* grads, hipStream_t st);

}  // namespace ddrl
