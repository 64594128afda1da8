// Internal launcher interface between the translation units of libddrl_hip.so.
#pragma once
#include "common.h"

namespace ddrl {

// per-kernel HIP-event ranges (diagnostic; null when profiling is off)
struct Profiler {
  virtual void begin(const char* name, hipStream_t st) = 0;
  virtual void end(hipStream_t st) = 0;
  virtual ~Profiler() {}
};
struct ProfRange {
  Profiler* p;
  hipStream_t st;
  ProfRange(Profiler* p_, const char* name, hipStream_t s) : p(p_), st(s) {
    if (p) p->begin(name, s);
  }
  ~ProfRange() {
    if (p) p->end(st);
  }
};

// Gradient buckets of the data-parallel all-reduce, in the order the backward COMPLETES them ("bucket by layer so
// that the all-reduce of the early buckets overlaps the rest of the backward"): heads + loss tail (after heads_loss), conv1 (its
// weight gradient runs right after the data-gradient chain), dense, conv3, conv2.  launch_encoder_backward records bucket_ev[b] on
// the compute stream once bucket b's slabs are reduced into the arena; a communication stream waits for it (api.hip).
constexpr int GRAD_BUCKETS = 5, BUCKET_HEADS = 0, BUCKET_CONV1 = 1, BUCKET_FC = 2, BUCKET_CONV3 = 3, BUCKET_CONV2 = 4;

struct EncCall {
  Profiler* prof;
  const Workspace* ws;
  const ParamLayout* L;
  const Splits* splits;
  const float* params;
  const uint8_t* frames;  // [n][4][84][84]
  int n;
  int64_t max_batch;
  hipEvent_t* bucket_ev = nullptr;  // [GRAD_BUCKETS] or null (single rank: nothing to overlap)
  bool keep_acts = false;           // acting launches: also store a1 / a2 (ddrl_debug_keep_activations; the fused kernel of act.hip keeps them on chip)
};
inline void bucket_done(const EncCall& c, int b, hipStream_t st) {
  if (c.bucket_ev) (void)hipEventRecord(c.bucket_ev[b], st);
}

// encoder.hip
void launch_encoder_forward(const EncCall& c, bool acting, hipStream_t st);
void launch_encoder_backward(const EncCall& c, float* grads, hipStream_t st, bool dh_normalised = false);
void launch_backward_amax_reset(const EncCall& c, hipStream_t st);  // zeroes the gradient slots of Workspace::amax

// fc2.hip (v2 engine)
// Split-K factor of the FC forward for a batch of n samples (1 = plain; >1 only on the acting
// path, where heads_act sums the partials).  7 k-blocks of 32 per split.
inline int fc_forward_splits(int n) { return n <= 1024 ? DDRL_FC_ACT_SPLITS : 1; }
void launch_fc_forward2(const EncCall& c, bool allow_split, hipStream_t st, bool per_sample_max = false);
void launch_fc_backward2(const EncCall& c, float* grads, hipStream_t st, int part = 0);

// conv2.hip (v2 engine)
void launch_conv_forward2(const EncCall& c, bool acting, hipStream_t st);

// act.hip: conv1 + conv2 + conv3 of an acting forward in one launch, one workgroup per (sample, encoder); leaves a3 and every sample's
// largest |a3| (Workspace::actmax), which the dense layer's split launch takes its plane scale from
// (DDRL_ACT_FUSED_MAX: common.h, next to the carve of Workspace::actmax)
void launch_act_convs(const EncCall& c, hipStream_t st);
void launch_conv_dgrad3_2(const EncCall& c, hipStream_t st);
void launch_conv_dgrad2_2(const EncCall& c, hipStream_t st);

// wgrad2.hip (v2 engine)
void launch_conv_wgrad3_2(const EncCall& c, float* grads, hipStream_t st);
void launch_conv_wgrad2_2(const EncCall& c, float* grads, hipStream_t st);
void launch_conv_wgrad1_2(const EncCall& c, float* grads, hipStream_t st);

// optim.hip
void launch_pack_weights(const Workspace& w, const ParamLayout& L, const float* params, hipStream_t st);
void launch_reduce_partials(const float* part, int nsplit, int64_t count, int ne, float* grads, int64_t off0,
                            int64_t off1, hipStream_t st);
void launch_clip_adam(const ddrl_config& cfg, const ParamLayout& L, const Workspace& w, float* params,
                      float* grads, float* m, float* v, int64_t step, hipStream_t st);
void launch_episode_returns(const float* rewards, const uint8_t* dones, int T, int N, float* rsum, float* rep, float* trace,
                            int* finished, hipStream_t st);
void launch_gae(const float* values, const float* rewards, const uint8_t* dones, int T, int N,
                float gamma, float landa, float* adv, float* ret, hipStream_t st);
void launch_fill_lut(float* lut, hipStream_t st);

// heads.hip
struct HeadsCall {
  const Workspace* ws;
  const ParamLayout* L;
  const ddrl_config* cfg;
  const float* params;
  int n;
  int64_t max_batch;
  // generic callers (ddrl_op_heads_*): explicit actor->critic feature / gradient strides and no
  // split-K FC partials to fold in; the Atari context leaves these at their defaults
  // (a difference of two independent allocations: ANY value, negative included, is a valid stride -- the "unset" mark is a
  // separate sentinel; a plain "< 0" test once sent every net whose critic buffer happened to lie below its actor buffer to
  // h + max_batch * 512: wrong values, out-of-bounds reads)
  static constexpr int64_t ES_UNSET = INT64_MIN;
  int64_t h_es = ES_UNSET, dh_es = ES_UNSET;
  bool plain_features = false;
  // heads_loss also normalises dh per sample (Workspace::gsc, amax slots DH / GMAX, which the caller has zeroed): the Atari context's
  // ddrl_ppo_iter; launch_encoder_backward is then told to skip its stand-alone dh_normalise_kernel
  bool normalise_dh = false;
};
void launch_heads_act(const HeadsCall& c, const float* act_in, uint64_t seed, uint64_t stream_id,
                      float* probs, float* value, float* action_out, float* logp_out, hipStream_t st);
void launch_heads_loss(const HeadsCall& c, const float* actions, const float* old_logps,
                       const float* advs, const float* rets, float inv_bglobal, float* grads,
                       hipStream_t st);
void launch_categorical_sample(const float* probs, int n, int A, uint64_t seed, uint64_t stream_id, float* action,
                               float* logp, hipStream_t st);
void launch_categorical_stats(const float* probs, int n, int A, float* p_hat, float* logits,
                              float* entropy, hipStream_t st);

}  // namespace ddrl
