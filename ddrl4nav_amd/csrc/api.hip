// extern "C" boundary of libddrl_hip.so (see include/ddrl.h).  Host-side only: argument
// validation, workspace carving, kernel sequencing, the pinned-host ring and HIP-event timers.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"

using namespace ddrl;

struct ProfEntry {
  std::string name;
  hipEvent_t a, b;
};

struct ddrl_ctx : public ddrl::Profiler {
  ddrl_config cfg;
  ParamLayout L;
  Workspace ws;
  Splits splits;
  float *params, *grads, *m, *v;
  int64_t step;
  bool dirty;
  int last_n;
  bool profile;
  bool profile_acting;  // ddrl_profile_enable(on = 1): also time the (small, latency-bound) ddrl_forward launches
  bool keep_acts = false;   // ddrl_debug_keep_activations: ddrl_forward also stores a1 / a2 (its fused kernel keeps them on chip)
  bool acts_stored = true;  // false after a ddrl_forward that left a1 / a2 on chip: ddrl_debug_buffer(0 / 1) refuses
  hipEvent_t bucket_ev[GRAD_BUCKETS];
  hipEvent_t comm_done;
  hipEvent_t call_ev;  // recorded on the compute stream when an overlapped reduction is requested (see ddrl_grad_allreduce_overlapped)
  bool buckets;  // ddrl_grad_buckets_enable: record the bucket events in every ddrl_ppo_iter
  bool bucket_events_fresh = false;  // a ddrl_ppo_iter has recorded all bucket events since the last overlapped reduction
  std::vector<ProfEntry> prof_pending;
  std::vector<std::string> prof_names;
  std::vector<double> prof_ms;
  std::vector<int> prof_calls;
  ProfEntry cur;
  void begin(const char* name, hipStream_t st) override {
    cur.name = name;
    hipEventCreate(&cur.a);
    hipEventCreate(&cur.b);
    hipEventRecord(cur.a, st);
  }
  void end(hipStream_t st) override {
    hipEventRecord(cur.b, st);
    prof_pending.push_back(cur);
  }
};

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return DDRL_ERR_HIP; \
  } while (0)

static int32_t check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DDRL_OK : DDRL_ERR_HIP;
}


extern "C" {

int32_t ddrl_abi_version(void) { return DDRL_ABI_VERSION; }

const char* ddrl_status_string(int32_t s) {
  switch (s) {
    case DDRL_OK: return "ok";
    case DDRL_ERR_INVALID_ARG: return "invalid argument";
    case DDRL_ERR_UNSUPPORTED: return "unsupported configuration";
    case DDRL_ERR_WORKSPACE: return "workspace too small";
    case DDRL_ERR_HIP: return "HIP runtime error";
    case DDRL_ERR_NO_DEVICE: return "no HIP device";
    case DDRL_ERR_TIMEOUT: return "timeout";
    case DDRL_ERR_NO_MEMORY: return "out of host memory";
    default: return "unknown status";
  }
}

int32_t ddrl_config_default(ddrl_config* c) {
  if (!c) return DDRL_ERR_INVALID_ARG;
  c->n_actions = 6;
  c->in_channels = 4;
  c->max_batch = 1024;
  c->share_cnn_net = 0;
  c->clip_grad = 1;
  c->clip_grad_norm = 0.5f;
  c->actor_lr = 5e-5f;
  c->critic_lr = 1e-3f;
  c->adam_beta1 = 0.9f;
  c->adam_beta2 = 0.999f;
  c->adam_eps = 1e-8f;
  c->ppo_clip = 0.2f;
  c->dual_clip = 3.0f;
  c->v_loss_theta = 1.0f;
  c->ent_loss_theta = 0.05f;
  c->learning_rate = 2e-4f;
  c->smooth_l1_loss = 0;
  return DDRL_OK;
}

static int32_t validate(const ddrl_config* c) {
  if (!c) return DDRL_ERR_INVALID_ARG;
  if (c->max_batch < 1) return DDRL_ERR_INVALID_ARG;
  if (c->n_actions < 2 || c->n_actions > 18) return DDRL_ERR_UNSUPPORTED;  // heads kernels: A <= 18 (full Atari set)
  if (c->in_channels < 1 || c->in_channels > 4) return DDRL_ERR_UNSUPPORTED;  // stacked frames: conv1's kernels give each of their four waves one channel
  if (c->share_cnn_net != 0 && c->share_cnn_net != 1) return DDRL_ERR_INVALID_ARG;
  // 32-bit element indexing inside one encoder's activation tensor
  // the kernels address a1 / da1 with 32-bit BYTE offsets from wave-uniform bases (max_batch <= 83,886)
  if ((int64_t)c->max_batch * 32 * 400 * 4 >= (int64_t)1 << 32) return DDRL_ERR_UNSUPPORTED;
  return DDRL_OK;
}

int32_t ddrl_param_count(const ddrl_config* c, int64_t* n_params, int64_t* n_actor) {
  int32_t s = validate(c);
  if (s != DDRL_OK) return s;
  ParamLayout L = make_layout(c->n_actions, c->in_channels, c->share_cnn_net != 0);
  if (n_params) *n_params = L.n_params;
  if (n_actor) *n_actor = L.n_actor;
  return DDRL_OK;
}

int32_t ddrl_workspace_bytes(const ddrl_config* c, int64_t* bytes) {
  int32_t s = validate(c);
  if (s != DDRL_OK) return s;
  if (!bytes) return DDRL_ERR_INVALID_ARG;
  Workspace w;
  *bytes = carve(w, *c, nullptr);
  return DDRL_OK;
}

int32_t ddrl_ctx_create(const ddrl_config* c, float* params, float* grads, float* m, float* v, void* workspace,
                        int64_t workspace_bytes, ddrl_ctx** out) {
  int32_t s = validate(c);
  if (s != DDRL_OK) return s;
  if (!params || !grads || !workspace || !out) return DDRL_ERR_INVALID_ARG;  // m / v: NULL for encoder-only contexts
  if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)grads & 15)) return DDRL_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DDRL_ERR_NO_DEVICE;
  ddrl_ctx* ctx = new (std::nothrow) ddrl_ctx();
  if (!ctx) return DDRL_ERR_NO_MEMORY;
  ctx->cfg = *c;
  ctx->L = make_layout(c->n_actions, c->in_channels, c->share_cnn_net != 0);
  const int64_t need = carve(ctx->ws, *c, workspace);
  if (need > workspace_bytes) {
    delete ctx;
    return DDRL_ERR_WORKSPACE;
  }
  ctx->splits = choose_splits(c->max_batch, ctx->L.NE);
  ctx->params = params;
  ctx->grads = grads;
  ctx->m = m;
  ctx->v = v;
  ctx->step = 0;
  ctx->dirty = true;
  ctx->last_n = 0;
  ctx->profile = false;
  ctx->profile_acting = false;
  ctx->buckets = false;
  *out = ctx;
  return DDRL_OK;
}

int32_t ddrl_ctx_destroy(ddrl_ctx* ctx) {
  if (!ctx) return DDRL_ERR_INVALID_ARG;
  for (auto& e : ctx->prof_pending) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  if (ctx->buckets) {
    for (int b = 0; b < GRAD_BUCKETS; ++b) hipEventDestroy(ctx->bucket_ev[b]);
    hipEventDestroy(ctx->comm_done);
    hipEventDestroy(ctx->call_ev);
  }

  delete ctx;
  return DDRL_OK;
}

int32_t ddrl_params_changed(ddrl_ctx* ctx) {
  if (!ctx) return DDRL_ERR_INVALID_ARG;
  ctx->dirty = true;
  return DDRL_OK;
}

int32_t ddrl_get_step(const ddrl_ctx* ctx, int64_t* step) {
  if (!ctx || !step) return DDRL_ERR_INVALID_ARG;
  *step = ctx->step;
  return DDRL_OK;
}
int32_t ddrl_set_step(ddrl_ctx* ctx, int64_t step) {
  if (!ctx || step < 0) return DDRL_ERR_INVALID_ARG;
  ctx->step = step;
  return DDRL_OK;
}

// the activation slots of Workspace::amax start every forward at zero: conv_fwd1_planes_kernel, the first launch of every forward,
// zeroes them itself (the conv epilogues then raise them); the gradient slots are reset by launch_encoder_backward
static void amax_begin(ddrl_ctx*, hipStream_t) {}

static void ensure_packed(ddrl_ctx* ctx, hipStream_t st) {
  if (!ctx->dirty) return;
  ProfRange ps(ctx->profile ? ctx : nullptr, "pack_weights", st);
  launch_pack_weights(ctx->ws, ctx->L, ctx->params, st);
  ctx->dirty = false;
}

int32_t ddrl_forward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, const float* act_in, uint64_t seed,
                     uint64_t stream_id, float* probs, float* value, float* action_out, float* logp_out, void* stream) {
  if (!ctx || !frames || !value) return DDRL_ERR_INVALID_ARG;
  if (n < 1 || n > ctx->cfg.max_batch) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  ensure_packed(ctx, st);
  amax_begin(ctx, st);
  // HIP events around launches of tens of microseconds cost about as much as the launches (an event
  // record drains the queue): the acting path is only timed when asked for explicitly (on = 1)
  Profiler* prof = ctx->profile && ctx->profile_acting ? ctx : nullptr;
  EncCall ec{prof, &ctx->ws, &ctx->L, &ctx->splits, ctx->params, frames, n, ctx->cfg.max_batch};
  ec.keep_acts = ctx->keep_acts;
  ctx->acts_stored = ctx->keep_acts || n > DDRL_ACT_FUSED_MAX;
  launch_encoder_forward(ec, true, st);
  HeadsCall hc{&ctx->ws, &ctx->L, &ctx->cfg, ctx->params, n, ctx->cfg.max_batch};
  {
    ProfRange ps(prof, "heads_act", st);
    launch_heads_act(hc, act_in, seed, stream_id, probs, value, action_out, logp_out, st);
  }
  ctx->last_n = n;
  return check_launch();
}

int32_t ddrl_categorical_stats(const float* probs, int32_t n, int32_t A, float* p_hat, float* logits, float* entropy,
                               void* stream) {
  if (!probs || n < 1 || A < 1) return DDRL_ERR_INVALID_ARG;
  launch_categorical_stats(probs, n, A, p_hat, logits, entropy, (hipStream_t)stream);
  return check_launch();
}

int32_t ddrl_categorical_sample(const float* probs, int32_t n, int32_t A, uint64_t seed, uint64_t stream_id,
                                float* action_out, float* logp_out, void* stream) {
  if (!probs || !action_out || n < 1 || A < 1) return DDRL_ERR_INVALID_ARG;
  launch_categorical_sample(probs, n, A, seed, stream_id, action_out, logp_out, (hipStream_t)stream);
  return check_launch();
}

int32_t ddrl_last_features(ddrl_ctx* ctx, int32_t n, float* h_actor, float* h_critic, void* stream) {
  if (!ctx || n < 1 || n > ctx->last_n) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)n * FEAT * sizeof(float);
  if (h_actor) HIP_TRY(hipMemcpyAsync(h_actor, ctx->ws.h, bytes, hipMemcpyDeviceToDevice, st));
  if (h_critic)
    HIP_TRY(hipMemcpyAsync(h_critic, ctx->ws.h + (ctx->L.NE == 2 ? (int64_t)ctx->cfg.max_batch * FEAT : 0), bytes,
                           hipMemcpyDeviceToDevice, st));
  return DDRL_OK;
}

int32_t ddrl_gae(const float* values, const float* rewards, const uint8_t* dones, int32_t T, int32_t N, float gamma,
                 float landa, float* adv, float* ret, void* stream) {
  if (!values || !rewards || !dones || !adv || !ret || T < 0 || N < 1) return DDRL_ERR_INVALID_ARG;
  if (T == 0) return DDRL_OK;  // len(experiences) <= 1 -> nothing to do (agent.py:125-126)
  launch_gae(values, rewards, dones, T, N, gamma, landa, adv, ret, (hipStream_t)stream);
  return check_launch();
}

int32_t ddrl_episode_returns(const float* rewards, const uint8_t* dones, int32_t T, int32_t N, float* rewards_sum,
                             float* rewards_episode, float* trace, int32_t* episodes_finished, void* stream) {
  if (!rewards || !dones || !rewards_sum || !rewards_episode || T < 0 || N < 1) return DDRL_ERR_INVALID_ARG;
  if (T == 0) return DDRL_OK;
  launch_episode_returns(rewards, dones, T, N, rewards_sum, rewards_episode, trace, episodes_finished, (hipStream_t)stream);
  return check_launch();
}

int32_t ddrl_ppo_iter(ddrl_ctx* ctx, const uint8_t* frames, const float* actions, const float* old_logps,
                      const float* advs, const float* rets, int32_t B, int64_t B_global, void* stream) {
  if (!ctx || !frames || !actions || !old_logps || !advs || !rets) return DDRL_ERR_INVALID_ARG;
  if (B < 1 || B > ctx->cfg.max_batch || B_global < B) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  ensure_packed(ctx, st);
  amax_begin(ctx, st);
  EncCall ec{ctx->profile ? ctx : nullptr, &ctx->ws, &ctx->L, &ctx->splits, ctx->params, frames, B, ctx->cfg.max_batch};
  if (ctx->buckets) ec.bucket_ev = ctx->bucket_ev;
  ctx->acts_stored = true;
  launch_encoder_forward(ec, false, st);
  HeadsCall hc{&ctx->ws, &ctx->L, &ctx->cfg, ctx->params, B, ctx->cfg.max_batch};
  hc.normalise_dh = true;  // dh leaves heads_loss already normalised per sample (268 MB less to re-read and re-write per iteration)
  launch_backward_amax_reset(ec, st);
  {
    ProfRange ps(ctx->profile ? ctx : nullptr, "heads_loss", st);
    launch_heads_loss(hc, actions, old_logps, advs, rets, (float)(1.0 / (double)B_global), ctx->grads, st);
  }
  bucket_done(ec, BUCKET_HEADS, st);  // head-layer gradients and the three loss shares of the tail are final
  launch_encoder_backward(ec, ctx->grads, st, true);
  ctx->last_n = B;
  ctx->bucket_events_fresh = ctx->buckets;
  return check_launch();
}

int32_t ddrl_grad_allreduce(ddrl_ctx* ctx, ddrl_comm* comm, void* stream) {
  if (!ctx || !comm) return DDRL_ERR_INVALID_ARG;
  return ddrl_allreduce_f32(comm, ctx->grads, ctx->L.n_params + DDRL_STATS_FLOATS, stream);
}

// ---- layer buckets of the gradient all-reduce ----------------------------------------------------------
// ranges (offset, count in floats) of bucket b inside the grad arena: one per encoder for the encoder layers; the head layers are
// one contiguous run (actor_linear, critic_linear) and carry the DDRL_STATS_FLOATS tail as their second range
static int bucket_ranges(const ddrl_ctx* ctx, int b, int64_t (&off)[2], int64_t (&cnt)[2]) {
  const ParamLayout& L = ctx->L;
  const EncLayout& E = L.enc;
  const int64_t C = ctx->cfg.in_channels;
  if (b == BUCKET_HEADS) {
    off[0] = L.actor_w;
    cnt[0] = (L.critic_b + 1) - L.actor_w;
    off[1] = L.n_params;
    cnt[1] = DDRL_STATS_FLOATS;
    return 2;
  }
  int64_t o, n;
  switch (b) {
    case BUCKET_CONV1: o = E.c1w; n = (int64_t)C1_OC * C * 64 + C1_OC; break;
    case BUCKET_CONV2: o = E.c2w; n = (int64_t)C2_OC * C2_K + C2_OC; break;
    case BUCKET_CONV3: o = E.c3w; n = (int64_t)C3_OC * C3_K + C3_OC; break;
    case BUCKET_FC: o = E.lw; n = (int64_t)FEAT * FLAT + FEAT; break;
    default: return 0;
  }
  for (int e = 0; e < L.NE; ++e) {
    off[e] = L.enc_base[e] + o;
    cnt[e] = n;
  }
  return L.NE;
}

int32_t ddrl_grad_buckets_enable(ddrl_ctx* ctx) {
  if (!ctx) return DDRL_ERR_INVALID_ARG;
  if (ctx->buckets) return DDRL_OK;
  for (int b = 0; b < GRAD_BUCKETS; ++b) HIP_TRY(hipEventCreateWithFlags(&ctx->bucket_ev[b], hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ctx->call_ev, hipEventDisableTiming));
  ctx->buckets = true;
  ctx->bucket_events_fresh = false;
  return DDRL_OK;
}

int32_t ddrl_grad_bucket_count(const ddrl_ctx* ctx, int32_t* n) {
  if (!ctx || !n) return DDRL_ERR_INVALID_ARG;
  *n = GRAD_BUCKETS;
  return DDRL_OK;
}

int32_t ddrl_grad_bucket_info(const ddrl_ctx* ctx, int32_t b, int64_t* offsets2, int64_t* counts2, int32_t* n_ranges) {
  if (!ctx || !offsets2 || !counts2 || !n_ranges || b < 0 || b >= GRAD_BUCKETS) return DDRL_ERR_INVALID_ARG;
  int64_t off[2] = {0, 0}, cnt[2] = {0, 0};
  *n_ranges = bucket_ranges(ctx, b, off, cnt);
  offsets2[0] = off[0]; offsets2[1] = off[1];
  counts2[0] = cnt[0]; counts2[1] = cnt[1];
  return DDRL_OK;
}

int32_t ddrl_grad_bucket_wait(ddrl_ctx* ctx, int32_t b, void* stream) {
  if (!ctx || !ctx->buckets || b < 0 || b >= GRAD_BUCKETS) return DDRL_ERR_INVALID_ARG;
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ctx->bucket_ev[b], 0));
  return DDRL_OK;
}

int32_t ddrl_grad_buckets_begin(ddrl_ctx* ctx, void* comm_stream, void* compute_stream, int32_t* fresh) {
  if (!ctx || !ctx->buckets || comm_stream == compute_stream) return DDRL_ERR_INVALID_ARG;
  // Whatever the compute stream holds AT THE CALL is ordered before the reduction: with fresh bucket events (a ddrl_ppo_iter
  // directly before) the communication stream waits for this event only in front of the LAST bucket (the earlier buckets keep
  // their overlap); with stale or never-recorded events (gradients from another producer, accumulation, a retried iteration --
  // hipStreamWaitEvent on such an event returns at once) it waits for it in front of the FIRST bucket, i.e. no overlap, no race.
  HIP_TRY(hipEventRecord(ctx->call_ev, (hipStream_t)compute_stream));
  const bool f = ctx->bucket_events_fresh;
  ctx->bucket_events_fresh = false;
  if (!f) HIP_TRY(hipStreamWaitEvent((hipStream_t)comm_stream, ctx->call_ev, 0));
  if (fresh) *fresh = f ? 1 : 0;
  return DDRL_OK;
}

int32_t ddrl_grad_bucket_wait_last(ddrl_ctx* ctx, void* comm_stream) {
  if (!ctx || !ctx->buckets) return DDRL_ERR_INVALID_ARG;
  HIP_TRY(hipStreamWaitEvent((hipStream_t)comm_stream, ctx->call_ev, 0));
  return DDRL_OK;
}

int32_t ddrl_grad_allreduce_overlapped(ddrl_ctx* ctx, ddrl_comm* comm, void* comm_stream, void* compute_stream) {
  if (!ctx || !comm || !ctx->buckets || comm_stream == compute_stream) return DDRL_ERR_INVALID_ARG;
  hipStream_t cs = (hipStream_t)comm_stream;
  int32_t fresh = 0;
  const int32_t sb = ddrl_grad_buckets_begin(ctx, comm_stream, compute_stream, &fresh);
  if (sb != DDRL_OK) return sb;
  for (int b = 0; b < GRAD_BUCKETS; ++b) {  // bucket order = completion order of the last ddrl_ppo_iter
    if (fresh) HIP_TRY(hipStreamWaitEvent(cs, ctx->bucket_ev[b], 0));
    if (b == GRAD_BUCKETS - 1) HIP_TRY(hipStreamWaitEvent(cs, ctx->call_ev, 0));
    int64_t off[2], cnt[2];
    const int nr = bucket_ranges(ctx, b, off, cnt);
    for (int r = 0; r < nr; ++r) {
      const int32_t s = ddrl_allreduce_f32(comm, ctx->grads + off[r], cnt[r], comm_stream);
      if (s != DDRL_OK) return s;
    }
  }
  HIP_TRY(hipEventRecord(ctx->comm_done, cs));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)compute_stream, ctx->comm_done, 0));  // ddrl_clip_adam_step sees the reduced arena
  return DDRL_OK;
}

int32_t ddrl_params_broadcast(ddrl_ctx* ctx, ddrl_comm* comm, int32_t root, void* stream) {
  if (!ctx || !comm) return DDRL_ERR_INVALID_ARG;
  int32_t s = ddrl_broadcast_f32(comm, ctx->params, ctx->L.n_params, root, stream);
  ctx->dirty = true;
  return s;
}

int32_t ddrl_encoder_forward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, void* stream) {
  if (!ctx || !frames || n < 1 || n > ctx->cfg.max_batch || ctx->L.NE != 1) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  ensure_packed(ctx, st);
  amax_begin(ctx, st);
  EncCall ec{ctx->profile ? ctx : nullptr, &ctx->ws, &ctx->L, &ctx->splits, ctx->params, frames, n, ctx->cfg.max_batch};
  ctx->acts_stored = true;
  launch_encoder_forward(ec, false, st);  // complete features (no split-K partials left for a head kernel to sum)
  ctx->last_n = n;
  return check_launch();
}

int32_t ddrl_encoder_backward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, void* stream) {
  if (!ctx || !frames || n < 1 || n > ctx->cfg.max_batch || ctx->L.NE != 1) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  EncCall ec{ctx->profile ? ctx : nullptr, &ctx->ws, &ctx->L, &ctx->splits, ctx->params, frames, n, ctx->cfg.max_batch};
  launch_encoder_backward(ec, ctx->grads, st);
  return check_launch();
}

int32_t ddrl_encoder_buffers(ddrl_ctx* ctx, float** h, float** dh) {
  if (!ctx || !h || !dh) return DDRL_ERR_INVALID_ARG;
  *h = ctx->ws.h;
  *dh = ctx->ws.dh;
  return DDRL_OK;
}

int32_t ddrl_clip_adam_step(ddrl_ctx* ctx, void* stream) {
  if (!ctx || !ctx->m || !ctx->v) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  ctx->step += 1;
  {
    ProfRange ps(ctx->profile ? ctx : nullptr, "clip_adam", st);
    launch_clip_adam(ctx->cfg, ctx->L, ctx->ws, ctx->params, ctx->grads, ctx->m, ctx->v, ctx->step, st);
  }
  ctx->dirty = true;
  return check_launch();
}

int32_t ddrl_u8_table(float* out256, void* stream) {
  if (!out256) return DDRL_ERR_INVALID_ARG;
  launch_fill_lut(out256, (hipStream_t)stream);
  return check_launch();
}

// Diagnostic views into the workspace (tests): which = 0 a1,1 a2,2 a3,3 h,4 dz1,5 dz2,6 dz3,7 dh,
// 8 dlogits, 9 dvalue, 10 / 11 / 12 the sign masks m1 / m2 / m3, 13 the per-sample scales of 4..7.  Returns the device pointer and the encoder stride in floats (words).
int32_t ddrl_debug_buffer(ddrl_ctx* ctx, int32_t which, float** ptr, int64_t* enc_stride) {
  if (!ctx || !ptr || !enc_stride) return DDRL_ERR_INVALID_ARG;
  const int64_t MB = ctx->cfg.max_batch;
  const Workspace& w = ctx->ws;
  switch (which) {
    case 0:
      if (!ctx->acts_stored) return DDRL_ERR_UNSUPPORTED;
      *ptr = w.a1; *enc_stride = MB * 32 * 400; break;
    case 1:
      if (!ctx->acts_stored) return DDRL_ERR_UNSUPPORTED;
      *ptr = w.a2; *enc_stride = MB * 64 * 81; break;
    case 2: *ptr = w.a3; *enc_stride = MB * FLAT; break;
    case 3: *ptr = w.h; *enc_stride = MB * FEAT; break;
    case 4: *ptr = w.dz1; *enc_stride = MB * 32 * 400; break;
    case 5: *ptr = w.dz2; *enc_stride = MB * 64 * 81; break;
    case 6: *ptr = w.dz3; *enc_stride = MB * FLAT; break;
    case 7: *ptr = w.dh; *enc_stride = MB * FEAT; break;
    case 8: *ptr = w.dlogits; *enc_stride = 0; break;
    case 9: *ptr = w.dvalue; *enc_stride = 0; break;
    // sign masks (32-bit words, returned through the float pointer; layouts in common.h Workspace::m1 / m2 / m3)
    case 10: *ptr = (float*)w.m1; *enc_stride = m1_words(MB); break;
    case 11: *ptr = (float*)w.m2; *enc_stride = MB * 81 * 2; break;
    case 12: *ptr = (float*)w.m3; *enc_stride = MB * 49 * 2; break;
    // per-sample scale g_s of the normalised backward (common.h Workspace::gsc): buffers 4..7 hold g_s^-1 x the true gradients
    case 13: *ptr = w.gsc; *enc_stride = MB; break;
    // running maxima / bounds behind the plane scales, amax[slot][encoder] (common.h AMAX_*): enc_stride = 1, slot stride = 2
    case 14: *ptr = w.amax; *enc_stride = 1; break;
    default: return DDRL_ERR_INVALID_ARG;
  }
  return DDRL_OK;
}

// ------------------------------------------------------------------------------------------
// pinned-host ring
// ------------------------------------------------------------------------------------------
}  // extern "C"

struct ddrl_ring {
  int64_t slot_bytes;
  int32_t n_slots;
  char* base;
  // slot states: 0 free, 1 being written, 2 committed, 3 copy in flight
  std::vector<int> state;
  std::vector<hipEvent_t> ev;
  int64_t head;    // next slot the producer acquires
  int64_t commit;  // next slot to be committed
  int64_t tail;    // next slot the consumer pops
  int64_t reclaim; // oldest slot whose copy may still be in flight
  std::mutex mu;
  std::condition_variable cv;
};

static void ring_reclaim_locked(ddrl_ring* r) {
  while (r->reclaim < r->tail) {
    const int s = (int)(r->reclaim % r->n_slots);
    if (r->state[s] != 3) break;
    if (hipEventQuery(r->ev[s]) != hipSuccess) break;
    r->state[s] = 0;
    r->reclaim++;
  }
}

extern "C" {

int32_t ddrl_ring_create(int64_t slot_bytes, int32_t n_slots, ddrl_ring** out) {
  if (slot_bytes < 1 || n_slots < 2 || !out) return DDRL_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DDRL_ERR_NO_DEVICE;
  ddrl_ring* r = new (std::nothrow) ddrl_ring();
  if (!r) return DDRL_ERR_NO_MEMORY;
  r->slot_bytes = (slot_bytes + 255) / 256 * 256;
  r->n_slots = n_slots;
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)r->slot_bytes * n_slots, hipHostMallocDefault) != hipSuccess) {
    delete r;
    return DDRL_ERR_HIP;
  }
  r->base = (char*)p;
  r->state.assign(n_slots, 0);
  r->ev.resize(n_slots);
  for (auto& e : r->ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  r->head = r->commit = r->tail = r->reclaim = 0;
  *out = r;
  return DDRL_OK;
}

int32_t ddrl_ring_destroy(ddrl_ring* r) {
  if (!r) return DDRL_ERR_INVALID_ARG;
  for (auto& e : r->ev) {
    hipEventSynchronize(e);
    hipEventDestroy(e);
  }
  hipHostFree(r->base);
  delete r;
  return DDRL_OK;
}

int32_t ddrl_ring_acquire(ddrl_ring* r, void** slot_host, int32_t timeout_ms) {
  if (!r || !slot_host) return DDRL_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(r->mu);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  for (;;) {
    ring_reclaim_locked(r);
    const int s = (int)(r->head % r->n_slots);
    if (r->head - r->reclaim < r->n_slots && r->state[s] == 0) {
      r->state[s] = 1;
      r->head++;
      *slot_host = r->base + (int64_t)s * r->slot_bytes;
      return DDRL_OK;
    }
    if (std::chrono::steady_clock::now() >= deadline) return DDRL_ERR_TIMEOUT;
    r->cv.wait_for(lk, std::chrono::microseconds(200));
  }
}

int32_t ddrl_ring_commit(ddrl_ring* r) {
  if (!r) return DDRL_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(r->mu);
  if (r->commit >= r->head) return DDRL_ERR_INVALID_ARG;
  const int s = (int)(r->commit % r->n_slots);
  if (r->state[s] != 1) return DDRL_ERR_INVALID_ARG;
  r->state[s] = 2;
  r->commit++;
  r->cv.notify_all();
  return DDRL_OK;
}

int32_t ddrl_ring_pop_to_device(ddrl_ring* r, void* dst, int64_t bytes, void* stream, int32_t timeout_ms) {
  if (!r || !dst || bytes < 1 || bytes > r->slot_bytes) return DDRL_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(r->mu);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  while (r->tail >= r->commit) {
    if (std::chrono::steady_clock::now() >= deadline) return DDRL_ERR_TIMEOUT;
    r->cv.wait_for(lk, std::chrono::microseconds(200));
  }
  const int s = (int)(r->tail % r->n_slots);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(dst, r->base + (int64_t)s * r->slot_bytes, (size_t)bytes, hipMemcpyHostToDevice, st) != hipSuccess)
    return DDRL_ERR_HIP;
  if (hipEventRecord(r->ev[s], st) != hipSuccess) return DDRL_ERR_HIP;
  r->state[s] = 3;
  r->tail++;
  r->cv.notify_all();
  return DDRL_OK;
}

int32_t ddrl_ring_pending(ddrl_ring* r, int32_t* n) {
  if (!r || !n) return DDRL_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(r->mu);
  *n = (int32_t)(r->commit - r->tail);
  return DDRL_OK;
}

// ------------------------------------------------------------------------------------------
// timers / profile
// ------------------------------------------------------------------------------------------
struct ddrl_timer {
  hipEvent_t a, b;
};

int32_t ddrl_timer_create(void** t) {
  if (!t) return DDRL_ERR_INVALID_ARG;
  ddrl_timer* x = new (std::nothrow) ddrl_timer();
  if (!x) return DDRL_ERR_NO_MEMORY;
  if (hipEventCreate(&x->a) != hipSuccess || hipEventCreate(&x->b) != hipSuccess) {
    delete x;
    return DDRL_ERR_HIP;
  }
  *t = x;
  return DDRL_OK;
}
int32_t ddrl_timer_destroy(void* t) {
  if (!t) return DDRL_ERR_INVALID_ARG;
  ddrl_timer* x = (ddrl_timer*)t;
  hipEventDestroy(x->a);
  hipEventDestroy(x->b);
  delete x;
  return DDRL_OK;
}
int32_t ddrl_timer_start(void* t, void* stream) {
  if (!t) return DDRL_ERR_INVALID_ARG;
  HIP_TRY(hipEventRecord(((ddrl_timer*)t)->a, (hipStream_t)stream));
  return DDRL_OK;
}
int32_t ddrl_timer_stop(void* t, void* stream) {
  if (!t) return DDRL_ERR_INVALID_ARG;
  HIP_TRY(hipEventRecord(((ddrl_timer*)t)->b, (hipStream_t)stream));
  return DDRL_OK;
}
int32_t ddrl_timer_elapsed_ms(void* t, float* ms) {
  if (!t || !ms) return DDRL_ERR_INVALID_ARG;
  ddrl_timer* x = (ddrl_timer*)t;
  HIP_TRY(hipEventSynchronize(x->b));
  HIP_TRY(hipEventElapsedTime(ms, x->a, x->b));
  return DDRL_OK;
}

int32_t ddrl_debug_keep_activations(ddrl_ctx* ctx, int32_t on) {
  if (!ctx) return DDRL_ERR_INVALID_ARG;
  ctx->keep_acts = on != 0;
  return DDRL_OK;
}

int32_t ddrl_profile_enable(ddrl_ctx* ctx, int32_t on) {
  if (!ctx) return DDRL_ERR_INVALID_ARG;
  if (on && !ctx->profile) {
    ctx->prof_names.clear();
    ctx->prof_ms.clear();
    ctx->prof_calls.clear();
  }
  ctx->profile = on != 0;
  ctx->profile_acting = on == 1;
  return DDRL_OK;
}

int32_t ddrl_profile_read(ddrl_ctx* ctx, char (*names)[48], float* ms, int32_t* calls, int32_t cap, int32_t* n) {
  if (!ctx || !n) return DDRL_ERR_INVALID_ARG;
  for (auto& e : ctx->prof_pending) {
    hipEventSynchronize(e.b);
    float t = 0.f;
    hipEventElapsedTime(&t, e.a, e.b);
    size_t i = 0;
    for (; i < ctx->prof_names.size(); ++i)
      if (ctx->prof_names[i] == e.name) break;
    if (i == ctx->prof_names.size()) {
      ctx->prof_names.push_back(e.name);
      ctx->prof_ms.push_back(0.0);
      ctx->prof_calls.push_back(0);
    }
    ctx->prof_ms[i] += t;
    ctx->prof_calls[i] += 1;
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  ctx->prof_pending.clear();
  const int32_t cnt = (int32_t)ctx->prof_names.size();
  *n = cnt;
  for (int32_t i = 0; i < cnt && i < cap; ++i) {
    if (names) {
      std::strncpy(names[i], ctx->prof_names[i].c_str(), 47);
      names[i][47] = 0;
    }
    if (ms) ms[i] = (float)ctx->prof_ms[i];
    if (calls) calls[i] = ctx->prof_calls[i];
  }
  return DDRL_OK;
}

}  // extern "C"
