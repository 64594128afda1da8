/*
 * ddrl.h -- C ABI of the MI355X-native DDRL4NAV actor-learner hot path (libddrl_hip.so).
 *
 * The reference has no FFI on this path: its boundary is a Python object protocol
 * (USTC_lab.nn.PPO called by USTC_lab/server/forward.py:107-182 and
 * USTC_lab/server/backward.py:168-217).  This header is what a cgo/ctypes/N-API binding of
 * that path would bind; every entry point names the reference code it replaces.
 *
 * Conventions
 *   - every function returns an int32 status (0 = DDRL_OK, negative = error); nothing throws;
 *   - all tensor pointers are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm tensors)
 *     unless the name ends in _host; the library allocates no device memory after
 *     ddrl_ctx_create (it allocates none at all: the caller passes the arenas and workspace);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are
 *     asynchronous on that stream and may be captured into a hipGraph;
 *   - one caller thread per context.
 *
 * Layouts (C-contiguous):
 *   frames   uint8  [n, C, 84, 84]     C = in_channels (1..4) stacked frames exactly as the env wrapper emits them
 *                                      before the /255.0 (reference warputils.py:274-301);
 *                                      the kernels apply float32(u8/255.0) themselves.
 *   params / grads / adam_m / adam_v   float32 flat arenas in the reference's
 *                                      named_parameters() order (USTC_lab/nn/base.py:60-66):
 *                                      actor.pre.{conv1,conv2,conv3,linear}.{weight,bias},
 *                                      actor.actor_linear.{weight,bias},
 *                                      critic.critic_linear.{weight,bias},
 *                                      critic.pre.{conv1,conv2,conv3,linear}.{weight,bias}.
 *                                      `grads` has DDRL_STATS_FLOATS extra floats at its tail
 *                                      (loss partial sums) so one all-reduce covers both.
 */
#ifndef DDRL_H_
#define DDRL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDRL_OK 0
#define DDRL_ERR_INVALID_ARG (-1)
#define DDRL_ERR_UNSUPPORTED (-2)
#define DDRL_ERR_WORKSPACE (-3)
#define DDRL_ERR_HIP (-4)
#define DDRL_ERR_NO_DEVICE (-5)
#define DDRL_ERR_TIMEOUT (-6)
#define DDRL_ERR_NO_MEMORY (-7) /* host allocation of a handle failed */

/* 2 (round 4): ddrl_op_clip_rmsprop takes alpha as a double; ddrl_encoder_backward consumes dh (rescales its rows in place);
 * ddrl_debug_buffer 4..7 hold per-sample NORMALISED gradients.  A host built against version 1 must be rebuilt. */
#define DDRL_ABI_VERSION 3
#define DDRL_STATS_FLOATS 8 /* tail of the grad arena, see ddrl_ppo_iter */

typedef struct ddrl_ctx ddrl_ctx;
typedef struct ddrl_ring ddrl_ring;
typedef struct ddrl_comm ddrl_comm;

/* Hyper-parameters: the ConfigNN contract (USTC_lab/config/config_nn.py:19-57). */
typedef struct ddrl_config {
  int32_t n_actions;      /* ACTION_OUTPUT_DIM, 6 for Pong                       */
  int32_t in_channels;    /* int_frame_stack: 1..4 (atari.yaml: 4)                */
  int32_t max_batch;      /* largest n / B any call will pass (sizes workspace);
                             1 .. 83,886 (32-bit byte offsets into the conv1 activations);
                             larger global batches: shard, or pass B_global to ddrl_ppo_iter */
  int32_t share_cnn_net;  /* SHARE_CNN_NET: 0 = actor and critic own an encoder each (default,
                             ppo.py:118-129), 1 = one shared prenet, one Adam over every
                             parameter on total_loss (ppo.py:110-117)                */
  int32_t clip_grad;      /* CLIP_GRID                                            */
  float clip_grad_norm;   /* CLIP_GRID_NUM = 0.5                                  */
  float actor_lr;         /* ACTOR_LEARNING_RATE = 5e-5                           */
  float critic_lr;        /* CRITIC_LEARNING_RATE = 1e-3                          */
  float adam_beta1;       /* torch.optim.Adam default 0.9                         */
  float adam_beta2;       /* 0.999                                                */
  float adam_eps;         /* 1e-8                                                 */
  float ppo_clip;         /* PPO_CLIP = 0.2                                       */
  float dual_clip;        /* DUEL_PPO_CLIP = 3                                    */
  float v_loss_theta;     /* V_LOSS_THETA = 1.0                                   */
  float ent_loss_theta;   /* ENTROPY_LOSS_THETA = 0.05                            */
  float learning_rate;    /* LEARNING_RATE = 2e-4, the single Adam of the shared mode */
  int32_t smooth_l1_loss; /* SMOOTH_L1_LOSS: 0 = mean((ret-v)^2)/2, 1 = F.smooth_l1_loss (ppo.py:53-57) */
} ddrl_config;

int32_t ddrl_abi_version(void);
const char* ddrl_status_string(int32_t status);

/* Fill `cfg` with the reference defaults (config_nn.py) for Pong. */
int32_t ddrl_config_default(ddrl_config* cfg);

/* Number of fp32 parameters (3,371,847 for the default net) and of the actor-side prefix
 * (the parameters stepped with actor_lr; the remainder is stepped with critic_lr --
 * USTC_lab/nn/ppo.py:41-42). */
int32_t ddrl_param_count(const ddrl_config* cfg, int64_t* n_params, int64_t* n_actor_params);

/* Bytes of caller-provided device workspace needed for cfg->max_batch. */
int32_t ddrl_workspace_bytes(const ddrl_config* cfg, int64_t* bytes);

/* Create a context over caller-owned device arenas.  grads must hold n_params +
 * DDRL_STATS_FLOATS floats; params/adam_m/adam_v hold n_params floats.  adam_m/adam_v must be
 * zeroed by the caller for a fresh optimiser.  Replaces PPO.__init__ (ppo.py:18-59). */
int32_t ddrl_ctx_create(const ddrl_config* cfg, float* params, float* grads, float* adam_m,
                        float* adam_v, void* workspace, int64_t workspace_bytes, ddrl_ctx** out);
int32_t ddrl_ctx_destroy(ddrl_ctx* ctx);

/* Tell the context that `params` was rewritten by the caller (load_state_dict /
 * updatenn_by_redis, USTC_lab/nn/base.py:68-95): derived weight layouts are rebuilt lazily. */
int32_t ddrl_params_changed(ddrl_ctx* ctx);

/* Adam step counter (torch keeps it in optimizer state; not saved by the reference's
 * checkpoints, backward.py:208-209). */
int32_t ddrl_get_step(const ddrl_ctx* ctx, int64_t* step);
int32_t ddrl_set_step(ddrl_ctx* ctx, int64_t step);

/* PPO.forward (ppo.py:72-75) + the acting glue of ForwardThread.run (forward.py:128-149).
 *   act_in == NULL : sample  action ~ Categorical(probs)  with the counter-based stream
 *                    u = U(seed, stream, sample_index)  and inverse-CDF over p_hat; write the
 *                    action (as float, like `.to(tensortype)`) to action_out.
 *   act_in != NULL : evaluate log-prob of the given actions (PPO.forward(states, act)).
 * Outputs (any may be NULL except value/probs): probs [n,A] = softmax output (what play_mode
 * returns, actor.py:94-96); value [n]; logp_out [n] = log(clamp(p_hat[a], eps, 1-eps)). */
int32_t ddrl_forward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, const float* act_in,
                     uint64_t seed, uint64_t stream_id, float* probs, float* value,
                     float* action_out, float* logp_out, void* stream);

/* torch.distributions.Categorical(probs) derived quantities (actor.py:97 + forward.py:137-138,
 * ppo.py:106): p_hat = p/sum(p), logits = log(clamp(p_hat, eps, 1-eps)), entropy. */
int32_t ddrl_categorical_stats(const float* probs, int32_t n, int32_t n_actions, float* p_hat,
                               float* logits, float* entropy, void* stream);

/* dist.sample() (forward.py:137) for an existing probs tensor: fresh inverse-CDF draws with the
 * counter-based stream U(seed, stream_id, i); logp_out may be NULL. */
int32_t ddrl_categorical_sample(const float* probs, int32_t n, int32_t n_actions, uint64_t seed,
                                uint64_t stream_id, float* action_out, float* logp_out, void* stream);

/* Encoder outputs of the last ddrl_forward / ddrl_ppo_iter call: h[e][i][512], e = 0 actor,
 * 1 critic (AtariPreNet.forward, atari_encoder.py:25-32).  Copies n*512 floats each. */
int32_t ddrl_last_features(ddrl_ctx* ctx, int32_t n, float* h_actor, float* h_critic, void* stream);

/* Agents._accumulate_rewards (USTC_lab/agent/agent.py:124-140), one value head.
 *   values [T+1, N] (row T = bootstrap), rewards [T, N], dones uint8 [T, N]
 *   adv [T, N], ret [T, N]   (ret = old value + advantage; no advantage normalisation). */
int32_t ddrl_gae(const float* values, const float* rewards, const uint8_t* dones, int32_t T,
                 int32_t N, float gamma, float landa, float* adv, float* ret, void* stream);

/* Status.update_reward_status (USTC_lab/agent/statistics.py:118-123) over the T steps of one rollout, on the device: per env
 *   rewards_sum += r_t;  rewards_episode = rewards_episode * (1 - d_t) + rewards_sum * d_t;  rewards_sum *= (1 - d_t)
 * in the reference's operation order.  rewards [T, N], dones uint8 [T, N]; rewards_sum / rewards_episode [N] are read and
 * written (state across rollouts; zero them once); trace [T, N] (may be NULL) receives rewards_episode after every step;
 * episodes_finished int32 [N] (may be NULL) is increased by the number of dones. */
int32_t ddrl_episode_returns(const float* rewards, const uint8_t* dones, int32_t T, int32_t N, float* rewards_sum,
                             float* rewards_episode, float* trace, int32_t* episodes_finished, void* stream);

/* One iteration of the loss + backward half of PPO.learn (ppo.py:82-126, non-shared branch):
 * forward both encoders on B samples, dual-clip surrogate / value / entropy terms, backward
 * into the grad arena.  Gradients and loss sums are scaled by 1/B_global so that a SUM over
 * data-parallel ranks equals the full-batch mean (B_global == B on one GPU).
 * grads[n_params + 0..2] receive this rank's share of (actor_loss, v_loss, entropy). */
int32_t ddrl_ppo_iter(ddrl_ctx* ctx, const uint8_t* frames, const float* actions,
                      const float* old_logps, const float* advs, const float* rets, int32_t B,
                      int64_t B_global, void* stream);

/* clip_grad_norm_(all params, clip_grad_norm) + actor Adam + critic Adam (ppo.py:125-129).
 * Run after the (optional) all-reduce of the grad arena.  grads[n_params + 4] <- global grad
 * norm, grads[n_params + 5] <- clip coefficient, grads[n_params + 3] <- total loss. */
int32_t ddrl_clip_adam_step(ddrl_ctx* ctx, void* stream);

/* ---- multi-GPU: one process per GPU, env shards, full parameter / Adam replica per rank.
 * The reference has no such path (USTC_lab/server/backward.py:167 "TODO support mutil GPU CARD").  A communicator wraps
 * an RCCL communicator (librccl is resolved with dlopen at first use; DDRL_ERR_UNSUPPORTED when it is absent):
 * rank 0 calls ddrl_comm_unique_id and hands the 128 bytes to the other ranks by any out-of-band means (a file, a TCP
 * socket, MPI, torchrun's store), then every rank calls ddrl_comm_create.
 *   ddrl_params_broadcast : make the replicas bit-identical at start-up (weights of `root`)
 *   ddrl_grad_allreduce   : SUM all-reduce of the flat gradient arena + DDRL_STATS_FLOATS loss tail (13,487,420 B for the
 *                           default net), between ddrl_ppo_iter (which scaled everything by 1/B_global) and
 *                           ddrl_clip_adam_step: every rank then holds the full-batch mean gradient and loss shares, the clip
 *                           sees the global norm (ppo.py:126) and the replicas stay in step.
 * All calls are asynchronous on `stream`. */
int32_t ddrl_comm_unique_id(uint8_t* out128);
/* Which librccl the library resolved (file path of the ncclAllReduce symbol, NUL-terminated into path_out[cap]) and its version code
 * (ncclGetVersion); diagnostics only (bench.py --preflight), no communicator or GPU needed.  Added in round 6 (additive: ABI 3). */
int32_t ddrl_comm_info(char* path_out, int64_t cap, int32_t* version_out);
int32_t ddrl_comm_create(const uint8_t* id128, int32_t rank, int32_t world, ddrl_comm** out);
int32_t ddrl_comm_destroy(ddrl_comm* comm);
int32_t ddrl_allreduce_f32(ddrl_comm* comm, float* buf, int64_t count, void* stream);
int32_t ddrl_broadcast_f32(ddrl_comm* comm, float* buf, int64_t count, int32_t root, void* stream);
int32_t ddrl_grad_allreduce(ddrl_ctx* ctx, ddrl_comm* comm, void* stream);
/* The same reduction in LAYER BUCKETS that overlap the backward.  After ddrl_grad_buckets_enable every
 * ddrl_ppo_iter records one event per bucket on its stream, in completion order: 0 head layers + loss tail, 1 conv1, 2 dense
 * (95 % of the bytes; ready before the conv3 / conv2 weight gradients run), 3 conv3, 4 conv2.  ddrl_grad_allreduce_overlapped
 * makes `comm_stream` wait for each event and reduces the bucket's ranges there, then makes `compute_stream` wait for the last
 * one: only the small conv2 bucket is exposed.  Call it right after ddrl_ppo_iter (it does not block the host).
 * ddrl_grad_bucket_info / _wait expose the ranges (offset, count in floats; up to two per bucket) and the stream wait, for
 * hosts that reduce through another communicator (torch.distributed in the Python mirror). */
int32_t ddrl_grad_buckets_enable(ddrl_ctx* ctx);
int32_t ddrl_grad_bucket_count(const ddrl_ctx* ctx, int32_t* n);
int32_t ddrl_grad_bucket_info(const ddrl_ctx* ctx, int32_t bucket, int64_t* offsets2, int64_t* counts2, int32_t* n_ranges);
int32_t ddrl_grad_bucket_wait(ddrl_ctx* ctx, int32_t bucket, void* stream);
/* Ordering against the compute stream AT THE CALL (round 4).  ddrl_grad_buckets_begin records an event on `compute_stream`;
 * *fresh = 1 when a ddrl_ppo_iter has recorded every bucket event since the previous reduction.  Otherwise (gradients from
 * another producer, accumulation, a retried iteration) the bucket events are stale: `comm_stream` is made to wait for the
 * call-time event at once and the host must NOT rely on ddrl_grad_bucket_wait.  ddrl_grad_bucket_wait_last makes `comm_stream`
 * wait for the call-time event; hosts issue it in front of the last bucket.  ddrl_grad_allreduce_overlapped does both itself. */
int32_t ddrl_grad_buckets_begin(ddrl_ctx* ctx, void* comm_stream, void* compute_stream, int32_t* fresh);
int32_t ddrl_grad_bucket_wait_last(ddrl_ctx* ctx, void* comm_stream);
int32_t ddrl_grad_allreduce_overlapped(ddrl_ctx* ctx, ddrl_comm* comm, void* comm_stream, void* compute_stream);
int32_t ddrl_params_broadcast(ddrl_ctx* ctx, ddrl_comm* comm, int32_t root, void* stream);

/* float32(uint8/255.0) for all 256 byte values, computed with the conv1 loader's arithmetic
 * (reference: warputils.py:300 divides in float64, forward.py:102-104 casts to float32). */
int32_t ddrl_u8_table(float* out256, void* stream);

/* Diagnostic view into the workspace, for parity tests of intermediate tensors:
 * which = 0 a1, 1 a2, 2 a3, 3 h, 4 dz1, 5 dz2, 6 dz3, 7 dh (all [e][max_batch][...], e stride
 * returned in floats), 8 dlogits [B,A], 9 dvalue [B]; 10 / 11 / 12: the sign masks of a1 / a2 / a3 that
 * the forward writes for the backward's leaky-ReLU decisions (32-bit words behind the float pointer,
 * bit set = activation not positive; m1 [e][max_batch * 400 columns], bit per output channel;
 * m2 [e][max_batch][81 pixels][2], m3 [e][max_batch][49 pixels][2], bit per channel of a lane half).
 * The backward runs on per-sample NORMALISED gradients: 4..7 hold the true per-sample gradient divided by the
 * power of two g_s = 2^floor(log2 max|dh_s|); 13 = g_s [e][max_batch].  14 = the running maxima / bounds behind
 * the fp16 plane scales, [slot][encoder] (e stride 1). */
int32_t ddrl_debug_buffer(ddrl_ctx* ctx, int32_t which, float** ptr, int64_t* enc_stride);
/* ddrl_forward of at most 512 samples runs conv1-conv3 in one kernel that keeps a1 / a2 on chip (csrc/act.hip); buffers 0 and 1
 * then hold nothing of that call and ddrl_debug_buffer answers DDRL_ERR_UNSUPPORTED for them.  on = 1: the same kernel also
 * stores a1 / a2 (for tests that look at them; outputs are bit-identical either way).  Training launches always store them. */
int32_t ddrl_debug_keep_activations(ddrl_ctx* ctx, int32_t on);

/* ---- pinned-host ring: replaces the Redis LPUSH/BRPOP shuttle of frames between env workers
 * and the learner (USTC_lab/agent/multiqueue.py:83-130, server/backward.py:145-151). --------- */
int32_t ddrl_ring_create(int64_t slot_bytes, int32_t n_slots, ddrl_ring** out);
int32_t ddrl_ring_destroy(ddrl_ring* ring);
/* Producer: get the next free pinned slot (blocks up to timeout_ms, DDRL_ERR_TIMEOUT). */
int32_t ddrl_ring_acquire(ddrl_ring* ring, void** slot_host, int32_t timeout_ms);
int32_t ddrl_ring_commit(ddrl_ring* ring);
/* Consumer: hipMemcpyAsync the oldest committed slot to dst (device) on `stream`; the slot is
 * recycled when the copy has completed. */
int32_t ddrl_ring_pop_to_device(ddrl_ring* ring, void* dst, int64_t bytes, void* stream, int32_t timeout_ms);
int32_t ddrl_ring_pending(ddrl_ring* ring, int32_t* n_committed);

/* ---- EasyBytes wire codec, host memory only -------------------
 * Byte-exact with USTC_lab/data/easybytes.py: array record = >h dtype code (1 u8, 2 f16, 3 f32,
 * 4 f64), >I count, >I ndim, ndim x >I dims, raw payload (:21-26,:63-75); forward-states message =
 * >Q payload length, 4 x >H ip, >I process_env_id, arrays (:141-149); backward blob = >Q len +
 * states arrays, >Q len + other-4 arrays, marshal tail (:151-162). */
typedef struct ddrl_eb_array {
  int32_t dtype, ndim;
  int64_t dims[8];
  int64_t count;
  int64_t data_offset; /* of the raw payload inside the scanned buffer */
  int64_t nbytes;
} ddrl_eb_array;
typedef struct ddrl_eb_msg {
  int32_t ip[4];
  uint32_t process_env_id;
  int64_t payload_offset, payload_len;
} ddrl_eb_msg;
int32_t ddrl_eb_array_bytes(int32_t dtype, int32_t ndim, const int64_t* dims, int64_t* nbytes);
int32_t ddrl_eb_encode_array(int32_t dtype, int32_t ndim, const int64_t* dims, const void* data,
                             uint8_t* out, int64_t cap, int64_t* written); /* encode_data, one array */
int32_t ddrl_eb_scan(const uint8_t* buf, int64_t len, ddrl_eb_array* out, int32_t cap, int32_t* n); /* decode_data */
int32_t ddrl_eb_forward_header(const int32_t ip[4], uint32_t process_env_id, uint64_t payload_len, uint8_t* out20);
int32_t ddrl_eb_scan_forward_states(const uint8_t* buf, int64_t len, ddrl_eb_msg* out, int32_t cap, int32_t* n);
/* Frames (array `state_index`) of every message of a batched forward-states item -> contiguous
 * uint8 in dst (e.g. a pinned ring slot); float payloads hold uint8/255.0 and are mapped back
 * exactly with round(x*255).  Replaces decode_forward_states + state2tensor for the frames. */
int32_t ddrl_eb_frames_to_u8(const uint8_t* buf, int64_t len, int32_t state_index, uint8_t* dst,
                             int64_t dst_cap, int64_t* n_samples, int64_t* sample_elems);
int32_t ddrl_eb_scan_backward(const uint8_t* buf, int64_t len, int64_t* states_off, int64_t* states_len,
                              int64_t* other_off, int64_t* other_len, int64_t* tail_off);
int32_t ddrl_eb_put_u64(uint64_t v, uint8_t* out8); /* big-endian >Q */

/* ---- measurement hooks (bench.py): HIP-event timing on the caller's stream -------------- */
int32_t ddrl_timer_create(void** timer);
int32_t ddrl_timer_destroy(void* timer);
int32_t ddrl_timer_start(void* timer, void* stream);
int32_t ddrl_timer_stop(void* timer, void* stream);
int32_t ddrl_timer_elapsed_ms(void* timer, float* ms); /* synchronises on the stop event */

/* Enable per-kernel HIP-event timing (diagnostic; off by default): on = 1 inside ddrl_ppo_iter,
 * ddrl_clip_adam_step and ddrl_forward; on = 2 not inside ddrl_forward, whose launches last tens of
 * microseconds and would be slowed by about a quarter by the event records around them; on = 0 off.
 * names/ms arrays are filled up to `cap` entries; returns the count in *n. */
int32_t ddrl_profile_enable(ddrl_ctx* ctx, int32_t on);
int32_t ddrl_profile_read(ddrl_ctx* ctx, char (*names)[48], float* ms, int32_t* calls, int32_t cap, int32_t* n);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points for the encoders outside the Atari fast path.
 *
 * The reference builds its other networks from torch modules: NavPreNet / NavPedPreNet /
 * NavPreNet1D (USTC_lab/nn/nav_encoder.py:12-128: Conv2d 3x3/5x5/7x7 + ReLU + max_pool2d(2),
 * Conv1d, Linear(+ReLU), torch.cat) and MLPPreNet (USTC_lab/nn/mlp_encoder.py:12-29).  Each
 * ddrl_op_* below replaces one of those torch operators (forward and the autograd backward it
 * implies) on caller-owned device buffers; the Python host composes them exactly where the
 * reference composes the torch modules (ddrl4nav_amd/nn/generic.py).  All tensors are fp32,
 * NCHW-dense inside a sample; `*_sn` are sample strides in floats (0 = dense).
 * ------------------------------------------------------------------------------------------ */
typedef struct ddrl_conv_desc {
  int32_t n;               /* samples                                                     */
  int32_t cin, h, w;       /* input  [n][cin][h][w]   (Conv1d: h = 1)                     */
  int32_t cout, kh, kw;    /* weight [cout][cin][kh][kw]  (torch layout)                  */
  int32_t stride;          /* 1, 2 or 4, both directions                                  */
  int32_t pad_h, pad_w;    /* zero padding                                                */
  int64_t in_sn, out_sn;   /* sample strides of input / output, 0 = dense                 */
} ddrl_conv_desc;

int32_t ddrl_op_conv_out_shape(const ddrl_conv_desc* d, int32_t* oh, int32_t* ow);
/* Derived weight layouts + index tables of one layer (rebuilt whenever the weights change; the size does not depend on d->n).
 * `packed` is READ-ONLY for every later call (ABI 3; ABI 2 kept per-launch scratch inside it).  The heavy nav layers (64->128 5x5 @22,
 * 128->256 3x3 @10, 64->128 3x3 @24, 128->256 3x3 @12) run as fp16 plane products on the 16-bit matrix pipe (csrc/pconv.hip) and need
 * one float of scratch per sample for the per-sample plane scales of a launch that is not handed its scales: `scales_scratch` of
 * ddrl_op_conv_forward / _dgrad / _forward_pool / _dgrad_pooled, ddrl_op_conv_scratch_floats(d) floats for a launch of d->n samples
 * (0 for layers that need none: NULL is accepted then, and whenever in_amax / dpool_amax are given).  One launch per scratch buffer
 * at a time.  The plane kernels need 16-byte aligned tensors and sample strides that are multiples of 4 floats; other views of the same
 * layers run on the generic gather kernels (csrc/gconv.hip, f32-input MFMA) like every geometry without a specialised kernel. */
int32_t ddrl_op_conv_pack_floats(const ddrl_conv_desc* d, int64_t* floats);
int32_t ddrl_op_conv_pack(const ddrl_conv_desc* d, const float* w, float* packed, void* stream);
int32_t ddrl_op_conv_scratch_floats(const ddrl_conv_desc* d, int64_t* floats);
/* out = act(conv2d(in, w) + bias); act: 0 none, 1 ReLU        (torch.nn.Conv2d / Conv1d + F.relu) */
int32_t ddrl_op_conv_forward(const ddrl_conv_desc* d, const float* in, const float* packed, const float* bias,
                             int32_t act, float* out, float* scales_scratch, float* out_amax, void* stream);
/* din = d(loss)/d(in) given dz = d(loss)/d(pre-activation output) */
/* Conv2d + ReLU + max_pool2d(2) in ONE launch (round 4), for the layers whose kernels pool in their epilogue -- NavPreNet1D's three
 * (3->64 7x7 @48, 64->128 5x5 @22, 128->256 3x3 @10): pooled [n][cout][oh/2][ow/2] (dense) and one decision byte per window as
 * ddrl_op_maxpool2_forward_idx leaves it (ddrl_op_maxpool2_backward_idx turns d(pooled) + code into d(pre-activation)); the
 * full-resolution activations are never written.  DDRL_ERR_UNSUPPORTED for every other layer: run ddrl_op_conv_forward and
 * ddrl_op_maxpool2_forward_idx instead.  `in` 16-byte aligned, sample stride a multiple of 4 floats. */
int32_t ddrl_op_conv_forward_pool(const ddrl_conv_desc* d, const float* in, const float* packed, const float* bias, float* pooled,
                                  uint8_t* code, const float* in_amax, float* scales_scratch, float* out_amax, void* stream);
/* PER-SAMPLE MAGNITUDES (`*_amax`, ABI 3).  The fp16-plane kernels scale every sample (dense layers: every row) by a power of two taken
 * from its largest magnitude.  What travels between operators is that magnitude itself -- n floats, amax[b] >= max |x[b][:]| (any upper
 * bound is valid; the tighter, the more bits the sample keeps) --, and it comes from the PRODUCER of the tensor wherever there is one:
 *   out_amax / din_amax (outputs, may be NULL): the operator RAISES amax[b] to the largest |value| it writes for sample b (an atomic
 *     maximum on the bit pattern: deterministic, and several operators may raise the same array -- the row tiles of one launch, or
 *     the layers that fill the slices of a torch.cat buffer).  The CALLER zeroes the array before the first producer runs.
 *   in_amax / dpool_amax / dout_amax (inputs, may be NULL): the magnitudes of the tensor the operator reads; NULL = the operator runs
 *     its own pre-pass over the tensor (a full read of it: ~12 % of a robot_nav PPO iteration when every operator did that).
 * ddrl_op_sample_amax (x: n samples of `elems` floats at stride sn) and ddrl_op_row_amax (x [n][ld], `width` columns; accumulate != 0:
 * raise instead of overwrite) are the stand-alone pre-passes for tensors that arrive from elsewhere.  The magnitudes of d(pooled) bound
 * those of the routed gradient.  ddrl_op_conv_pooled_uses_scales: 1 for the layers that read in_amax / dpool_amax (the first layer of a
 * nav encoder finds its scales inside its kernels). */
int32_t ddrl_op_sample_amax(const float* x, int64_t sn, int32_t elems, int32_t n, float* amax, void* stream);
int32_t ddrl_op_conv_pooled_uses_scales(const ddrl_conv_desc* d);
int32_t ddrl_op_conv_has_forward_pool(const ddrl_conv_desc* d);  /* 1 when ddrl_op_conv_forward_pool serves the layer, else 0 (host only) */
/* The backward of the same layers straight from d(pooled) [n][cout][oh/2][ow/2] and the decision bytes: the kernels form
 * d(pre-activation) while they stage it (the pooled gradient at each window's first maximum under the ReLU's sign, zero elsewhere --
 * what ddrl_op_maxpool2_backward_idx would write), so the full-resolution gradient is neither written nor read.  Same layers as
 * ddrl_op_conv_forward_pool (the 3-channel first layer has no data gradient: DDRL_ERR_UNSUPPORTED); dpool 16-byte aligned, dense. */
int32_t ddrl_op_conv_dgrad_pooled(const ddrl_conv_desc* d, const float* dpool, const uint8_t* code, const float* packed, float* din,
                                  const float* dpool_amax, float* scales_scratch, float* din_amax, void* stream);
int32_t ddrl_op_conv_wgrad_pooled(const ddrl_conv_desc* d, const float* in, const float* dpool, const uint8_t* code, const float* packed,
                                  float* ws, float* dw, float* db, const float* in_amax, const float* dpool_amax, void* stream);
int32_t ddrl_op_conv_dgrad(const ddrl_conv_desc* d, const float* dz, const float* packed, float* din, float* scales_scratch, void* stream);
/* dw [cout][cin][kh][kw], db [cout] (overwritten); `ws` = split-K scratch of ddrl_op_conv_ws_floats.
 * Requires oh*ow >= 32. */
int32_t ddrl_op_conv_ws_floats(const ddrl_conv_desc* d, int64_t* floats);
int32_t ddrl_op_conv_wgrad(const ddrl_conv_desc* d, const float* in, const float* dz, const float* packed, float* ws,
                           float* dw, float* db, void* stream);

/* F.max_pool2d(x, 2, stride=2) over `planes` = n*c planes of h x w (both even), and its backward
 * fused with the ReLU that precedes it in the reference (`a` = relu(conv) at full resolution):
 * dz = dpool routed to the first maximum of each window (PyTorch scan order), zero where a <= 0.
 * The full-resolution pointers (in, a, dz) must be 8-byte aligned (INVALID_ARG otherwise). */
int32_t ddrl_op_maxpool2_forward(const float* in, int64_t planes, int32_t h, int32_t w, float* out, void* stream);
int32_t ddrl_op_maxpool2_relu_backward(const float* a, const float* dpool, int64_t planes, int32_t h, int32_t w,
                                       float* dz, void* stream);

/* The same pair with the decisions kept in ONE BYTE per window (round 4): the forward also writes code[planes][h/2][w/2] (bits 0-1 =
 * position of the first maximum in PyTorch's scan order, bit 2 = the maximum is positive, i.e. the ReLU in front of the pool lets the
 * gradient through); the backward reads dpool and the code instead of the full-resolution activations (1.31 instead of 2.25 tensor
 * sizes of HBM traffic) and writes the same dz as ddrl_op_maxpool2_relu_backward.  Same alignment rule for `in`; dz 16-byte aligned. */
int32_t ddrl_op_maxpool2_forward_idx(const float* in, int64_t planes, int32_t h, int32_t w, float* out, uint8_t* code, void* stream);
int32_t ddrl_op_maxpool2_backward_idx(const float* dpool, const uint8_t* code, int64_t planes, int32_t h, int32_t w, float* dz,
                                      void* stream);

/* nn.Linear(K, N) (+ReLU): out[b][:] = act(in[b][:K] W^T + bias).  Leading dimensions are
 * multiples of 4 floats (>= K rounded up to 4), pointers 16-byte aligned, N a multiple of 4;
 * padding columns [K, ld_in) must hold finite values (they meet zero weights).
 * wt / wn = derived layouts written by ddrl_op_linear_pack (sizes from ddrl_op_linear_pack_floats). */
int32_t ddrl_op_linear_pack_floats(int32_t K, int32_t N, int64_t* wt_floats, int64_t* wn_floats);
int32_t ddrl_op_linear_pack(const float* w, int32_t K, int32_t N, float* wt, float* wn, void* stream);
/* ws: scratch of ddrl_op_linear_ws_floats(n, K, N) floats (lets small n x N problems split K over
 * workgroups), or NULL for a single pass.
 * in_amax / dout_amax: layers of K >= 128, N >= 64 in launches of n >= 128 rows (ddrl_op_linear_uses_planes: 1) run as fp16 plane
 * products and scale every ROW of `in` / `dout` by a power of two from the row's largest magnitude ("per-sample magnitudes" above: from
 * the tensor's producer, from ddrl_op_row_amax, or NULL = the operator's own pre-pass).  The forward and the weight gradient read the
 * same `in`, the data and the weight gradient the same `dout`: one array per tensor serves both. */
int32_t ddrl_op_linear_uses_planes(int32_t n, int32_t K, int32_t N);
int32_t ddrl_op_row_amax(const float* x, int64_t ld, int32_t width, int32_t n, float* amax, int32_t accumulate, void* stream);
int32_t ddrl_op_linear_forward(const float* in, int64_t ld_in, const float* wt, const float* bias, int32_t act,
                               float* out, int64_t ld_out, int32_t n, int32_t K, int32_t N, float* ws, const float* in_amax,
                               void* stream);
/* din[b][k] = [mask_src[b][k] > 0 or mask_src == NULL] * sum_n dout[b][n] W[n][k]; mask_src is the
 * (ReLU) output of the layer that produced `in`.  ws: the same scratch as the forward's (the 16-bit plane kernels of layers with
 * K >= 128, N >= 64 keep their per-row scales there for launches of n >= 128 rows), or NULL for the f32-input kernels.
 * din_amax (may be NULL): raised to the largest |din[b][k]| over the columns amax_lo <= k < amax_hi (amax_hi <= 0: all K columns;
 * amax_lo a multiple of 4) -- a slice when the consumer reads a slice of din (the layers behind a torch.cat). */
int32_t ddrl_op_linear_dgrad(const float* dout, int64_t ld_dout, const float* wn, const float* mask_src, int64_t ld_mask,
                             float* din, int64_t ld_din, int32_t n, int32_t K, int32_t N, float* ws, const float* dout_amax,
                             float* din_amax, int32_t amax_lo, int32_t amax_hi, void* stream);
int32_t ddrl_op_linear_ws_floats(int32_t n, int32_t K, int32_t N, int64_t* floats);
/* dw [N][K] = dout^T in, db [N] = column sums of dout (overwritten) */
int32_t ddrl_op_linear_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* ws, float* dw,
                             float* db, int32_t n, int32_t K, int32_t N, const float* in_amax, const float* dout_amax, void* stream);

/* Actor / critic heads on 512-wide encoder features (AC_INPUT_DIM, config_nn.py:23) with the PPO
 * loss block and its backward, for nets assembled from the operators above.  `continuous` selects
 * GaussionActor (USTC_lab/nn/actor.py:43-70: mu = Linear(h), std = exp(log_std), Normal, log-prob
 * summed over action dims) instead of CategoricalActor (actor.py:73-101).  Offsets are positions
 * (in floats) of the head parameters inside the caller's flat parameter / gradient arenas. */
typedef struct ddrl_heads_desc {
  int32_t continuous;  /* 0 = CategoricalActor (A <= 18), 1 = GaussionActor (D <= 8)          */
  int32_t n_actions;   /* ACTION_OUTPUT_DIM                                                   */
  int32_t shared;      /* SHARE_CNN_NET: both heads read h_actor, total_loss is differentiated */
  int32_t reserved;
  int64_t actor_w, actor_b, log_std, critic_w, critic_b; /* log_std: Gaussian only             */
  int64_t n_params;    /* grads[n_params .. +8) receive the loss statistics (see ddrl_ppo_iter) */
} ddrl_heads_desc;

int32_t ddrl_op_heads_ws_floats(const ddrl_heads_desc* d, int32_t max_n, int64_t* floats);
/* h_actor / h_critic: [n][512].  dist_out: probs [n][A] (categorical) or mu [n][D] (Gaussian).
 * act_in == NULL: sample (categorical: inverse CDF on the counter-based uniform stream, as
 * ddrl_forward; Gaussian: mu + std * Box-Muller normal of the same stream); action_out is [n] or [n][D]. */
int32_t ddrl_op_heads_act(const ddrl_heads_desc* d, const float* params, const float* h_actor, const float* h_critic,
                          int32_t n, const float* act_in, uint64_t seed, uint64_t stream_id, float* dist_out, float* value,
                          float* action_out, float* logp_out, void* stream);
/* Loss terms of PPO.learn (ppo.py:82-108) for this shard scaled by 1/B_global, d(loss)/d(h) into
 * dh_actor / dh_critic (shared: both into dh_actor), head parameter gradients and the loss
 * statistics into `grads` (overwritten at the head offsets). */
int32_t ddrl_op_heads_loss(const ddrl_heads_desc* d, const ddrl_config* cfg, const float* params, const float* h_actor,
                           const float* h_critic, int32_t n, const float* actions, const float* old_logps, const float* advs,
                           const float* rets, int64_t B_global, float* dh_actor, float* dh_critic, float* grads, float* ws,
                           void* stream);
/* clip_grad_norm_ + Adam (two lr groups split at n_actor, or one when shared) on flat arenas;
 * `step` is the 1-based Adam step count; ws = ddrl_op_clip_adam_ws_bytes bytes. */
int32_t ddrl_op_clip_adam_ws_bytes(int64_t* bytes);
int32_t ddrl_op_clip_adam(const ddrl_config* cfg, float* params, float* grads, float* m, float* v, int64_t n_params,
                          int64_t n_actor, int32_t shared, int64_t step, void* ws, void* stream);
/* d[b][k] = 0 where act[b][k] <= 0: ReLU backward for an encoder whose OUTPUT is a ReLU (MLPPreNet) */
int32_t ddrl_op_relu_mask(float* d, int64_t ld_d, const float* act, int64_t ld_act, int32_t n, int32_t width, void* stream);
/* dst[i] += src[i]: gradient accumulation over micro-batches */
int32_t ddrl_op_accumulate(float* dst, const float* src, int64_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * GAIL (USTC_lab/nn/GAIL.py, nn/ppo.py:61-62,97-107).
 * ------------------------------------------------------------------------------------------ */
/* The Atari encoder on its own, for callers that compose it with other operators (the GAIL discriminator's
 * `pre`, GAIL.py:27,65-66, and a generator whose heads carry an extra critic): AtariPreNet.forward
 * (atari_encoder.py:25-32) into the context's feature buffer and its backward from the context's dh buffer into
 * the context's grad arena (encoder slots only; the head slots are not touched).  The context is created with
 * share_cnn_net = 1; `params` / `grads` point at the encoder's first parameter (prenet.conv1.weight) inside
 * the caller's arenas; adam_m / adam_v may be NULL for such a context (ddrl_clip_adam_step then fails).
 * ddrl_encoder_buffers returns the device addresses of h [max_batch][512] and dh [max_batch][512].
 * ddrl_encoder_backward CONSUMES dh: its rows are rescaled in place by per-sample powers of two (the backward
 * runs on normalised gradients); write dh again before every call. */
int32_t ddrl_encoder_forward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, void* stream);
int32_t ddrl_encoder_backward(ddrl_ctx* ctx, const uint8_t* frames, int32_t n, void* stream);
int32_t ddrl_encoder_buffers(ddrl_ctx* ctx, float** h, float** dh);

/* One more value head next to the critic (PPO.add_critic, ppo.py:61-62): value[b] = w . h[b] + bias on 512-wide
 * features (Critic.forward, critic.py:14-21).  w [512] / b [1] may sit anywhere in an fp32 arena. */
int32_t ddrl_op_value_head_forward(const float* w, const float* b, const float* h, int64_t ld_h, int32_t n, float* value,
                                   void* stream);
/* Its loss term vlossf(rets, value) (ppo.py:101-103) scaled by 1/B_global: the loss share is ADDED to vloss_accum[0]
 * (= &grads[n_params + 1], so that VLoss / PpoTotalLoss include it, ppo.py:107-108), d(loss)/d(h) is ADDED to dh
 * (run after ddrl_op_heads_loss; `shared` != 0: the gradient carries v_loss_theta, as total_loss.backward() does),
 * dw [512] / db [1] receive the head's own gradient (either may be NULL: the reference never steps this head, see
 * nn/gail.py).  ws: ddrl_op_value_head_ws_floats floats. */
int32_t ddrl_op_value_head_ws_floats(int64_t* floats);
int32_t ddrl_op_value_head_loss(const ddrl_config* cfg, int32_t shared, const float* w, const float* b, const float* h,
                                int64_t ld_h, int32_t n, const float* rets, int64_t B_global, float* dh, int64_t ld_dh, float* dw,
                                float* db, float* vloss_accum, float* ws, void* stream);
/* One term of the discriminator loss (GAIL.py:78-80): loss[0] (+)= sign * sum(score[0:n, 0]) / n_total, and the
 * gradient it seeds: dscore[i][0] = sign / n_total, dscore[i][1..width) = 0 (score / dscore rows are `ld` / `ld_d`
 * floats apart; `width` = padded output width of the last dense layer). */
int32_t ddrl_op_wgan_terms(const float* score, int64_t ld, int32_t n, int64_t n_total, float sign, float* dscore, int64_t ld_d,
                           int32_t width, float* loss, int32_t accumulate, void* stream);
/* out[c] = sum over the n rows of x[:, c] for c < width, accumulated in double and rounded once (bias gradient of a
 * narrow dense layer, e.g. the discriminator's 1-wide score layer). */
int32_t ddrl_op_colsum(const float* x, int64_t ld, int32_t n, int32_t width, float* out, void* stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.RMSprop(lr, alpha, eps).step() (GAIL.py:28,83-84)
 * on flat arenas; grads[n_params + 4] <- grad norm, grads[n_params + 5] <- clip coefficient.  ws: as ddrl_op_clip_adam.
 * alpha is a double: torch casts alpha and (1 - alpha) to float32 separately, and 1 - alpha must be formed in double. */
int32_t ddrl_op_clip_rmsprop(float* params, float* grads, float* square_avg, int64_t n_params, float lr, double alpha, float eps,
                             float max_norm, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDRL_H_ */
