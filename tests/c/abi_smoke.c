/* C-only consumer of include/ddrl.h: proves that the drop-in boundary needs nothing but a C
 * compiler, the HIP runtime for device memory, and libddrl_hip.so -- the call sequence of
 * INTEGRATION.md section 2 on a tiny batch.  Built and run by tests/test_surface_gpu.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ddrl.h"

#define CHECK(x)                                                         \
  do {                                                                   \
    int32_t s_ = (x);                                                    \
    if (s_ != DDRL_OK) {                                                 \
      fprintf(stderr, "%s -> %s\n", #x, ddrl_status_string(s_));         \
      return 1;                                                          \
    }                                                                    \
  } while (0)
#define HIP(x)                                             \
  do {                                                     \
    if ((x) != hipSuccess) {                               \
      fprintf(stderr, "HIP call failed: %s\n", #x);        \
      return 2;                                            \
    }                                                      \
  } while (0)

#define STAGE(msg)              \
  do {                          \
    fprintf(stderr, "[abi_smoke] %s\n", msg); \
    fflush(stderr);             \
  } while (0)

int main(void) {
  enum { N = 8, T = 4, B = N * T };
  ddrl_config cfg;
  CHECK(ddrl_config_default(&cfg));
  cfg.max_batch = B;
  int64_t n_params = 0, n_actor = 0, ws_bytes = 0;
  CHECK(ddrl_param_count(&cfg, &n_params, &n_actor));
  CHECK(ddrl_workspace_bytes(&cfg, &ws_bytes));
  if (n_params != 3371847 || n_actor != 1687206) return 3;

  float *params, *grads, *m, *v, *probs, *values, *actions, *logps, *adv, *ret, *rewards;
  uint8_t *frames, *dones;
  void* ws;
  HIP(hipMalloc((void**)&params, n_params * 4));
  HIP(hipMalloc((void**)&grads, (n_params + DDRL_STATS_FLOATS) * 4));
  HIP(hipMalloc((void**)&m, n_params * 4));
  HIP(hipMalloc((void**)&v, n_params * 4));
  HIP(hipMalloc(&ws, (size_t)ws_bytes));
  HIP(hipMalloc((void**)&frames, (size_t)(T + 1) * N * 4 * 84 * 84));
  HIP(hipMalloc((void**)&probs, N * 6 * 4));
  HIP(hipMalloc((void**)&values, (T + 1) * N * 4));
  HIP(hipMalloc((void**)&actions, (T + 1) * N * 4));
  HIP(hipMalloc((void**)&logps, (T + 1) * N * 4));
  HIP(hipMalloc((void**)&adv, T * N * 4));
  HIP(hipMalloc((void**)&ret, T * N * 4));
  HIP(hipMalloc((void**)&rewards, T * N * 4));
  HIP(hipMalloc((void**)&dones, T * N));
  HIP(hipMemset(m, 0, n_params * 4));
  HIP(hipMemset(v, 0, n_params * 4));
  HIP(hipMemset(rewards, 0, T * N * 4));
  HIP(hipMemset(dones, 0, T * N));

  /* small deterministic weights and frames from the host */
  float* hp = (float*)malloc(n_params * 4);
  uint32_t s = 12345u;
  for (int64_t i = 0; i < n_params; ++i) {
    s = s * 1664525u + 1013904223u;
    hp[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * 0.04f;
  }
  HIP(hipMemcpy(params, hp, n_params * 4, hipMemcpyHostToDevice));
  size_t fbytes = (size_t)(T + 1) * N * 4 * 84 * 84;
  uint8_t* hf = (uint8_t*)malloc(fbytes);
  for (size_t i = 0; i < fbytes; ++i) {
    s = s * 1664525u + 1013904223u;
    hf[i] = (uint8_t)(s >> 24);
  }
  HIP(hipMemcpy(frames, hf, fbytes, hipMemcpyHostToDevice));

  ddrl_ctx* ctx = NULL;
  CHECK(ddrl_ctx_create(&cfg, params, grads, m, v, ws, ws_bytes, &ctx));
  STAGE("context created");
  const size_t fstep = (size_t)N * 4 * 84 * 84;
  for (int t = 0; t <= T; ++t)
    CHECK(ddrl_forward(ctx, frames + t * fstep, N, NULL, 7, (uint64_t)t, probs, values + t * N, actions + t * N,
                       logps + t * N, NULL));
  CHECK(ddrl_gae(values, rewards, dones, T, N, 0.99f, 0.95f, adv, ret, NULL));
  HIP(hipDeviceSynchronize());
  STAGE("acting + GAE done");
  /* the multi-GPU step of INTEGRATION.md section 2 with a communicator of ONE rank (a one-GPU box cannot hold two RCCL
   * ranks): unique id -> communicator -> parameter broadcast -> gradient all-reduce between ppo_iter and clip_adam */
  uint8_t id[128];
  ddrl_comm* comm = NULL;
  int32_t cs = ddrl_comm_unique_id(id);
  if (cs == DDRL_OK) {
    STAGE("creating the one-rank RCCL communicator");
    CHECK(ddrl_comm_create(id, 0, 1, &comm));
    CHECK(ddrl_params_broadcast(ctx, comm, 0, NULL));
    HIP(hipDeviceSynchronize());
    STAGE("communicator + broadcast done");
  } else if (cs != DDRL_ERR_UNSUPPORTED) { /* UNSUPPORTED = no librccl on this host */
    fprintf(stderr, "ddrl_comm_unique_id -> %s\n", ddrl_status_string(cs));
    return 5;
  }
  for (int it = 0; it < 2; ++it) {
    CHECK(ddrl_ppo_iter(ctx, frames, actions, logps, adv, ret, B, B, NULL));
    if (comm) CHECK(ddrl_grad_allreduce(ctx, comm, NULL));
    CHECK(ddrl_clip_adam_step(ctx, NULL));
    HIP(hipDeviceSynchronize());
    STAGE("PPO iteration done");
  }
  HIP(hipDeviceSynchronize());
  float stats[DDRL_STATS_FLOATS], hprobs[N * 6];
  HIP(hipMemcpy(stats, grads + n_params, sizeof(stats), hipMemcpyDeviceToHost));
  HIP(hipMemcpy(hprobs, probs, sizeof(hprobs), hipMemcpyDeviceToHost));
  float psum = 0.f;
  for (int j = 0; j < 6; ++j) psum += hprobs[j];
  int64_t step = 0;
  CHECK(ddrl_get_step(ctx, &step));
  if (comm) CHECK(ddrl_comm_destroy(comm));
  CHECK(ddrl_ctx_destroy(ctx));
  if (!(fabsf(psum - 1.0f) < 1e-5f) || !isfinite(stats[0]) || !isfinite(stats[1]) || !(stats[2] > 0.f) || step != 2) {
    fprintf(stderr, "unexpected results: psum %g losses %g %g %g step %lld\n", psum, stats[0], stats[1], stats[2], (long long)step);
    return 4;
  }
  printf("rccl communicator: %s\n", comm ? "used (1 rank)" : "librccl not found, skipped");
  printf("C ABI OK: actor_loss %.6f v_loss %.6f entropy %.6f grad_norm %.6f\n", stats[0], stats[1], stats[2], stats[4]);
  return 0;
}
