// Actor / critic heads, Categorical bookkeeping, PPO dual-clip loss and its gradient, fused with
// the backward of the two head layers.  One wavefront per sample: the 512-wide dot products
// are 8 elements per lane + a wave64 xor-shuffle butterfly; the A (<= 18) logits then live
// redundantly in every lane so softmax / loss / gradient need no further communication.
//
// Reference arithmetic replaced:
//   CategoricalActor._distribution / log_prob  USTC_lab/nn/actor.py:90-101
//   Critic.forward                             USTC_lab/nn/critic.py:14-21
//   ForwardThread.run sampling + log_prob      USTC_lab/server/forward.py:132-138
//   PPO.learn loss block + autograd            USTC_lab/nn/ppo.py:82-108,122-123
#include "kernels.h"
#include "ppo_math.h"

namespace ddrl {

// Two instantiations of the head kernels: A <= 8 keeps the actor head weights and their
// gradient accumulators in registers; 8 < A <= 18 (the full Atari action set) reads the weights
// from LDS and leaves the actor-head weight gradient to head_wgrad_kernel.
constexpr int MAXA_SMALL = 8, MAXA_LARGE = 18;
constexpr float CAT_EPS = 1.1920928955078125e-07f;  // torch.finfo(float32).eps


template <int MAXA, bool WLDS>
struct HeadRegs {
  float wa[WLDS ? 1 : MAXA][8];  // this lane's 8 columns of every actor row (register variant)
  const float* wl;               // LDS copy [MAXA][512] (LDS variant)
  float wc[8];
  float ba[MAXA];
  float bc;
  // this lane's 8 weights of actor row j
  __device__ __forceinline__ void row(int j, int lane, float* o) const {
    if constexpr (WLDS) {
      load8(wl + j * FEAT + lane * 8, o);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = wa[j][i];
    }
  }
};

// `wl` = MAXA*512 floats of LDS for the WLDS variant (every thread of the block must call this)
template <int MAXA, bool WLDS>
__device__ __forceinline__ void load_head_weights(HeadRegs<MAXA, WLDS>& R, const float* params, const ParamLayout& L,
                                                  int lane, float* wl) {
  if constexpr (WLDS) {
    for (int i = threadIdx.x; i < MAXA * FEAT; i += blockDim.x)
      wl[i] = (i < L.A * FEAT) ? params[L.actor_w + i] : 0.0f;
    R.wl = wl;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXA; ++j) R.ba[j] = (j < L.A) ? params[L.actor_b + min(j, L.A - 1)] : 0.0f;
  } else {
    R.wl = nullptr;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) {
      // rows past the last action repeat it: their logits are never read (softmax_categorical masks j >= A) and their gradients
      // are exact zeros times finite weights.  No branch around the loads: a branch makes the compiler wait for them at its
      // join, and every caller has more loads to request before it needs these
      const int jj = min(j, L.A - 1);
      const float4* src = (const float4*)(params + L.actor_w + (int64_t)jj * FEAT + lane * 8);
      const float4 x = src[0], y = src[1];
      R.wa[j][0] = x.x; R.wa[j][1] = x.y; R.wa[j][2] = x.z; R.wa[j][3] = x.w;
      R.wa[j][4] = y.x; R.wa[j][5] = y.y; R.wa[j][6] = y.z; R.wa[j][7] = y.w;
      R.ba[j] = params[L.actor_b + jj];
    }
  }
  // the flat arena is only 4-byte aligned at critic_w in general -> scalar loads
#pragma unroll
  for (int i = 0; i < 8; ++i) R.wc[i] = params[L.critic_w + lane * 8 + i];
  R.bc = params[L.critic_b];
}


template <int MAXA>
struct Dist {
  float p[MAXA];    // softmax output
  float q[MAXA];    // p / sum(p)                      (Categorical.probs)
  float lc[MAXA];   // log(clamp(q, eps, 1-eps))       (Categorical.logits)
  float ps;
};

template <int MAXA>
__device__ __forceinline__ void softmax_categorical(const float* z, int A, Dist<MAXA>& d) {
  float m = z[0];
#pragma unroll
  for (int j = 1; j < MAXA; ++j)
    if (j < A) m = fmaxf(m, z[j]);
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < MAXA; ++j) {
    d.p[j] = (j < A) ? expf(z[j] - m) : 0.0f;
    s += d.p[j];
  }
  d.ps = 0.0f;
#pragma unroll
  for (int j = 0; j < MAXA; ++j) {
    d.p[j] = d.p[j] / s;
    d.ps += d.p[j];
  }
#pragma unroll
  for (int j = 0; j < MAXA; ++j) {
    d.q[j] = d.p[j] / d.ps;
    d.lc[j] = logf(fminf(fmaxf(d.q[j], CAT_EPS), 1.0f - CAT_EPS));
  }
}

template <int MAXA>
__device__ __forceinline__ float pick(const float (&a)[MAXA], int idx) {
  // a chain of selects on registers.  Left to itself the compiler turns it into an indexed load from a private (scratch) copy of
  // the array: a dependent round trip through the vector memory path per call, four per sample in heads_loss; the empty asm keeps
  // every step a v_cndmask
  float r = a[0];
#pragma unroll
  for (int j = 1; j < MAXA; ++j) {
    r = (idx == j) ? a[j] : r;
    asm volatile("" : "+v"(r));
  }
  return r;
}

// --------------------------------------------------------------------------------------------
// acting: probs / value / sample-or-evaluate
// --------------------------------------------------------------------------------------------
// When fc_nsplit > 0 the encoder outputs arrive as split-K partial sums of the FC layer,
// fc_part[s][e][n][512] without bias (small-batch acting path, fc2.hip); they are summed here in
// split order, the linear bias is added and the finished h is written back to `h`.
template <int MAXA, bool WLDS>
__global__ __launch_bounds__(256) void heads_act_kernel(float* __restrict__ h, int64_t h_es,
                                                        const float* __restrict__ fc_part, int fc_nsplit,
                                                        const float* __restrict__ params, ParamLayout L, int n,
                                                        const float* __restrict__ act_in, uint64_t seed,
                                                        uint64_t stream_id, float* __restrict__ probs,
                                                        float* __restrict__ value, float* __restrict__ action_out,
                                                        float* __restrict__ logp_out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[WLDS ? MAXA * FEAT : 1];
  HeadRegs<MAXA, WLDS> R;
  load_head_weights(R, params, L, lane, wl);
  const int ec = L.NE - 1;  // encoder feeding the critic head (0 when the prenet is shared)
  float lba[8], lbc[8];  // the dense layer's bias (added here when the features arrive as split-K partial sums): requested with the head weights
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // no branch around the loads (no join to wait at): with finished features (fc_nsplit = 0; the operator form has no encoder in
    // its layout at all) they read the first 512 floats of the arena instead -- the actor head's, always there -- and are not used
    lba[i] = params[(fc_nsplit > 0 ? L.enc_base[0] + L.enc.lb : L.actor_w) + lane * 8 + i];
    lbc[i] = params[(fc_nsplit > 0 ? L.enc_base[1] + L.enc.lb : L.actor_w) + lane * 8 + i];
  }
  // one sample; called once outside the loop (the common case: one sample per wave) so that the loads above are only waited for
  // where their values are used -- in front of a loop the compiler drains them first
  auto one_sample = [&](const int b) {
    float ha[8], hc[8];
    if (fc_nsplit > 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ha[i] = 0.0f;
        hc[i] = 0.0f;
      }
      if (fc_nsplit == DDRL_FC_ACT_SPLITS) {
        // the usual case: ALL partials of both encoders requested at once (a loop with a run-time trip count waits for every pair
        // before it asks for the next: 14 dependent round trips); same order of additions.  224 staging registers: this kernel
        // runs one wave per SIMD
        constexpr int U = DDRL_FC_ACT_SPLITS;
        float ta[U][8], tc[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          load8(fc_part + (((int64_t)u * 2 + 0) * n + b) * FEAT + lane * 8, ta[u]);
          load8(fc_part + (((int64_t)u * 2 + ec) * n + b) * FEAT + lane * 8, tc[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ha[i] += ta[u][i];
            hc[i] += tc[u][i];
          }
      } else {
        for (int sp = 0; sp < fc_nsplit; ++sp) {
          float ta[8], tc[8];
          load8(fc_part + (((int64_t)sp * 2 + 0) * n + b) * FEAT + lane * 8, ta);
          load8(fc_part + (((int64_t)sp * 2 + ec) * n + b) * FEAT + lane * 8, tc);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            ha[i] += ta[i];
            hc[i] += tc[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ha[i] += lba[i];
        hc[i] += lbc[i];
      }
      store8(h + (int64_t)b * FEAT + lane * 8, ha);
      if (ec) store8(h + h_es + (int64_t)b * FEAT + lane * 8, hc);
    } else {
      load8(h + (int64_t)b * FEAT + lane * 8, ha);
      load8(h + ec * h_es + (int64_t)b * FEAT + lane * 8, hc);
    }
    float z[MAXA];
#pragma unroll
    for (int j = 0; j < MAXA; ++j) {
      float s = 0.0f, w[8];
      R.row(j, lane, w);
#pragma unroll
      for (int i = 0; i < 8; ++i) s = __builtin_fmaf(ha[i], w[i], s);
      z[j] = wave_sum(s) + R.ba[j];
    }
    float sv = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sv = __builtin_fmaf(hc[i], R.wc[i], sv);
    const float v = wave_sum(sv) + R.bc;
    Dist<MAXA> d;
    softmax_categorical(z, L.A, d);
    int a;
    if (act_in != nullptr) {
      a = (int)act_in[b];
    } else {
      const float u = hash_uniform(seed, stream_id, (uint64_t)b);
      float c = 0.0f;
      a = L.A - 1;
      bool done = false;
#pragma unroll
      for (int j = 0; j < MAXA; ++j) {
        if (j < L.A) {
          c += d.q[j];
          if (!done && u < c) {
            a = j;
            done = true;
          }
        }
      }
    }
    if (lane == 0) {
      value[b] = v;
      if (action_out) action_out[b] = (float)a;
      if (logp_out) logp_out[b] = pick(d.lc, a);
    }
    if (probs && lane < L.A) probs[(int64_t)b * L.A + lane] = pick(d.p, lane);
  };
  if (gw < n) one_sample(gw);
  for (int b = gw + nw; b < n; b += nw) one_sample(b);
}

// --------------------------------------------------------------------------------------------
// training: loss terms, d(loss)/d(logits, value), backward of both head layers
// hpart layout per workgroup: [A*512 dWa][512 dwc][A dba][1 dbc][actor_sum, v_sum, ent_sum]
// --------------------------------------------------------------------------------------------
// WLDS (A > 8): the actor-head weight / bias gradient slots of hpart are written by
// head_wgrad_kernel from dlogits instead (288 accumulator + weight registers do not fit a lane).
constexpr int LOSS_WAVES = 4;      // waves per workgroup of heads_loss: one per SIMD (register-resident head weights, A = 7 .. 8 and A > 8)
#ifndef DDRL_LOSS_WAVES6
#define DDRL_LOSS_WAVES6 8         // A <= 6 (Pong): the weight rows live in LDS, 245 registers -> two waves per SIMD (round 6)
#endif
constexpr int LOSS_WAVES6 = DDRL_LOSS_WAVES6;
#ifndef DDRL_LOSS_NS
#define DDRL_LOSS_NS 4      // samples per wave and turn (a power of two): the scalar chain of the loss block runs once per NS samples
#endif
constexpr int LOSS_NS = DDRL_LOSS_NS;
// GREG: the actor head's weight / bias gradient is accumulated in registers (else: head_wgrad_kernel, A > 8)
template <int MAXA, bool WLDS, bool GREG = !WLDS, int WAVES = LOSS_WAVES>
__global__ __launch_bounds__(WAVES * 64) void heads_loss_kernel(
    const float* __restrict__ h, int64_t h_es, const float* __restrict__ params, ParamLayout L, ddrl_config cfg, int n,
    const float* __restrict__ actions, const float* __restrict__ old_logps, const float* __restrict__ advs,
    const float* __restrict__ rets, float inv_b, float* __restrict__ dh, int64_t dh_es, float* __restrict__ dlogits,
    float* __restrict__ dvalue, float* __restrict__ hpart, int64_t hstride, float* __restrict__ gsc, int64_t gsc_es,
    float* __restrict__ amax) {
  __shared__ float red[(MAXA + 1) * FEAT + 2 * MAXA + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * WAVES + wave;
  const int nw = gridDim.x * WAVES;
  const int A = L.A;
  HeadRegs<MAXA, WLDS> R;
  load_head_weights(R, params, L, lane, red);  // LDS weights alias the reduction buffer (used after the loop)
  constexpr int GA = GREG ? MAXA : 1;
  float gwa[GA][8], gwc[8], gba[GA], gbc = 0.0f;
#pragma unroll
  for (int j = 0; j < GA; ++j) {
    gba[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) gwa[j][i] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) gwc[i] = 0.0f;
  double s_actor = 0.0, s_v = 0.0, s_ent = 0.0;
  // per-sample normalisation of the backward fused into the producer of dh (encoder.hip dh_normalise_kernel is the stand-alone form
  // for dh that arrives from elsewhere): running maxima of this wave's scales g_s and of its normalised |dh|
  float run_g[2] = {0.0f, 0.0f}, run_m[2] = {0.0f, 0.0f};
  float wc_absmax = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) wc_absmax = fmaxf(wc_absmax, fabsf(R.wc[i]));
  wc_absmax = wave_max(wc_absmax);
  // (exponent arithmetic on the bit pattern: frexpf / ldexpf are library calls, and this runs once per sample and encoder)
  auto scale_of = [&](float m, float& g, float& gi) {
    int ex = GSC_EXP_MIN;
    if (m > 0.0f && m < 3.0e38f) ex = min(max((int)((__float_as_uint(m) >> 23) & 0xFFu) - 127, GSC_EXP_MIN), GSC_EXP_MAX);  // floor(log2 m); subnormals clamp
    g = __uint_as_float((unsigned)(ex + 127) << 23);
    gi = __uint_as_float((unsigned)(127 - ex) << 23);
  };
  auto normalise = [&](float (&d)[8], float m, int e, int b) {  // m = max_k |d[k]| over the whole wave
    float g, gi;
    scale_of(m, g, gi);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] *= gi;
    if (lane == 0) gsc[e * gsc_es + b] = g;
    run_g[e] = fmaxf(run_g[e], g);
    run_m[e] = fmaxf(run_m[e], m * gi);
  };
  auto wave_absmax = [&](const float (&d)[8]) {
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(d[i]));
    return wave_max(m);
  };
  // shared prenet (ppo.py:110-117): one backward of total_loss = actor_loss + theta_v * v_loss
  // - theta_e * entropy, so the value gradient carries theta_v, the entropy has a gradient, and
  // both heads feed the one encoder.  Non-shared (ppo.py:118-129): actor_loss.backward() and
  // v_loss.backward() only -- no theta_v, no entropy gradient.
  const bool shared = (L.NE == 1);
  const int ec = L.NE - 1;
  // A wave takes NS samples per turn (round 6; one per turn before: 1,046 vector instructions per sample, most of them the softmax /
  // log / divide chain that all 64 lanes executed on the same wave-uniform numbers).  Phases of a turn:
  //   A  the NS x (A + 1) dot products: 8 features per lane, then wave reductions (TR below: all of them together);
  //   B  ONCE for the NS samples: one lane per sample -- softmax, Categorical bookkeeping, surrogate, value loss, entropy and
  //      d(loss)/d(logits, value) -- so the transcendental chain is paid once per NS samples;
  //   C  per sample: its dlogits / dvalue are read back from its lane (v_readlane: wave-uniform again); C1 = the weight-gradient sums and
  //      the request of the next turn's sample into the same registers, C2 = d(features), their per-sample normalisation, the stores.
  // Registers: 8 x (A + 1) gradient accumulators, NS x 16 features, the head weights.  With the weights in registers that is one wave per
  // SIMD (A = 7, 8); the A <= 6 form reads the weight rows from LDS (60 ds_read_b128 per turn) and runs two (245 registers).
  // No second register set for the next turn: the kernel would need 394 registers, 138 of them accumulation registers used as spill area.
  constexpr int NS = LOSS_NS;
  // TR: the NS x (A + 1) dot products of a turn are reduced TOGETHER (ppo_math.h, transposing reduction: 32 values in 32 exchanges
  // instead of 28 butterflies of 6); sample i's totals then sit in ROW i' = 2 (i & 1) + (i >> 1) of the wave (16 lanes), value j at lane
  // 8 (j & 1) + 4 (j >> 1 & 1) + 2 (j >> 2) of the row, and the row's first lane -- the one that works on the sample in phase B -- collects
  // them with seven DPP row shifts for all four samples at once.
  constexpr bool TR = (NS == 4) && (MAXA + 1 <= 8);
  const int ls = TR ? ((lane >> 5) | ((lane >> 3) & 2)) : (lane & (NS - 1));  // the sample of the turn this lane works on in phase B
  const bool owner = TR ? (lane & 15) == 0 : lane < NS;
  auto owner_lane = [](int i) { return TR ? 32 * (i & 1) + 16 * (i >> 1) : i; };
  const bool bit3 = lane & 8, bit2 = lane & 4, bit1 = lane & 2;
  float ha[NS][8], hc[NS][8];
  auto request = [&](int b0, float (&xa)[NS][8], float (&xc)[NS][8]) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int b = min(b0 + i, n - 1);               // past the end: re-read the last sample, masked below
      load8(h + (int64_t)b * FEAT + lane * 8, xa[i]);
      load8(h + ec * h_es + (int64_t)b * FEAT + lane * 8, xc[i]);
    }
  };
  auto bcast = [&](float x, int i) { return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), owner_lane(i))); };
  if (gw * NS < n) request(gw * NS, ha, hc);
  for (int b0 = gw * NS; b0 < n; b0 += nw * NS) {
    // the lane's own sample of phase B: its four scalars come straight from memory (contiguous over the NS lanes)
    const int bl = min(b0 + ls, n - 1);
    const bool live = b0 + ls < n && owner;           // this lane's phase-B results count (the other lanes repeat them or work on junk)
    const float s_act = actions[bl], s_adv = advs[bl], s_olp = old_logps[bl], s_ret = rets[bl];
    // ---- phase A (a lane keeps only the results of ITS sample of phase B: no [NS][A + 1] array of logits in registers)
    float zl[MAXA], v = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) zl[j] = 0.0f;
    if constexpr (TR) {
      // value index = i + 4 j (j = MAXA: the critic's dot product; j > MAXA: padding): level 32 pairs samples (0, 1) / (2, 3) of one j
      float zz[8];  // after level 16: one register per j, row i' = sample i
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j > MAXA) { zz[j] = 0.0f; continue; }
        float w[8];
        if (j < MAXA) R.row(j, lane, w);
        float y[2];
#pragma unroll
        for (int i1 = 0; i1 < 2; ++i1) {
          float s[2] = {0.0f, 0.0f};
#pragma unroll
          for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
            for (int k = 0; k < 8; ++k)
              s[i0] = j < MAXA ? __builtin_fmaf(ha[2 * i1 + i0][k], w[k], s[i0]) : __builtin_fmaf(hc[2 * i1 + i0][k], R.wc[k], s[i0]);
          y[i1] = swap_add32(s[0], s[1]);
        }
        zz[j] = swap_add16(y[0], y[1]);
      }
      float w4[4], w2[2];
#pragma unroll
      for (int q = 0; q < 4; ++q) w4[q] = fold_add8(zz[2 * q], zz[2 * q + 1], bit3);
#pragma unroll
      for (int q = 0; q < 2; ++q) w2[q] = fold_add4(w4[2 * q], w4[2 * q + 1], bit2);
      float tot = fold_add2(w2[0], w2[1], bit1);
      tot += dpp_lane<0xB1>(tot, tot);
      // the row's first lane collects its sample's values: row_shl:n reads lane + n of the row
      float g[8];
      g[0] = tot;
      g[1] = dpp_lane<0x108>(tot, tot), g[2] = dpp_lane<0x104>(tot, tot), g[3] = dpp_lane<0x10C>(tot, tot);
      g[4] = dpp_lane<0x102>(tot, tot), g[5] = dpp_lane<0x10A>(tot, tot), g[6] = dpp_lane<0x106>(tot, tot);
      g[7] = dpp_lane<0x10E>(tot, tot);
#pragma unroll
      for (int j = 0; j < MAXA; ++j) zl[j] = g[j] + R.ba[j];
      v = g[MAXA] + R.bc;
    } else {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
          float s = 0.0f, w[8];
          R.row(j, lane, w);
#pragma unroll
          for (int k = 0; k < 8; ++k) s = __builtin_fmaf(ha[i][k], w[k], s);
          const float zj = wave_sum(s) + R.ba[j];
          zl[j] = (ls == i) ? zj : zl[j];
        }
        float sv = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sv = __builtin_fmaf(hc[i][k], R.wc[k], sv);
        const float vi = wave_sum(sv) + R.bc;
        v = (ls == i) ? vi : v;
      }
    }
    // ---- phase B: this lane's sample
    Dist<MAXA> d;
    softmax_categorical(zl, A, d);
    const int a = (int)s_act;
    const float adv = s_adv;
    const float logp = pick(d.lc, a);
    const SurrogateTerm sg = ppo_surrogate(logp, s_olp, adv, cfg, inv_b);
    const float err = s_ret - v;
    double v_el = 0.0;
    const float gv_unit = value_loss_element(err, cfg, v_el);  // d(v_loss element)/d(v) before the 1/B
    float ent = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
      if (j < A) ent += d.lc[j] * d.q[j];
    if (live) {
      s_actor += (double)sg.term;
      s_v += v_el;
      s_ent += (double)(-ent);
    }

    const float g_logp = sg.g_logp;
    // ---- log(clamp(q_a)) , q = p / sum(p) , softmax ----
    const float qa = pick(d.q, a), pa = pick(d.p, a);
    const float qa_c = fminf(fmaxf(qa, CAT_EPS), 1.0f - CAT_EPS);
    const float g_qa = (qa >= CAT_EPS && qa <= 1.0f - CAT_EPS) ? g_logp / qa_c : 0.0f;
    const float g_ps = -g_qa * pa / (d.ps * d.ps);
    float gp[MAXA], dot = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) gp[j] = ((j == a) ? g_qa / d.ps : 0.0f) + g_ps;
    if (shared) {
      // -theta_e * mean(H), H = -sum_j q_j * log(clamp(q_j)): dH/dq_j = -(lc_j + q_j * [in range] / clamp(q_j))
      const float g_h = -cfg.ent_loss_theta * inv_b;
      float gq[MAXA], gps_e = 0.0f;
#pragma unroll
      for (int j = 0; j < MAXA; ++j) {
        const float qj = d.q[j];
        const float qc = fminf(fmaxf(qj, CAT_EPS), 1.0f - CAT_EPS);
        const float inr = (qj >= CAT_EPS && qj <= 1.0f - CAT_EPS) ? qj / qc : 0.0f;
        gq[j] = (j < A) ? -g_h * (d.lc[j] + inr) : 0.0f;
        gps_e -= gq[j] * d.p[j];
      }
      gps_e = gps_e / (d.ps * d.ps);
#pragma unroll
      for (int j = 0; j < MAXA; ++j) gp[j] += gq[j] / d.ps + gps_e;
    }
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
      if (j < A) dot += gp[j] * d.p[j];
    float gzl[MAXA];
#pragma unroll
    for (int j = 0; j < MAXA; ++j) gzl[j] = (j < A) ? (gp[j] - dot) * d.p[j] : 0.0f;
    const float gvl = shared ? gv_unit * inv_b * cfg.v_loss_theta : gv_unit * inv_b;

    // ---- phase C: head layers backward, sample by sample.  C1 takes what needs the features (the weight-gradient sums) and
    // requests the next turn's sample into the registers it has just finished with; C2 (d(features), their normalisation, the stores)
    // needs only weights and the sample's dlogits / dvalue, so the requests have all of C2 to arrive before the next phase A -- which,
    // reducing the four samples together, wants all four at its start.
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int b = b0 + i;
      if (b >= n) break;  // wave-uniform
      float gz[MAXA];
#pragma unroll
      for (int j = 0; j < MAXA; ++j) gz[j] = bcast(gzl[j], i);
      const float gv = bcast(gvl, i);
#pragma unroll
      for (int k = 0; k < 8; ++k) gwc[k] = __builtin_fmaf(gv, hc[i][k], gwc[k]);
      if constexpr (GREG) {
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
          gba[j] += gz[j];
#pragma unroll
          for (int k = 0; k < 8; ++k) gwa[j][k] = __builtin_fmaf(gz[j], ha[i][k], gwa[j][k]);
        }
      }
      gbc += gv;
      {  // the next turn's sample i (past the end: the last sample again, never used)
        const int bn = min(b + nw * NS, n - 1);
        load8(h + (int64_t)bn * FEAT + lane * 8, ha[i]);
        load8(h + ec * h_es + (int64_t)bn * FEAT + lane * 8, hc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int b = b0 + i;
      if (b >= n) break;  // wave-uniform
      float gz[MAXA];
#pragma unroll
      for (int j = 0; j < MAXA; ++j) gz[j] = bcast(gzl[j], i);
      const float gv = bcast(gvl, i);
      float da[8], dc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) da[k] = 0.0f;
#pragma unroll
      for (int j = 0; j < MAXA; ++j) {  // per element the same j-ordered fma chain as sum_j gz[j] * Wa[j][k]
        float w[8];
        R.row(j, lane, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) da[k] = __builtin_fmaf(gz[j], w[k], da[k]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dc[k] = gv * R.wc[k];
      if (shared) {
#pragma unroll
        for (int k = 0; k < 8; ++k) da[k] += dc[k];
        if (gsc != nullptr) normalise(da, wave_absmax(da), 0, b);
        store8(dh + (int64_t)b * FEAT + lane * 8, da);
      } else {
        if (gsc != nullptr) {
          normalise(da, wave_absmax(da), 0, b);
          // the critic's row is gv * w_c: its largest magnitude is |gv| x the (loop-invariant) largest |w_c| -- exactly, since rounding
          // a product is monotone in the factor -- so only the actor's row needs a reduction
          normalise(dc, fabsf(gv) * wc_absmax, 1, b);
        }
        store8(dh + (int64_t)b * FEAT + lane * 8, da);
        store8(dh + dh_es + (int64_t)b * FEAT + lane * 8, dc);
      }
      if (lane < A) dlogits[(int64_t)b * A + lane] = pick(gz, lane);
      if (lane == 0) dvalue[b] = gv;
    }
  }
  // the loss sums live in the NS owner lanes (one sample each per turn): added in sample order into lane 0
  {
    double ta = 0.0, tv = 0.0, te = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      ta += __shfl(s_actor, owner_lane(i), 64);
      tv += __shfl(s_v, owner_lane(i), 64);
      te += __shfl(s_ent, owner_lane(i), 64);
    }
    s_actor = ta, s_v = tv, s_ent = te;
  }

  if (gsc != nullptr && lane == 0) {  // AMAX_GMAX / AMAX_DH: one look (and rarely an atomic) per wave and encoder
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (e > ec) break;
      const unsigned gb = __float_as_uint(run_g[e]), mb = __float_as_uint(run_m[e]);
      unsigned* sg = (unsigned*)(amax + amax_idx(AMAX_GMAX, e));
      unsigned* sm = (unsigned*)(amax + amax_idx(AMAX_DH, e));
      if (gb > __hip_atomic_load(sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sg, gb);
      if (mb > __hip_atomic_load(sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(sm, mb);
    }
  }
  // ---- workgroup reduction, waves accumulate in turn (fixed order) -> hpart[blockIdx.x] ----
  constexpr int SCAL = (MAXA + 1) * FEAT;
  if constexpr (WLDS) __syncthreads();  // every wave is done with the LDS weights aliased by `red`
  for (int w = 0; w < WAVES; ++w) {
    if (wave == w) {
      const bool first = (w == 0);
      if constexpr (GREG) {
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int idx = j * FEAT + lane * 8 + i;
            red[idx] = first ? gwa[j][i] : red[idx] + gwa[j][i];
          }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = MAXA * FEAT + lane * 8 + i;
        red[idx] = first ? gwc[i] : red[idx] + gwc[i];
      }
      if (lane == 0) {
        if constexpr (GREG) {
#pragma unroll
          for (int j = 0; j < MAXA; ++j) red[SCAL + j] = first ? gba[j] : red[SCAL + j] + gba[j];
        }
        red[SCAL + MAXA] = first ? gbc : red[SCAL + MAXA] + gbc;
        red[SCAL + MAXA + 1] = first ? (float)s_actor : red[SCAL + MAXA + 1] + (float)s_actor;
        red[SCAL + MAXA + 2] = first ? (float)s_v : red[SCAL + MAXA + 2] + (float)s_v;
        red[SCAL + MAXA + 3] = first ? (float)s_ent : red[SCAL + MAXA + 3] + (float)s_ent;
      }
    }
    __syncthreads();
  }
  float* out = hpart + (int64_t)blockIdx.x * hstride;
  if constexpr (GREG) {
    for (int i = threadIdx.x; i < A * FEAT; i += WAVES * 64) out[i] = red[i];
    if (threadIdx.x < A) out[(A + 1) * FEAT + threadIdx.x] = red[SCAL + threadIdx.x];
  }
  for (int i = threadIdx.x; i < FEAT; i += WAVES * 64) out[A * FEAT + i] = red[MAXA * FEAT + i];
  if (threadIdx.x == 0) out[(A + 1) * FEAT + A] = red[SCAL + MAXA];
  if (threadIdx.x < 3) out[(A + 1) * FEAT + A + 1 + threadIdx.x] = red[SCAL + MAXA + 1 + threadIdx.x];
}

// Actor-head weight / bias gradient for A > 8, from the dlogits heads_loss_kernel left behind:
//   hpart[wg][j*512 + k] = sum_{b in the workgroup's samples} dlogits[b][j] * h_actor[b][k],  [A*512+512 + j] = sum dlogits[b][j]
// (samples are dealt to workgroups round-robin; fixed order -> deterministic).
template <int MAXA>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ h, const float* __restrict__ dlogits,
                                                         int n, int A, float* __restrict__ hpart, int64_t hstride) {
  float acc[MAXA][2], bsum[MAXA];
#pragma unroll
  for (int j = 0; j < MAXA; ++j) acc[j][0] = acc[j][1] = bsum[j] = 0.0f;
  const int k = threadIdx.x;
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    const float h0 = h[(int64_t)b * FEAT + k], h1 = h[(int64_t)b * FEAT + 256 + k];
#pragma unroll
    for (int j = 0; j < MAXA; ++j) {
      const float g = (j < A) ? dlogits[(int64_t)b * A + min(j, A - 1)] : 0.0f;
      acc[j][0] = __builtin_fmaf(g, h0, acc[j][0]);
      acc[j][1] = __builtin_fmaf(g, h1, acc[j][1]);
      bsum[j] += g;
    }
  }
  float* out = hpart + (int64_t)blockIdx.x * hstride;
#pragma unroll
  for (int j = 0; j < MAXA; ++j) {
    if (j < A) {
      out[j * FEAT + k] = acc[j][0];
      out[j * FEAT + 256 + k] = acc[j][1];
      if (threadIdx.x == 0) out[(A + 1) * FEAT + j] = bsum[j];
    }
  }
}

// grads[head params] = sum over workgroups (fixed order, ppo_math.h sum_partials8: the arithmetic of the earlier one-thread-per-element
// form, so the gradients are bit-identical to it); grads[n_params+0..2] = loss shares, by the last workgroup (one wave per loss).
__global__ __launch_bounds__(256) void heads_reduce_kernel(const float* __restrict__ hpart, int64_t hstride, int nwg,
                                                           ParamLayout L, ddrl_config cfg, float inv_b,
                                                           float* __restrict__ grads) {
  __shared__ double sh[8][RED_OUT];
  const int A = L.A;
  const int nloss0 = (A + 1) * FEAT + A + 1;
  if (blockIdx.x == gridDim.x - 1) {
    const int k = threadIdx.x >> 6;
    if (k >= 3) return;
    const double s = wave_sum_partials(hpart, hstride, nwg, nloss0 + k);
    double r;
    if (k == 0) r = -s * (double)inv_b;            // actor_loss = -mean(term)
    else if (k == 1) r = s * (double)inv_b * (cfg.smooth_l1_loss ? 1.0 : 0.5);  // v_loss = mean(err^2)/2
    else r = s * (double)inv_b;                    // entropy = mean(H)
    if ((threadIdx.x & 63) == 0) grads[L.n_params + k] = (float)r;
    return;
  }
  const int i = blockIdx.x * RED_OUT + (threadIdx.x & (RED_OUT - 1));
  const float sum = sum_partials8(hpart, hstride, nwg, min(i, nloss0 - 1), sh);
  if (threadIdx.x >= RED_OUT || i >= nloss0) return;
  int64_t dst;
  if (i < A * FEAT) dst = L.actor_w + i;
  else if (i < (A + 1) * FEAT) dst = L.critic_w + (i - A * FEAT);
  else if (i < (A + 1) * FEAT + A) dst = L.actor_b + (i - (A + 1) * FEAT);
  else dst = L.critic_b;
  grads[dst] = sum;
}

__global__ __launch_bounds__(256) void categorical_stats_kernel(const float* __restrict__ probs, int n, int A,
                                                                float* __restrict__ p_hat, float* __restrict__ logits,
                                                                float* __restrict__ entropy) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  float ps = 0.0f;
  for (int j = 0; j < A; ++j) ps += probs[(int64_t)b * A + j];
  float ent = 0.0f;
  for (int j = 0; j < A; ++j) {
    const float q = probs[(int64_t)b * A + j] / ps;
    const float lc = logf(fminf(fmaxf(q, CAT_EPS), 1.0f - CAT_EPS));
    if (p_hat) p_hat[(int64_t)b * A + j] = q;
    if (logits) logits[(int64_t)b * A + j] = lc;
    ent += lc * q;
  }
  if (entropy) entropy[b] = -ent;
}

// fresh draws from Categorical(probs) with the same inverse-CDF contract as heads_act
__global__ __launch_bounds__(256) void categorical_sample_kernel(const float* __restrict__ probs, int n, int A,
                                                                 uint64_t seed, uint64_t stream_id,
                                                                 float* __restrict__ action, float* __restrict__ logp) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  float ps = 0.0f;
  for (int j = 0; j < A; ++j) ps += probs[(int64_t)b * A + j];
  const float u = hash_uniform(seed, stream_id, (uint64_t)b);
  float c = 0.0f, qa = 0.0f;
  int a = A - 1;
  bool done = false;
  for (int j = 0; j < A; ++j) {
    const float q = probs[(int64_t)b * A + j] / ps;
    c += q;
    if (!done && u < c) {
      a = j;
      qa = q;
      done = true;
    }
    if (!done && j == A - 1) qa = q;
  }
  action[b] = (float)a;
  if (logp) logp[b] = logf(fminf(fmaxf(qa, CAT_EPS), 1.0f - CAT_EPS));
}
void launch_categorical_sample(const float* probs, int n, int A, uint64_t seed, uint64_t stream_id, float* action,
                               float* logp, hipStream_t st) {
  hipLaunchKernelGGL(categorical_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, st, probs, n, A, seed, stream_id,
                     action, logp);
}

void launch_heads_act(const HeadsCall& c, const float* act_in, uint64_t seed, uint64_t stream_id, float* probs,
                      float* value, float* action_out, float* logp_out, hipStream_t st) {
#ifndef DDRL_HEADS_ACT_WAVES
#define DDRL_HEADS_ACT_WAVES 1
#endif
  // register-resident head weights (A <= 8): ONE wave per workgroup, so that the samples of a small acting batch spread over the CUs (each
  // wave pulls 57 KB of split-K partials through its CU's path to L2; four per CU were 64 busy CUs of 256); A > 8 shares the LDS copy of
  // the head weights between four waves
  const bool small = c.L->A <= MAXA_SMALL;
  const int wpw = small ? DDRL_HEADS_ACT_WAVES : 4;
  int wgs = (c.n + wpw - 1) / wpw;
  if (wgs > 1024) wgs = 1024;
  const int nsplit = c.plain_features ? 1 : fc_forward_splits(c.n);
  auto kern = small ? heads_act_kernel<MAXA_SMALL, false> : heads_act_kernel<MAXA_LARGE, true>;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * wpw), 0, st, c.ws->h, c.h_es != HeadsCall::ES_UNSET ? c.h_es : c.max_batch * FEAT,
                     nsplit > 1 ? c.ws->wpart : nullptr, nsplit > 1 ? nsplit : 0, c.params, *c.L, c.n,
                     act_in, seed, stream_id, probs, value, action_out, logp_out);
}

void launch_heads_loss(const HeadsCall& c, const float* actions, const float* old_logps, const float* advs,
                       const float* rets, float inv_b, float* grads, hipStream_t st) {
  const int64_t hs = hpart_stride(c.L->A);
  const bool large = c.L->A > MAXA_SMALL;
  // A <= 6 (Pong's six actions): the four samples' 7 x 4 dot products are reduced together (TR in the kernel), weight rows in LDS, eight waves
  const bool six = c.L->A <= 6;
  auto kern = large ? heads_loss_kernel<MAXA_LARGE, true> : (six ? heads_loss_kernel<6, LOSS_WAVES6 == 8, true, LOSS_WAVES6> : heads_loss_kernel<MAXA_SMALL, false>);
  hipLaunchKernelGGL(kern, dim3(HEAD_WG), dim3((six ? LOSS_WAVES6 : LOSS_WAVES) * 64), 0, st, c.ws->h, c.h_es != HeadsCall::ES_UNSET ? c.h_es : c.max_batch * FEAT, c.params,
                     *c.L, *c.cfg, c.n, actions, old_logps, advs, rets, inv_b, c.ws->dh,
                     c.dh_es != HeadsCall::ES_UNSET ? c.dh_es : c.max_batch * FEAT, c.ws->dlogits, c.ws->dvalue, c.ws->hpart, hs,
                     c.normalise_dh ? c.ws->gsc : nullptr, c.max_batch, c.ws->amax);
  if (large)
    hipLaunchKernelGGL(head_wgrad_kernel<MAXA_LARGE>, dim3(HEAD_WG), dim3(256), 0, st, c.ws->h, c.ws->dlogits, c.n,
                       c.L->A, c.ws->hpart, hs);
  const int nsum = (c.L->A + 1) * FEAT + c.L->A + 1;  // gradient elements; + one workgroup for the three loss sums
  hipLaunchKernelGGL(heads_reduce_kernel, dim3((nsum + RED_OUT - 1) / RED_OUT + 1), dim3(256), 0, st, c.ws->hpart, hs, HEAD_WG, *c.L,
                     *c.cfg, inv_b, grads);
}

void launch_categorical_stats(const float* probs, int n, int A, float* p_hat, float* logits, float* entropy,
                              hipStream_t st) {
  hipLaunchKernelGGL(categorical_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, st, probs, n, A, p_hat, logits,
                     entropy);
}

}  // namespace ddrl
