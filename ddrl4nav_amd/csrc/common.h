// Shared definitions for the DDRL4NAV hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ddrl.h"

namespace ddrl {

// ---- AtariPreNet geometry (reference USTC_lab/nn/atari_encoder.py:16-21) -------------------
constexpr int IMG = 84;
constexpr int FEAT = 512;   // AC_INPUT_DIM (config_nn.py:23)
constexpr int FLAT = 3136;  // 64*7*7
// conv1: C x84x84 -> 32x20x20, k8 s4 ; conv2: 32x20x20 -> 64x9x9, k4 s2 ; conv3: 64x9x9 -> 64x7x7, k3 s1
constexpr int C1_OC = 32, C1_KS = 8, C1_S = 4, C1_OW = 20, C1_P = 400;
constexpr int C2_IC = 32, C2_OC = 64, C2_KS = 4, C2_S = 2, C2_IW = 20, C2_OW = 9, C2_P = 81, C2_K = 512;
constexpr int C3_IC = 64, C3_OC = 64, C3_KS = 3, C3_S = 1, C3_IW = 9, C3_OW = 7, C3_P = 49, C3_K = 576;
constexpr float LEAKY = 0.01f;  // F.leaky_relu default negative_slope

// ---- flat parameter arena (named_parameters() order, reference nn/base.py:60-66) ----------
struct EncLayout {  // offsets relative to the encoder's base
  int64_t c1w, c1b, c2w, c2b, c3w, c3b, lw, lb, size;
};
struct ParamLayout {
  int A, C;
  int NE;  // encoders: 2 = actor.pre + critic.pre (SHARE_CNN_NET=False), 1 = one shared prenet
  EncLayout enc;
  int64_t enc_base[2];  // [0]=actor.pre, [1]=critic.pre  (both = prenet when shared)
  int64_t actor_w, actor_b, critic_w, critic_b;
  int64_t n_params, n_actor;
};

// named_parameters() order of the reference's PPO module (ppo.py:26-28): prenet (when shared,
// runner/utils.py:136-143), then actor.* (its own pre first), then critic.*.
inline ParamLayout make_layout(int A, int C, bool shared = false) {
  ParamLayout L;
  L.A = A;
  L.C = C;
  L.NE = shared ? 1 : 2;
  int64_t o = 0;
  L.enc.c1w = o; o += (int64_t)C1_OC * C * 64;
  L.enc.c1b = o; o += C1_OC;
  L.enc.c2w = o; o += (int64_t)C2_OC * C2_K;
  L.enc.c2b = o; o += C2_OC;
  L.enc.c3w = o; o += (int64_t)C3_OC * C3_K;
  L.enc.c3b = o; o += C3_OC;
  L.enc.lw = o; o += (int64_t)FEAT * FLAT;
  L.enc.lb = o; o += FEAT;
  L.enc.size = o;
  int64_t p = 0;
  L.enc_base[0] = p; p += L.enc.size;
  L.actor_w = p; p += (int64_t)A * FEAT;
  L.actor_b = p; p += A;
  L.n_actor = p;
  L.critic_w = p; p += FEAT;
  L.critic_b = p; p += 1;
  if (shared) {
    L.enc_base[1] = L.enc_base[0];
    L.n_params = p;
    L.n_actor = p;  // one Adam group (ppo.py:39)
    return L;
  }
  L.enc_base[1] = p; p += L.enc.size;
  L.n_params = p;
  return L;
}

// ---- device workspace carved out of the caller's buffer ------------------------------------
struct Workspace {
  // derived weight layouts, [e] major
  // 16-bit weight planes, NPL per tensor (engine2.h: two scaled fp16 planes by default).  conv1, conv2.hip conv_fwd1_planes_kernel:
  // [channel 4][ky pair 4][plane NPL][lane half 2][row 32 NE][kx 8] 16-bit
  unsigned short* wp1b;
  // conv2 weights as planes [e][in channel 32][plane NPL][oc 64][tap 16] (conv2.hip conv_fwd2_planes_kernel)
  unsigned short* wp2b;
  // conv3 weights as planes [e][k-block 8][tap pair 5][plane NPL][oc 64][tap parity 2][channel 8] (conv_fwd3_planes_kernel)
  unsigned short* wp3b;
  // conv2 weights for the data gradient as planes [e][row parity 2][k-block 8][u 2][plane NPL][(c, ic) 64][v 2][oc 8]
  unsigned short* wd2b;
  // conv3 weights for the data gradient as planes [e][k-block 4 (16 oc)][tap 9][plane][ic 64][oc half 2][oc 8]
  unsigned short* wd3b;
  // dense-layer weights as planes [e][plane NPL][512][3136] (fc2.hip fc_fwd_planes_kernel)
  unsigned short* wlb;
  // the same planes transposed, [e][plane NPL][3136][512] (fc2.hip fc_dgrad_planes_kernel)
  unsigned short* wdlb;
  float* amax;  // [AMAX_SLOTS][2 encoders], see AMAX_* below
  float* actmax;  // [2 encoders][512]: largest |a3| per sample of the last fused acting forward (act.hip)
  // activations (post leaky-relu) and their gradients, [e][max_batch][...]
  float *a1, *a2, *a3, *h;
  // sign bits of a1 (1 = NOT positive: the leaky slope applies), written by conv1's forward for the leaky-ReLU mask of the conv1 weight gradient (which would
  // otherwise re-read all of a1 for its signs): [e][column] 32-bit words, column = sample * 400 + output pixel, bit m1_bit(oc) of
  // the word = output channel oc (the order in which an MFMA lane holds its 16 accumulator rows, so that the producer shifts
  // the bits in as it walks its registers and stores its half-word).  m1_words(max_batch) words per encoder.
  unsigned* m1;
  // sign bits of a2 for conv3's data gradient, whose epilogue has conv2's forward tile layout (column = (sample, pixel), lane half
  // hi, accumulator register r <-> channel i * 32 + acc_row(r, hi)): [e][sample][pixel 81][hi 2] words, bit 16 i + 15 - r set =
  // that channel's a2 is not positive.  2.7 GB of a2 reads become 85 MB.
  unsigned* m2;
  // sign bits of a3 for the dense layer's data gradient, in conv3's forward tile layout: [e][sample][pixel 49][hi 2] words, bit
  // 16 i + 15 - r as above (channel = i * 32 + acc_row(r, hi))
  unsigned* m3;
  float *dz1, *dz2, *dz3, *dh;
  // per-sample power-of-two scale of the backward, [e][max_batch]: the data-gradient chain runs on NORMALISED gradients
  // (dh / dz3 / dz2 / dz1 hold g_s^-1 x the true per-sample gradient, g_s = 2^floor(log2 max|dh_s|), written by dh_normalise_kernel in
  // encoder.hip), so that every sample keeps the full 22 bits of the fp16 planes whatever its advantage; the weight-gradient
  // kernels multiply g_s back in while they stage a sample (exact: a power of two)
  float* gsc;
  float* dlogits;  // [max_batch][A] (diagnostics / tests)
  float* dvalue;   // [max_batch]
  float* wpart;    // split-K partial slabs for the weight gradients
  int64_t wpart_floats;
  float* hpart;    // heads partials [HEAD_WG][HPART]
  double* npart;   // grad-norm partials [NORM_WG]
  float* lut;      // [256] float32(u8/255.0)
  int64_t total_bytes;
};

// ---- running |max| of the tensors that are split into scaled fp16 planes (engine2.h "plane scheme") --------------------
// amax[slot][encoder]: the weight slots (and the bound of a1, below) are refreshed by pack_weights; the activation slots are zeroed at the start of every
// forward (ddrl_forward, ddrl_ppo_iter, ddrl_encoder_forward) and raised by the conv epilogues (atomic max); the gradient
// slots are zeroed by launch_encoder_backward, which measures dh and lets the data-gradient epilogues raise dz3 / dz2.
constexpr int AMAX_WL = 0, AMAX_W2 = 1, AMAX_W3 = 2, AMAX_W1 = 3, AMAX_A1 = 4, AMAX_A2 = 5, AMAX_A3 = 6, AMAX_DH = 7, AMAX_DZ3 = 8,
              AMAX_DZ2 = 9, AMAX_DZ1 = 10, AMAX_GMAX = 11, AMAX_SLOTS = 12, AMAX_FIRST_ACT = AMAX_A2;
// The gradient slots hold the maxima of the NORMALISED tensors (Workspace::gsc); AMAX_GMAX = the largest per-sample scale g_s of
// the batch: a weight gradient stages sample s with the factor  plane_scale(slot) x g_s / g_max  (<= plane_scale(slot): no
// overflow) and multiplies its sums by g_max / (the two plane scales).
// A weight gradient stages sample s at  plane_scale(normalised maximum) x g_s / g_max: the batch's largest RAW element lands at
// or below [2^12, 2^13) -- below when the sample with the largest g_s is not the one with the largest normalised gradient (a few
// binades at most).  Two of the three spare binades under fp16's 65,504 go back into the scale (top < 2^15), so that the
// absolute error floor of the planes (2^-25 of their unit) does not rise against the per-tensor scheme.
constexpr float WGRAD_HEADROOM = 4.0f;
constexpr int GSC_EXP_MIN = -60, GSC_EXP_MAX = 60;  // clamp of log2 g_s: keeps plane_scale / g_max and g_max / scales inside fp32
// AMAX_A1 is not measured: pixels / 255 lie in [0, 1], so |a1[oc]| <= sum_k |w1[oc][k]| + |b1[oc]|; pack_weights stores the largest
// such bound with the weight slots (a few times the measured maximum: two of the sixteen spare binades of the fp16 planes),
// which spares conv1's epilogue a maximum per output and an atomic per wave
__host__ __device__ inline int amax_idx(int slot, int e) { return slot * 2 + e; }
// power-of-two scale that puts a tensor whose largest magnitude is m into [2^12, 2^13)
__host__ __device__ inline float f16_scale(float m) {
  if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
  int e;
  frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 13 - e);
}
constexpr int HEAD_WG = 256;   // workgroups of the heads/loss kernel (fixed -> deterministic)
constexpr int NORM_WG = 1024;  // workgroups of the grad-norm kernel

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
// words of the a1 sign mask per encoder (one per column), and the bit of output channel oc inside a column's word: lane half
// hi = (oc >> 2) & 1 of the 32x32 MFMA tile holds oc as accumulator register r = (oc & 3) + 4 (oc >> 3) and shifts its 16 signs in
// first register first, i.e. register r ends at bit 15 - r of half-word hi
__host__ __device__ inline int64_t m1_words(int64_t max_batch) { return max_batch * 400; }
__host__ __device__ inline int m1_bit(int oc) { return ((oc >> 2) & 1) * 16 + 15 - ((oc & 3) + 4 * (oc >> 3)); }

#ifdef DDRL_PLANES_BF16
#define DDRL_ACT_FUSED_MAX 0    // three bf16 planes do not fit the fused kernel's LDS budget: every acting launch takes the batch-tiled kernels
#elif !defined(DDRL_ACT_FUSED_MAX)
#define DDRL_ACT_FUSED_MAX 512  // acting launches of at most this many samples (Workspace::actmax is carved for it)
#endif
// split counts for the weight-gradient GEMMs (fixed per context -> deterministic sums)
#ifndef DDRL_FC_ACT_SPLITS
#define DDRL_FC_ACT_SPLITS 14  // split-K factor of the dense layer's forward in acting launches: a divisor of its 98 k-blocks
#endif
// the fused acting kernel (act.hip) leaves a3's scale as per-sample maxima in Workspace::actmax, which only the SPLIT launch of the
// dense forward reads (fc2.hip); an unsplit acting launch would read a stale AMAX_A3 slot
static_assert(DDRL_FC_ACT_SPLITS > 1 && 98 % DDRL_FC_ACT_SPLITS == 0, "DDRL_FC_ACT_SPLITS: a divisor of 98 larger than 1");
#ifndef DDRL_C1_SPLITS
#define DDRL_C1_SPLITS 512  // two workgroups per CU in ONE round: 3.37 ms against 3.66 at 768 / 1024 / 1536, 4.1 at 256 / 384 / 640 (profiles/README.md)
#endif
#ifndef DDRL_C2_SPLITS
#define DDRL_C2_SPLITS 256
#endif
#ifndef DDRL_C3_SPLITS
#define DDRL_C3_SPLITS 256
#endif
struct Splits {
  int c1, c2, c3, fc;
};
inline Splits choose_splits(int max_batch, int NE = 2) {
  // ~1024 workgroups per weight-gradient launch (2 resident per CU x 256 CUs x 2 rounds)
  const int pairs = (max_batch + 1) / 2;
  auto cap = [&](int want, int limit) { return want < limit ? (want < 1 ? 1 : want) : (limit < 1 ? 1 : limit); };
  Splits s;
  s.c1 = cap(DDRL_C1_SPLITS, pairs);      // 1 column tile, encoders fused
  const int k = 2 / NE;         // one encoder: twice the splits keep the same number of workgroups
  s.c2 = cap(DDRL_C2_SPLITS * k, pairs);   // 2 column tiles x 2 encoders
  s.c3 = cap(DDRL_C3_SPLITS * k, pairs);   // conv_wgrad3_planes_kernel: one workgroup per (split, encoder), up to two per CU
  s.fc = cap(5 * k, (max_batch + 31) / 32);  // 25 x 4 tiles x 2 encoders
  return s;
}

inline int64_t hpart_stride(int A) { return align_up((int64_t)(A + 1) * FEAT + A + 1 + 4, 64); }

inline int64_t carve(Workspace& w, const ddrl_config& c, void* base) {
  const int64_t MB = c.max_batch, C = c.in_channels, A = c.n_actions;
  int64_t off = 0;
  auto take = [&](int64_t floats) -> float* {
    float* p = base ? (float*)((char*)base + off) : nullptr;
    off += align_up(floats * 4, 256);
    return p;
  };
  w.wp1b = (unsigned short*)take(4 * 4 * 3 * 2 * 64 * 8 / 2);
  w.wlb = (unsigned short*)take(2 * 3 * (int64_t)FLAT * FEAT / 2);
  w.wdlb = (unsigned short*)take(2 * 3 * (int64_t)FLAT * FEAT / 2);
  w.amax = take(64);
  w.actmax = take(2 * (DDRL_ACT_FUSED_MAX > 512 ? DDRL_ACT_FUSED_MAX : 512));  // [2 encoders][DDRL_ACT_FUSED_MAX]
  w.wp2b = (unsigned short*)take(2 * 32 * 3 * 64 * 16 / 2);
  w.wp3b = (unsigned short*)take(2 * 8 * 5 * 3 * 64 * 16 / 2);
  w.wd2b = (unsigned short*)take(2 * 2 * 4 * 4 * 3 * 64 * 16 / 2);
  w.wd3b = (unsigned short*)take(2 * 8 * 5 * 3 * 64 * 16 / 2);
  w.a1 = take(2 * MB * 32 * 400);
  w.m1 = (unsigned*)take(2 * m1_words(MB));
  w.m2 = (unsigned*)take(2 * MB * 81 * 2);
  w.m3 = (unsigned*)take(2 * MB * 49 * 2);
  w.a2 = take(2 * MB * 64 * 81);
  w.a3 = take(2 * MB * FLAT);
  w.h = take(2 * MB * FEAT);
  w.dz1 = take(2 * MB * 32 * 400);
  w.dz2 = take(2 * MB * 64 * 81);
  w.dz3 = take(2 * MB * FLAT);
  w.dh = take(2 * MB * FEAT);
  w.gsc = take(2 * MB);
  w.dlogits = take(MB * A);
  w.dvalue = take(MB);
  Splits s = choose_splits(c.max_batch, c.share_cnn_net ? 1 : 2);
  // slabs hold weights followed by bias, like the arena
  int64_t p1 = (int64_t)s.c1 * 2 * (32 * C * 64 + 32);
  int64_t p2 = (int64_t)s.c2 * 2 * (64 * 512 + 64);
  int64_t p3 = (int64_t)s.c3 * 2 * (64 * 576 + 64);
  int64_t pf = (int64_t)s.fc * 2 * ((int64_t)FEAT * FLAT + FEAT);
  int64_t pm = p1;
  if (p2 > pm) pm = p2;
  if (p3 > pm) pm = p3;
  if (pf > pm) pm = pf;
  // the acting path parks the split-K partial sums of the FC forward here (fc_forward_splits)
  const int64_t pact = (int64_t)DDRL_FC_ACT_SPLITS * 2 * (MB < 1024 ? MB : 1024) * FEAT;
  if (pact > pm) pm = pact;
  w.wpart_floats = pm;
  w.wpart = take(pm);
  w.hpart = take((int64_t)HEAD_WG * hpart_stride(A));
  w.npart = (double*)take(NORM_WG * 2);
  w.lut = take(256);
  w.total_bytes = off;
  return off;
}

// counter-based uniform in [0,1) of the samplers: identical to ddrl4nav_amd/utils/recipe.py:sample_uniform.
// All 64 bits of the stream id take part (callers put rollout / draw counters in the high bits).
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline float hash_uniform(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t base = splitmix64(splitmix64(seed) ^ stream);
  uint64_t z = splitmix64(base + idx);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

}  // namespace ddrl
