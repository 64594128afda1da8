// The three convolutions of an ACTING forward (ddrl_forward, n <= DDRL_ACT_FUSED_MAX) in ONE launch.
//
// Reference: AtariPreNet.forward (nn/atari_encoder.py:25-32) inside ForwardThread.run (server/forward.py:128-149): a few hundred
// samples per call, latency-bound.  The training kernels of conv2.hip tile the BATCH (3-5 samples per workgroup, k loop of 8-16
// barrier pairs); run on 256 samples they are five dependent launches of 15-25 us, each a serial chain on one wave per SIMD.
// Here one workgroup of eight waves owns ONE (sample, encoder) pair from the frame bytes to a3, on 62 KB of LDS and <= 128 registers,
// so that TWO workgroups share a CU (256 environments x 2 encoders = 512 workgroups = one round on 256 CUs; LDS = frames + conv1 weights
// + biases = 61.7 KB):
//
//   frames (u8, 28 KB)  -> LDS as they are; a fragment's eight bytes become fp16 operands when it is read (1024 + b is 0x6400 | b:
//          one v_perm_b32 and one v_pk_add_f16 per pixel pair, exact)
//   conv1  32 x 32 x 16 MFMA, A = 32 output pixels, B = the 32 output channels (this encoder's weight planes, 32 KB of LDS); wave = two
//          pixel tiles, kept in registers until every wave is done with the frames and the conv1 weights
//          -> bias, leaky, split into two scaled fp16 planes -> LDS [plane][channel][20 rows of pitch 44 B]  56.3 KB, OVER frames + weights
//   conv2  16 x 16 x 32 MFMA: wave = (32 output channels = two tiles, one HALF of K, three of the six column tiles); its weight
//          fragments come straight from L2 into registers, streamed in groups; the two K halves meet through 24 KB of LDS (each wave
//          hands over the output-channel tile its partner finishes) -> bias, leaky, per-SAMPLE plane scale (the workgroup's own maximum)
//          -> LDS [plane][k-block][pixel][8 channels] (20.7 KB, over the dead a1 planes)
//   conv3  the same roles on two of the four column tiles, hand-over through 16 KB
//          -> bias, leaky -> a3 in global memory (the dense layer batches over samples: fc2.hip), the sample's maximum -> a3max
//
// How the roles were chosen (round 6, profiles/r06_act_convs_ab.txt; the round-2 kernel: one workgroup per CU on 155 KB, 32 x 32 tiles,
// K quarters summed through 98 + 65 KB of LDS, 36-37 us per 256-environment forward): a forward is a chain of short phases, bound in
// turn by the weight bytes its 512 workgroups pull out of L2 and by LDS reads of the activation fragments.  A wave with ALL of K and
// half the column tiles reads every weight fragment twice and every activation fragment four times (311 MB from L2: 34 us); (16
// channels, K half, all column tiles) reads the weights once but the activations four times (LDS-bound, 4 LDS cycles per 3 matrix
// cycles: 33 us); (32 channels, K half, half the columns) reads both twice: 31 us.  Deeper weight prefetch (2 -> 4 groups) and a
// start-up skew between the two resident workgroups change nothing / cost time.
//
// The arithmetic is that of the training kernels (f16x2 for conv1, f16x3 for conv2 / conv3, fp32 accumulation, smallest plane
// products first); only the plane scale of a2 is the sample's instead of the batch's, and the summation order differs, so the
// results agree with them to fp32 rounding, not bit for bit.  a1 / a2 stay on chip; KEEP = true (ddrl_debug_keep_activations) also
// stores them for the tests that look at them.
#include "engine2.h"

#ifndef DDRL_ACT_STOP
#define DDRL_ACT_STOP 0  // timing-only: 1 = return after the staging barrier, 2 = after conv1, 3 = after conv2 (results are WRONG)
#endif

namespace ddrl {

#ifdef DDRL_PLANES_BF16
// three bf16 planes per operand do not fit this kernel's LDS budget: DDRL_ACT_FUSED_MAX is 0 in that build (kernels.h) and nothing calls in here
void launch_act_convs(const EncCall&, hipStream_t) { abort(); }
#else

using u4a = __attribute__((ext_vector_type(4))) unsigned;
struct __attribute__((packed, aligned(4))) lds_pair {
  unsigned x, y;
};

struct ActArgs {
  const uint8_t* frames;
  const unsigned short *wp1b, *wp2b, *wp3b;
  const float *amax, *params;
  int64_t b1[2], b2[2], b3[2];  // bias offsets in params, per encoder
  float *a1, *a2, *a3;
  int64_t a1_es, a2_es, a3_es;
  float* a3max;  // [e][a3max_es]: largest |a3| of every sample
  int a3max_es, n, C, NE;
};

struct ActG {
  static constexpr int THREADS = 512;
  static constexpr int FR_CH = 7056, FR_ROW = 84, FR_BYTES = 4 * FR_CH;                 // frame bytes as they are
  static constexpr int W1_OFF = FR_BYTES, W1_BYTES = 4 * 4 * NPL * 2 * 32 * 16;          // conv1 weight planes of this encoder
  // a1 rows of 20 pixels = 40 B at pitch 44: measured best of 40 / 44 / 48 / 52 (the training kernel's, conflict-free for ITS 32-lane column
  // tiles) / 56 / 60 for the 16-lane tiles of this kernel's conv2 (30.6 us against 31.3-33.4, profiles/r06_act_convs_ab.txt)
  static constexpr int A1_ROW = 44, A1_CH = 20 * A1_ROW, A1_PLANE = 32 * A1_CH, A1_OFF = 0, A1_BYTES = NPL * A1_PLANE;
  static constexpr int BIAS_OFF = A1_BYTES > W1_OFF + W1_BYTES ? A1_BYTES : W1_OFF + W1_BYTES, MAX_OFF = BIAS_OFF + 160 * 4, LDS_BYTES = MAX_OFF + 16 * 4;
  static constexpr int A2_KB = 81 * 16, A2_PLANE = 8 * A2_KB, A2_OFF = 0;                // conv2.hip Fwd3B with all 8 k-blocks resident
  static constexpr int GK = 2;                                     // k-steps per weight group (16 registers per k-step: two tiles x two planes)
  static constexpr int NB = 2;                                     // groups in registers: the one in use + NB - 1 requested (3, 4: no gain)
  static constexpr int NS2 = 8, NS3 = 10;                          // k-steps of a wave: conv2 (2 input channels each), conv3 (k-block pair x tap pair)
  static constexpr int P2_OFF = 0, P2_BYTES = 4 * 2 * 3 * 4 * 256;  // conv2 hand-over [wave][column tile][register][lane]: over the dead a1 planes
  static constexpr int P3_OFF = 24576, P3_BYTES = 4 * 2 * 2 * 4 * 256;  // conv3 hand-over: clear of the a2 planes
  static_assert(NS2 % GK == 0 && NS3 % GK == 0, "whole groups");
  static_assert(P2_OFF + P2_BYTES <= BIAS_OFF && NPL * A2_PLANE <= P3_OFF && P3_OFF + P3_BYTES <= BIAS_OFF, "hand-over buffers");
  static_assert(W1_OFF % 16 == 0 && W1_OFF + W1_BYTES <= BIAS_OFF && NPL * A2_PLANE <= BIAS_OFF, "aliases stay below the biases");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ f4 mfma16_planes(frag8 a, frag8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// four frame bytes -> four fp16 operands: 0x6400 | b is 1024 + b, the subtraction is exact
__device__ __forceinline__ uint2 bytes_to_f16(unsigned v) {
  using h2v = __attribute__((ext_vector_type(2))) _Float16;
  const h2v k = {(_Float16)1024.0f, (_Float16)1024.0f};
  const h2v lo = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, v, 0x04010400u)) - k;
  const h2v hi = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, v, 0x04030402u)) - k;
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}

template <bool KEEP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void act_convs_kernel(const ActArgs A) {
  using K = ActG;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = wave_u(), l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.x, e = blockIdx.y, C = A.C, ROWS = 32 * A.NE;
  const float r255 = PIXEL_UNIT / (255.0f * plane_scale(A.amax[amax_idx(AMAX_W1, e)]));
  const float sa1 = plane_scale(A.amax[amax_idx(AMAX_A1, e)]);
  const float inv2 = 1.0f / (sa1 * plane_scale(A.amax[amax_idx(AMAX_W2, e)]));
  const float sw3 = plane_scale(A.amax[amax_idx(AMAX_W3, e)]);
  float* bias = (float*)(lds + K::BIAS_OFF);
  float* wmax = (float*)(lds + K::MAX_OFF);
  // conv2 / conv3 roles: output channels 32 mp .. (two tiles of 16), K half kh, column half nh (three / two tiles of 16 pixels); after
  // the hand-over a wave finishes the tiles of output channels 16 mt ..
  const int mp = w & 1, kh = (w >> 1) & 1, nh = w >> 2, mt = 2 * mp + kh;

  // ---------------- phase 0: requests
  unsigned fr[4][4];
  {
    const uint8_t* fsrc = A.frames + (size_t)b * ((size_t)C * 7056);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[ch][j] = *(const unsigned*)(fsrc + (ch < C ? ch : C - 1) * 7056 + min(tid + 512 * j, 1763) * 4);
  }
  u4a w1r[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const int f = tid;
    static_assert(4 * NPL * 2 * 32 == 512, "one conv1 weight fragment per thread and stacked frame");
    w1r[ch] = *(const u4a*)(A.wp1b + ((size_t)((ch < C ? ch : C - 1) * (4 * NPL * 2) + (f >> 5)) * ROWS + e * 32 + (f & 31)) * 8);
  }
  // conv2 weights wp2b[e][in channel][plane][oc][tap 16]: k-step ks = channels 2 ks, 2 ks + 1; lane chunk q4 = (channel parity, tap half)
  const unsigned short* w2 = A.wp2b + (size_t)e * (32 * NPL * 64 * 16) + (size_t)(((q4 >> 1) * NPL * 64 + 32 * mp + l15) * 16 + (q4 & 1) * 8);
  frag8 w2f[K::NB][K::GK][2][NPL];
  auto fetch2 = [&](int g, frag8 (&dst)[K::GK][2][NPL]) {  // k-steps NS2 kh + GK g ..; [oc tile of the pair][plane]
#pragma unroll
    for (int i = 0; i < K::GK; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          dst[i][m][p] = *(const frag8*)(w2 + ((size_t)(2 * (K::NS2 * kh + g * K::GK + i)) * NPL + p) * 1024 + m * 256);
  };
#pragma unroll
  for (int g = 0; g < K::NB - 1; ++g) fetch2(g, w2f[g]);
  if (tid < 160) {
    const int64_t off = tid < 32 ? A.b1[e] + tid : (tid < 96 ? A.b2[e] + (tid - 32) : A.b3[e] + (tid - 96));
    bias[tid] = A.params[off];
  }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 512 * j;
      if (ch < C && idx < 1764) *(unsigned*)(lds + ch * K::FR_CH + idx * 4) = fr[ch][j];
    }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) *(u4a*)(lds + K::W1_OFF + (ch * 512 + tid) * 16) = w1r[ch];
  __syncthreads();
  if (DDRL_ACT_STOP == 1) {
    if (lds[tid] == 77 && w2f[0][0][0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
    return;
  }

  // ---------------- phase 1: conv1.  wave w: pixel tiles w and w + 8 (13 tiles of 32 cover the 400 output pixels)
  {
    f32x16 acc[2];
    int abase[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][r] = 0.0f;
      const int P = min((w + 8 * ti) * 32 + l31, 399);
      abase[ti] = (4 * (P / 20) + hi) * K::FR_ROW + (P % 20) * 4;
    }
    const bool two = w + 8 < 13;
    const char* wl = lds + K::W1_OFF + (hi * 32 + l31) * 16;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch >= C) break;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        frag8 wf[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) wf[p] = *(const frag8*)(wl + ((ch * 4 + g) * NPL + p) * 1024);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          if (ti == 1 && !two) continue;
          const lds_pair by = *(const lds_pair*)(lds + ch * K::FR_CH + abase[ti] + g * (2 * K::FR_ROW));  // eight pixels of row 4 y + 2 g + hi
          const uint2 lo = bytes_to_f16(by.x), up = bytes_to_f16(by.y);
          const frag8 px = __builtin_bit_cast(frag8, (u4a){lo.x, lo.y, up.x, up.y});
#pragma unroll
          for (int p = NPL - 1; p >= 0; --p) acc[ti] = mfma_planes(px, wf[p], acc[ti]);  // smallest plane first
        }
      }
    }
    __syncthreads();  // every wave is done with the frame bytes and the conv1 weights: the a1 planes go over them
    const float b1v = bias[l31];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (ti == 1 && !two) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int P = (w + 8 * ti) * 32 + 8 * q + 4 * hi;
        if (P >= 400) continue;
        f4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = leaky_f(__builtin_fmaf(acc[ti][4 * q + k], r255, b1v));
        unsigned pa[NPL], pb[NPL];
        split_planes(y[0], y[1], sa1, pa);
        split_planes(y[2], y[3], sa1, pb);
        char* d = lds + K::A1_OFF + l31 * K::A1_CH + (P / 20) * K::A1_ROW + (P % 20) * 2;
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(lds_pair*)(d + p * K::A1_PLANE) = lds_pair{pa[p], pb[p]};
        if (KEEP) *(f4*)(A.a1 + e * A.a1_es + (int64_t)b * 12800 + l31 * 400 + P) = y;
      }
    }
  }
  __syncthreads();  // a1 planes complete
  if (DDRL_ACT_STOP == 2) {
    if (lds[K::A1_OFF + tid] == 77 && w2f[0][0][0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
    return;
  }

  // conv3 weights wp3b[e][k-block][tap pair 5][plane][oc][tap parity][8 channels]: k-step s of the wave = (k-block pair 2 kh + s / 5,
  // tap pair s % 5); lane chunk q4 = (k-block parity, tap parity)
  const unsigned short* w3 = A.wp3b + (size_t)e * (8 * 5 * NPL * 64 * 16) + (size_t)(((q4 >> 1) * 5 * NPL * 64 + 32 * mp + l15) * 16 + (q4 & 1) * 8);
  frag8 w3f[K::NB][K::GK][2][NPL];
  auto fetch3 = [&](int g, frag8 (&dst)[K::GK][2][NPL]) {
#pragma unroll
    for (int i = 0; i < K::GK; ++i) {
      const int st = g * K::GK + i, kp = 2 * kh + st / 5, kg = st % 5;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < NPL; ++p) dst[i][m][p] = *(const frag8*)(w3 + ((size_t)((2 * kp) * 5 + kg) * NPL + p) * 1024 + m * 256);
    }
  };

  // ---------------- phase 2: conv2.  wave = output channels 32 mp .. (two tiles), input channels 16 kh .., column tiles 3 nh .. 3 nh + 2
  float y2[3][4];
  {
    f4 acc[2][3];
    int bB[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      acc[0][j] = acc[1][j] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      const int c = (3 * nh + j) * 16 + l15, cc = c < 81 ? c : 0;
      // chunk q4: input channel parity q4 >> 1, window rows 2 (q4 & 1), + 1
      bB[j] = K::A1_OFF + (q4 >> 1) * K::A1_CH + (2 * (cc / 9) + 2 * (q4 & 1)) * K::A1_ROW + 4 * (cc % 9);
    }
    auto steps2 = [&](int g, const frag8 (&wg)[K::GK][2][NPL]) {
#pragma unroll
      for (int i = 0; i < K::GK; ++i) {
        const int ks = K::NS2 * kh + g * K::GK + i;
        frag8 bq[3][NPL];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            const char* q = lds + bB[j] + p * K::A1_PLANE + (2 * ks) * K::A1_CH;
            const lds_pair lo = *(const lds_pair*)q, up = *(const lds_pair*)(q + K::A1_ROW);
            bq[j][p] = __builtin_bit_cast(frag8, (u4a){lo.x, lo.y, up.x, up.y});
          }
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[m][j] = mfma16_planes(wg[i][m][PA[t]], bq[j][PB[t]], acc[m][j]);
      }
    };
#pragma unroll
    for (int g = 0; g < K::NS2 / K::GK; ++g) {
      if (g + K::NB - 1 < K::NS2 / K::GK) fetch2(g + K::NB - 1, w2f[(g + K::NB - 1) % K::NB]);
      if (g + K::NB - 1 >= K::NS2 / K::GK && g + K::NB - 1 - K::NS2 / K::GK < K::NB - 1)  // the first conv3 groups ride on the tail of conv2
        fetch3(g + K::NB - 1 - K::NS2 / K::GK, w3f[g + K::NB - 1 - K::NS2 / K::GK]);
      steps2(g, w2f[g % K::NB]);
    }
    __syncthreads();  // every wave is done with the a1 planes: the hand-over goes over them
    // the wave finishes output channels 16 mt .. (tile kh of its pair) and hands the other tile's three column tiles to its partner
    // (mp, 1 - kh, nh) = wave w ^ 2
    {
      float* part = (float*)(lds + K::P2_OFF) + (w * 3) * (4 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = acc[0][j][r], v1 = acc[1][j][r];
          part[(j * 4 + r) * 64] = kh ? v0 : v1;
        }
    }
    __syncthreads();
    float big = 0.0f;
    {
      const float* part = (const float*)(lds + K::P2_OFF) + ((w ^ 2) * 3) * (4 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = (3 * nh + j) * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float own = kh ? acc[1][j][r] : acc[0][j][r], other = part[(j * 4 + r) * 64];
          const float sum = kh ? other + own : own + other;  // input channels 0..15 first
          const int oc = 16 * mt + 4 * q4 + r;
          y2[j][r] = leaky_f(__builtin_fmaf(sum, inv2, bias[32 + oc]));
          if (c < 81) {
            big = fmaxf(big, fabsf(y2[j][r]));
            if (KEEP) A.a2[e * A.a2_es + (int64_t)b * 5184 + oc * 81 + c] = y2[j][r];
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, 64));
    if (lane == 0) wmax[w] = big;
  }
  __syncthreads();  // the maxima are there, every wave has read its hand-over: the a2 planes go over it
  float m2 = wmax[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) m2 = fmaxf(m2, wmax[i]);
  const float sa2 = plane_scale(m2), inv3 = 1.0f / (sa2 * sw3);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = (3 * nh + j) * 16 + l15;
    if (c < 81) {
      // channels 16 mt + 4 q4 .. + 3 of pixel c: half of the 16-byte channel-innermost fragment of k-block 2 mt + (q4 >> 1)
      unsigned pa[NPL], pb[NPL];
      split_planes(y2[j][0], y2[j][1], sa2, pa);
      split_planes(y2[j][2], y2[j][3], sa2, pb);
      char* d = lds + K::A2_OFF + (2 * mt + (q4 >> 1)) * K::A2_KB + c * 16 + (q4 & 1) * 8;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::A2_PLANE) = make_uint2(pa[p], pb[p]);
    }
  }
  __syncthreads();
  if (DDRL_ACT_STOP == 3) {
    if (lds[tid] == 77 && w3f[0][0][0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
    return;
  }

  // ---------------- phase 3: conv3.  wave = output channels 32 mp .. (two tiles), k-blocks 4 kh .. 4 kh + 3, column tiles 2 nh, 2 nh + 1
  {
    f4 acc[2][2];
    int bB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[0][j] = acc[1][j] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      const int c = (2 * nh + j) * 16 + l15, cc = c < 49 ? c : 0;
      bB[j] = K::A2_OFF + (q4 >> 1) * K::A2_KB + ((cc / 7) * 9 + cc % 7) * 16;
    }
    auto steps3 = [&](int g, const frag8 (&wg)[K::GK][2][NPL]) {
#pragma unroll
      for (int i = 0; i < K::GK; ++i) {
        const int st = g * K::GK + i, kp = 2 * kh + st / 5, kg = st % 5;
        const int tap = min(2 * kg + (q4 & 1), 8);  // the tenth tap re-reads tap 8 against zero weights
        const int toff = ((tap / 3) * 9 + tap % 3) * 16 + (2 * kp) * K::A2_KB;
        frag8 bq[2][NPL];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < NPL; ++p) bq[j][p] = *(const frag8*)(lds + bB[j] + toff + p * K::A2_PLANE);
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[m][j] = mfma16_planes(wg[i][m][PA[t]], bq[j][PB[t]], acc[m][j]);
      }
    };
#pragma unroll
    for (int g = 0; g < K::NS3 / K::GK; ++g) {
      if (g + K::NB - 1 < K::NS3 / K::GK) fetch3(g + K::NB - 1, w3f[(g + K::NB - 1) % K::NB]);
      steps3(g, w3f[g % K::NB]);
    }
    // the wave finishes tile kh of its pair and hands the other tile's two column tiles to wave w ^ 2 (the buffer is clear of the a2 planes)
    {
      float* part = (float*)(lds + K::P3_OFF) + (w * 2) * (4 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v0 = acc[0][j][r], v1 = acc[1][j][r];
          part[(j * 4 + r) * 64] = kh ? v0 : v1;
        }
    }
    __syncthreads();
    float big = 0.0f;
    {
      const float* part = (const float*)(lds + K::P3_OFF) + ((w ^ 2) * 2) * (4 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = (2 * nh + j) * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float own = kh ? acc[1][j][r] : acc[0][j][r], other = part[(j * 4 + r) * 64];
          const float sum = kh ? other + own : own + other;  // k-blocks 0..3 first
          const int oc = 16 * mt + 4 * q4 + r;
          const float y = leaky_f(__builtin_fmaf(sum, inv3, bias[96 + oc]));
          if (c < 49) {
            A.a3[e * A.a3_es + (int64_t)b * FLAT + oc * 49 + c] = y;
            big = fmaxf(big, fabsf(y));
          }
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, 64));
    if (lane == 0) wmax[8 + w] = big;
  }
  __syncthreads();
  if (tid == 0) {
    float m = wmax[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, wmax[8 + i]);
    A.a3max[e * A.a3max_es + b] = m;
  }
}

void launch_act_convs(const EncCall& c, hipStream_t st) {
  using K = ActG;
  const Workspace& w = *c.ws;
  const ParamLayout& L = *c.L;
  const int64_t MB = c.max_batch;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)act_convs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)act_convs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  ActArgs a{c.frames, w.wp1b, w.wp2b, w.wp3b, w.amax, c.params,
            {L.enc_base[0] + L.enc.c1b, L.enc_base[L.NE - 1] + L.enc.c1b},
            {L.enc_base[0] + L.enc.c2b, L.enc_base[L.NE - 1] + L.enc.c2b},
            {L.enc_base[0] + L.enc.c3b, L.enc_base[L.NE - 1] + L.enc.c3b},
            w.a1, w.a2, w.a3, MB * 12800, MB * 5184, MB * FLAT, w.actmax, DDRL_ACT_FUSED_MAX, c.n, L.C, L.NE};
  ProfRange pr(c.prof, "ActConvs", st);
  if (c.keep_acts)
    hipLaunchKernelGGL(act_convs_kernel<true>, dim3((unsigned)c.n, (unsigned)L.NE), dim3(K::THREADS), K::LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL(act_convs_kernel<false>, dim3((unsigned)c.n, (unsigned)L.NE), dim3(K::THREADS), K::LDS_BYTES, st, a);
}

#endif  // DDRL_PLANES_BF16

}  // namespace ddrl
