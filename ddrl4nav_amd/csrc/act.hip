// The three convolutions of an ACTING forward (ddrl_forward, n <= DDRL_ACT_FUSED_MAX) in ONE launch.
//
// Reference: AtariPreNet.forward (nn/atari_encoder.py:25-32) inside ForwardThread.run (server/forward.py:128-149): a few hundred
// samples per call, latency-bound.  The training kernels of conv2.hip tile the BATCH (3-5 samples per workgroup, k loop of 8-16
// barrier pairs); run on 256 samples they are five dependent launches of 15-25 us, each a serial chain on one wave per SIMD.
// Here one workgroup of eight waves owns ONE (sample, encoder) pair from the frame bytes to a3:
//
//   frames (u8, 28 KB)  -> LDS as fp16 rows (exact)                                      56.4 KB
//   conv1  MFMA A = 32 output pixels, B = the 32 output channels (this encoder's weight planes, 32 KB of LDS); wave = two pixel tiles
//          -> bias, leaky, split into two scaled fp16 planes -> LDS [plane][channel][20 rows of pitch 52 B]    66.6 KB
//   conv2  wave = (oc tile, K quarter of 8 input channels) x all three column tiles: its 16 weight fragments come straight from L2
//          into registers (requested at kernel start), so every weight is read ONCE per workgroup; the K quarters are summed
//          through LDS in fixed order -> bias, leaky, per-SAMPLE plane scale (the workgroup's own maximum)
//          -> LDS [plane][k-block][pixel][8 channels] (20.7 KB, over the dead frame rows)
//   conv3  wave = (oc tile, K quarter of 2 k-blocks) x both column tiles, 20 weight fragments from L2, K quarters summed through LDS
//          -> bias, leaky -> a3 in global memory (the dense layer batches over samples: fc2.hip), the sample's maximum -> a3max
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

struct ActF {
  static constexpr int THREADS = 512;
  static constexpr int FR_PITCH = 168, FR_CH = 84 * FR_PITCH, FR_BYTES = 4 * FR_CH;  // fp16 frame rows, no pad (conv2.hip Fwd1B)
  static constexpr int A1_ROW = 52, A1_CH = 20 * A1_ROW, A1_PLANE = 32 * A1_CH;       // conv2.hip Fwd2B: bank-conflict free row pitch
  static constexpr int A1_OFF = FR_BYTES, A1_BYTES = NPL * A1_PLANE;
  static constexpr int BIAS_OFF = A1_OFF + A1_BYTES;                                  // b1[32] b2[64] b3[64]
  static constexpr int MAX_OFF = BIAS_OFF + 160 * 4;                                  // wave maxima: conv2 [0..5], conv3 [8..15]
  static constexpr int W1_OFF = MAX_OFF + 16 * 4, W1_BYTES = 4 * 4 * NPL * 2 * 32 * 16;  // conv1 weight planes of this encoder
  static constexpr int LDS_BYTES = W1_OFF + W1_BYTES;
  // over the frame rows and the a1 planes once they are dead:
  static constexpr int P2_OFF = 0, P2_BYTES = 4 * 2 * 3 * 16 * 256;                   // conv2 partial sums [K quarter][oc tile][column tile][register][lane]
  static constexpr int A2_KB = 81 * 16, A2_PLANE = 8 * A2_KB;                         // conv2.hip Fwd3B with all 8 k-blocks resident
  static constexpr int A2_OFF = 0, A2_BYTES = NPL * A2_PLANE;
  static constexpr int P3_OFF = 32768, P3_BYTES = 4 * 2 * 2 * 16 * 256;               // conv3 partial sums [K quarter][oc tile][column tile][register][lane]
  static_assert(P2_OFF + P2_BYTES <= BIAS_OFF && A2_OFF + A2_BYTES <= P3_OFF && P3_OFF + P3_BYTES <= BIAS_OFF, "aliases stay below the biases");
  static_assert(W1_OFF % 16 == 0 && LDS_BYTES <= 160 * 1024, "one workgroup per CU");
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

template <bool KEEP>
__global__ __launch_bounds__(512) void act_convs_kernel(const ActArgs A) {
  using K = ActF;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = wave_u(), l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.x, e = blockIdx.y, C = A.C, ROWS = 32 * A.NE;
  const float r255 = PIXEL_UNIT / (255.0f * plane_scale(A.amax[amax_idx(AMAX_W1, e)]));
  const float sa1 = plane_scale(A.amax[amax_idx(AMAX_A1, e)]);
  const float inv2 = 1.0f / (sa1 * plane_scale(A.amax[amax_idx(AMAX_W2, e)]));
  const float sw3 = plane_scale(A.amax[amax_idx(AMAX_W3, e)]);
  float* bias = (float*)(lds + K::BIAS_OFF);
  float* wmax = (float*)(lds + K::MAX_OFF);
  // wave roles: conv2 / conv3 MFMA phases: output-channel tile iw, K quarter kq (every weight fragment is read by exactly one wave
  // of the workgroup); their reductions / epilogues: one (oc tile, column tile) per wave
  const int iw = w & 1, kq = w >> 1;

  // ---------------- phase 0: everything that can be requested now is requested now
  // the sample's frame bytes: 1,764 dwords per stacked frame, four per thread and frame
  unsigned fr[4][4];
  {
    const uint8_t* fsrc = A.frames + (size_t)b * ((size_t)C * 7056);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int j = 0; j < 4; ++j) fr[ch][j] = *(const unsigned*)(fsrc + (ch < C ? ch : C - 1) * 7056 + min(tid + 512 * j, 1763) * 4);
  }
  // conv1 weight planes wp1b[channel][ky pair g][plane][lane half][row = e * 32 + oc][kx 8] (optim.hip) -> LDS [..][lane half][oc][kx 8]:
  // 16-byte fragments, 512 per stacked frame
  u4a w1r[4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const int f = tid;  // fragment ((g * NPL + p) * 2 + hi) * 32 + row of channel ch  (4 * NPL * 2 * 32 = 512 at two planes)
    static_assert(4 * NPL * 2 * 32 == 512, "one conv1 weight fragment per thread and stacked frame");
    w1r[ch] = *(const u4a*)(A.wp1b + ((size_t)((ch < C ? ch : C - 1) * (4 * NPL * 2) + (f >> 5)) * ROWS + e * 32 + (f & 31)) * 8);
  }
  // conv2 weight planes wp2b[e][in channel][plane][oc][tap 16]: the wave's 8 input channels x NPL planes, one 16-byte fragment each
  frag8 w2f[8][NPL];
  {
    const unsigned short* w2 = A.wp2b + (size_t)e * (32 * NPL * 64 * 16) + (size_t)((iw * 32 + l31) * 16 + hi * 8);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int p = 0; p < NPL; ++p) w2f[kg][p] = *(const frag8*)(w2 + ((kq * 8 + kg) * NPL + p) * 1024);
  }
  if (tid < 160) {
    const int64_t off = tid < 32 ? A.b1[e] + tid : (tid < 96 ? A.b2[e] + (tid - 32) : A.b3[e] + (tid - 96));
    bias[tid] = A.params[off];
  }
  // frames -> fp16 rows
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 512 * j;
      if (ch < C && idx < 1764) {
        const unsigned v = fr[ch][j];
        *(uint2*)(lds + ch * K::FR_CH + (idx / 21) * K::FR_PITCH + (idx % 21) * 8) = pixel_quad(v);
      }
    }
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) *(u4a*)(lds + K::W1_OFF + (ch * 512 + tid) * 16) = w1r[ch];
  __syncthreads();
  if (DDRL_ACT_STOP == 1) {
    if (lds[tid] == 77 && w2f[0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
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
      abase[ti] = (4 * (P / 20) + hi) * K::FR_PITCH + (P % 20) * 8;
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
          const char* q = lds + ch * K::FR_CH + abase[ti] + g * (2 * K::FR_PITCH);
          const uint2 lo = *(const uint2*)q, up = *(const uint2*)(q + 8);
          const frag8 px = __builtin_bit_cast(frag8, (u4a){lo.x, lo.y, up.x, up.y});
#pragma unroll
          for (int p = NPL - 1; p >= 0; --p) acc[ti] = mfma_planes(px, wf[p], acc[ti]);  // smallest plane first
        }
      }
    }
    // the lane holds output channel l31 at pixels 32 t + 8 q + 4 hi + (0..3): four neighbours of one image row
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
        char* d = lds + K::A1_OFF + l31 * K::A1_CH + (P / 20) * K::A1_ROW + (P % 20) * 2;  // 4-byte aligned (odd rows start at 4 mod 8)
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(lds_pair*)(d + p * K::A1_PLANE) = lds_pair{pa[p], pb[p]};
        if (KEEP) *(f4*)(A.a1 + e * A.a1_es + (int64_t)b * 12800 + l31 * 400 + P) = y;
      }
    }
  }
  __syncthreads();  // a1 planes complete, frame rows dead
  if (DDRL_ACT_STOP == 2) {
    if (lds[K::A1_OFF + tid] == 77 && w2f[0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
    return;
  }

  // ---------------- phase 2: conv2.  wave = (oc tile iw, input channels 8 kq .. + 7), all three column tiles
  {
    f32x16 acc[3];
    int bB[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
      const int c = j * 32 + l31, cc = c < 81 ? c : 0;
      bB[j] = K::A1_OFF + (2 * (cc / 9) + 2 * hi) * K::A1_ROW + 4 * (cc % 9);
    }
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      frag8 bq[3][NPL];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          const char* q = lds + bB[j] + p * K::A1_PLANE + (kq * 8 + kg) * K::A1_CH;
          const lds_pair lo = *(const lds_pair*)q, up = *(const lds_pair*)(q + K::A1_ROW);
          bq[j][p] = __builtin_bit_cast(frag8, (u4a){lo.x, lo.y, up.x, up.y});
        }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = mfma_planes(w2f[kg][PA[t]], bq[j][PB[t]], acc[j]);
    }
    __syncthreads();  // every wave is done with the a1 planes: the partial sums go over them
    float* part = (float*)(lds + K::P2_OFF) + ((kq * 2 + iw) * 3) * (16 * 64) + lane;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(j * 16 + r) * 64] = acc[j][r];
  }
  // conv3 weight planes wp3b[e][k-block][tap pair 5][plane][oc][tap parity][8 channels]: the wave's 2 k-blocks, 20 fragments
  frag8 w3f[2][5][NPL];
  {
    const unsigned short* w3 = A.wp3b + (size_t)e * (8 * 5 * NPL * 64 * 16) + (size_t)((iw * 32 + l31) * 16 + hi * 8);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kg = 0; kg < 5; ++kg)
#pragma unroll
        for (int p = 0; p < NPL; ++p) w3f[kb][kg][p] = *(const frag8*)(w3 + (((2 * kq + kb) * 5 + kg) * NPL + p) * 1024);
  }
  __syncthreads();
  // conv2 epilogue on waves 0..5: output tile (oc tile i2, column tile j2) = the sum of its four K quarters
  const int i2 = w & 1, j2 = w >> 1, c2 = j2 * 32 + l31;
  float y2[16];
  if (w < 6) {
    const float* part = (const float*)(lds + K::P2_OFF) + (i2 * 3 + j2) * (16 * 64) + lane;
    float big = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = part[r * 64];
#pragma unroll
      for (int q = 1; q < 4; ++q) s += part[q * (2 * 3 * 16 * 64) + r * 64];
      const int oc = i2 * 32 + acc_row(r, hi);
      y2[r] = leaky_f(__builtin_fmaf(s, inv2, bias[32 + oc]));
      if (c2 < 81) {
        big = fmaxf(big, fabsf(y2[r]));
        if (KEEP) A.a2[e * A.a2_es + (int64_t)b * 5184 + oc * 81 + c2] = y2[r];
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, 64));
    if (lane == 0) wmax[w] = big;
  }
  __syncthreads();
  float m2 = wmax[0];
#pragma unroll
  for (int i = 1; i < 6; ++i) m2 = fmaxf(m2, wmax[i]);
  const float sa2 = plane_scale(m2), inv3 = 1.0f / (sa2 * sw3);
  if (w < 6 && c2 < 81) {
    // the lane holds channels 32 i2 + 8 q + 4 hi + (0..3) of pixel c2: half a 16-byte channel-innermost fragment
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned pa[NPL], pb[NPL];
      split_planes(y2[4 * q], y2[4 * q + 1], sa2, pa);
      split_planes(y2[4 * q + 2], y2[4 * q + 3], sa2, pb);
      char* d = lds + K::A2_OFF + (i2 * 4 + q) * K::A2_KB + c2 * 16 + hi * 8;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::A2_PLANE) = make_uint2(pa[p], pb[p]);
    }
  }
  __syncthreads();
  if (DDRL_ACT_STOP == 3) {
    if (lds[tid] == 77 && w3f[0][0][0][0] == (_Float16)3.0f) A.a3max[0] = 1.0f;
    return;
  }

  // ---------------- phase 3: conv3.  wave = (oc tile iw, k-blocks 2 kq, 2 kq + 1), both column tiles
  {
    f32x16 acc[2];
    int bB[2], tapoff[5];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
      const int c = j * 32 + l31, cc = c < 49 ? c : 0;
      bB[j] = K::A2_OFF + ((cc / 7) * 9 + cc % 7) * 16;
    }
#pragma unroll
    for (int kg = 0; kg < 5; ++kg) {
      const int tap = min(2 * kg + hi, 8);  // the tenth tap re-reads tap 8 against zero weights
      tapoff[kg] = ((tap / 3) * 9 + tap % 3) * 16;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kg = 0; kg < 5; ++kg) {
        frag8 bq[2][NPL];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < NPL; ++p) bq[j][p] = *(const frag8*)(lds + bB[j] + tapoff[kg] + p * K::A2_PLANE + (2 * kq + kb) * K::A2_KB);
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = mfma_planes(w3f[kb][kg][PA[t]], bq[j][PB[t]], acc[j]);
      }
    float* part = (float*)(lds + K::P3_OFF) + ((kq * 2 + iw) * 2) * (16 * 64) + lane;  // clear of the a2 planes other waves still read
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(j * 16 + r) * 64] = acc[j][r];
  }
  __syncthreads();
  // conv3 epilogue: wave = (oc tile i3, column tile j3, register half h3)
  {
    const int i3 = w & 1, j3 = (w >> 1) & 1, h3 = w >> 2, c3 = j3 * 32 + l31;
    const float* part = (const float*)(lds + K::P3_OFF) + (i3 * 2 + j3) * (16 * 64) + lane;
    float big = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = 8 * h3 + rr;
      float s = part[r * 64];
#pragma unroll
      for (int q = 1; q < 4; ++q) s += part[q * (2 * 2 * 16 * 64) + r * 64];
      const int oc = i3 * 32 + (rr & 3) + 8 * (2 * h3 + (rr >> 2)) + 4 * hi;  // acc_row(r, hi)
      const float y = leaky_f(__builtin_fmaf(s, inv3, bias[96 + oc]));
      if (c3 < 49) {
        A.a3[e * A.a3_es + (int64_t)b * FLAT + oc * 49 + c3] = y;
        big = fmaxf(big, fabsf(y));
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
  using K = ActF;
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
