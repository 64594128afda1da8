// Convolution kernels of the Atari encoder on the 16-bit matrix pipe: the three forwards and the data gradients of conv3 / conv2,
// each "direct convolution from an LDS image".
//
// Instead of gathering an im2col tile, a workgroup stages the RAW input planes it needs (coalesced 16-byte loads, every element
// fetched once per k-block, split into fp16 planes on the way into LDS: engine2.h "plane scheme") and every MFMA operand is one
// 16-byte LDS read (or two 8-byte reads) at  lane_base + immediate :
//     lane_base  = position of the lane's output pixel inside the staged planes (+ what the lane half adds)
//     immediate  = offset of the k-group's (channel, taps)                      (compile-time constant)
// Weights arrive pre-split into planes, in exactly the order the k index walks (optim.hip).
//
// Reference arithmetic: F.conv2d + F.leaky_relu (USTC_lab/nn/atari_encoder.py:26-28) and the autograd data-gradients of conv3 /
// conv2 (ppo.py:122-123).  Acting launches of at most DDRL_ACT_FUSED_MAX samples do not come here: act.hip.
#include "engine2.h"

namespace ddrl {


// ================================================================================================
// conv1 forward on the 16-bit matrix pipe, fp32-accurate.  The input pixels are integers 0..255 and therefore EXACT in
// fp16 (and bf16); the weights come as NPL planes from optim.hip (two scaled fp16 planes whose sum reproduces them to 22
// bits; -DDDRL_PLANES_BF16: three bf16 planes, 24 bits).  Every product plane x pixel is exact in fp32, the MFMA
// accumulates in fp32, and the 1/255 of the reference's frame normalisation (and the planes' scale) is applied once to the
// sum:  z = (sum_k (W0 + W1)[k] x[k]) / (255 S) + b.  Two v_mfma_f32_32x32x16_f16 (32 cycles each) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each).
//   rows = (e, oc), cols = b*400 + pix (256 per workgroup), k-block = one input channel = 4 k-groups of 16:
//   k = (ky = 2 g + h, kx = j): lane half h picks the image row, the 8 elements of a fragment are 8 consecutive
//   pixels of that row (stride-4 convolution: x = 4 ox + kx), i.e. one 16-byte LDS read per operand.
// The image is kept as 16-bit rows (pitch 176 B) in natural pixel order.
// ================================================================================================
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using u4v = __attribute__((ext_vector_type(4))) unsigned;
using bf4 = __attribute__((ext_vector_type(4))) __bf16;

// Timing-only knock-outs (-DDDRL_F1_KO=bits; results are WRONG): 1 no a1 stores, 2 no epilogue at all, 4 no MFMAs, 8 no image
// conversion / LDS writes, 16 no weight copies
#ifndef DDRL_F1_KO
#define DDRL_F1_KO 0
#endif
#ifndef DDRL_F1_LDS_PAD
#define DDRL_F1_LDS_PAD 0  // A/B: 256 restores the two-workgroups-per-CU footprint
#endif
#ifndef DDRL_F1_PITCH
#define DDRL_F1_PITCH 168
#endif
template <int NE>
struct Fwd1B {
  // image row pitch 168 B = the 84 pixels of a row, no pad: a 32-pixel column tile reads 20 pixels of one row and 12 of the row 4 below,
  // whose banks (4 x 168 B = 40 words further) follow the first 20 pixels without overlap, and 2 x (16 KB + 10.5 KB) lets THREE workgroups
  // share a CU (at pitch 176 the third missed by 1.4 KB)
  static constexpr int ROWS = 32 * NE, A_BYTES = 4 * NPL * 2 * ROWS * 16, PITCH = DDRL_F1_PITCH, IMG_BYTES = 64 * PITCH;
  static constexpr int STAGE_BYTES = A_BYTES + IMG_BYTES, AQ = A_BYTES / 16, NAJ = AQ / 256;  // weight quads per thread
  // 54,272 B = 106 allocation units of 512 B: THREE workgroups share a CU's 160 KB (the bias used to sit behind the stages: 54,528 B
  // -> 107 units -> two workgroups).  The epilogue reads the bias from stage 0, which is free by then.
  static constexpr size_t LDS_BYTES = 2 * STAGE_BYTES + DDRL_F1_LDS_PAD;
};

template <int NE>
__global__ __launch_bounds__(256) void conv_fwd1_planes_kernel(const uint8_t* __restrict__ frames, const unsigned short* __restrict__ wp1b,
                                                               float* __restrict__ amax, const float* __restrict__ params, int64_t bias_off0,
                                                               int64_t bias_off1, float* __restrict__ out, int64_t out_es, int n,
                                                               unsigned* __restrict__ m1, int64_t m1_es, int C) {
  using K = Fwd1B<NE>;
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  // tile geometry: 256 output pixels of at most two samples and the input rows they need
  const int c0 = blockIdx.x * 256, ctot = n * 400, clast = min(c0 + 255, ctot - 1);
  const int b0 = c0 / 400, b1 = clast / 400, oyf0 = (c0 % 400) / 20, iy0_start = 4 * oyf0;
  int nrows0, nrows1 = 0;
  if (b1 == b0) {
    nrows0 = 4 * ((clast % 400) / 20 - oyf0) + 8;
  } else {
    nrows0 = 84 - iy0_start;
    nrows1 = 4 * ((clast % 400) / 20) + 8;
  }
  const int nd0 = nrows0 * 21, nd_total = nd0 + nrows1 * 21;
  const int64_t src0 = (int64_t)b0 * (C * 7056) + iy0_start * 84, src1 = (int64_t)b1 * (C * 7056);  // C stacked frames per sample (1..4)
  // this kernel opens every forward: it also resets the running maxima that the conv2 / conv3 epilogues raise afterwards (a
  // separate 16-byte memset is a kernel of its own: 5 of the ~100 us of an acting step)
  if (blockIdx.x == 0 && tid < (AMAX_DH - AMAX_FIRST_ACT) * 2) amax[amax_idx(AMAX_FIRST_ACT, 0) + tid] = 0.0f;
  int64_t imsrc[6];
  int imdst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int idx = tid + 256 * j;
    imsrc[j] = (idx >= nd_total) ? src0 : (idx < nd0 ? src0 + idx * 4 : src1 + (idx - nd0) * 4);
    imdst[j] = K::A_BYTES + (idx / 21) * K::PITCH + (idx % 21) * 8;  // 4 pixels -> 4 halfwords
  }
  int abase[NE], bbase[2];
#pragma unroll
  for (int i = 0; i < NE; ++i) abase[i] = (hi * K::ROWS + i * 32 + l31) * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int c = c0 + wc * 64 + j * 32 + l31;
    if (c >= ctot) c = c0;
    const int b = c / 400, pix = c % 400, oy = pix / 20, ox = pix % 20;
    const int lr = (b == b0) ? (4 * oy - iy0_start) : (nrows0 + 4 * oy);
    bbase[j] = K::A_BYTES + (lr + hi) * K::PITCH + ox * 8;
  }
  // the frame bytes of all four channels are requested at once (24 dwords per thread): with 48 short MFMAs per
  // k-block a one-block prefetch distance does not cover an HBM round trip
  unsigned imreg[4][6];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < 6; ++j) imreg[ch][j] = *(const unsigned*)(frames + (ch < C ? ch : C - 1) * 7056 + imsrc[j]);  // unconditional, clamped
  // the weight planes of a channel are a plain copy of global memory: LDS-direct, one k-block ahead (L2-resident)
  const int wave = wave_u();
  uint32_t woff[K::NAJ];
#pragma unroll
  for (int j = 0; j < K::NAJ; ++j) woff[j] = (uint32_t)((tid + 256 * j) * 16);
  auto stage_w = [&](int ch, char* st) {
    if (DDRL_F1_KO & 16) return;
    direct_copy((const char*)wp1b + (size_t)ch * K::A_BYTES, woff, (float*)st, wave, tid, K::AQ);
  };
  auto commit_img = [&](char* st, int ch) {
    if (DDRL_F1_KO & 8) return;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (tid + 256 * j < nd_total) {
        const unsigned v = imreg[ch][j];
        *(uint2*)(st + imdst[j]) = pixel_quad(v);
      }
    }
  };
  f32x16 acc[NE][2];
#pragma unroll
  for (int i = 0; i < NE; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  stage_w(0, ldsb);
  commit_img(ldsb, 0);
  wait_vmcnt<0>();
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    if (ch >= C) break;  // wave-uniform: fewer stacked frames = fewer k-blocks (the last one run reads stage (C - 1) & 1)
    const char* cur = ldsb + (ch & 1) * K::STAGE_BYTES;
    char* nxt = ldsb + ((ch + 1) & 1) * K::STAGE_BYTES;
    if (ch + 1 < C) stage_w(ch + 1, nxt);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      frag8 b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // 8-byte aligned: two 8-byte reads
        const uint2 lo = *(const uint2*)(cur + bbase[j] + g * (2 * K::PITCH));
        const uint2 hi4 = *(const uint2*)(cur + bbase[j] + g * (2 * K::PITCH) + 8);
        b[j] = __builtin_bit_cast(frag8, (u4v){lo.x, lo.y, hi4.x, hi4.y});
      }
#pragma unroll
      for (int p = NPL - 1; p >= 0; --p) {  // smallest plane first
        frag8 a[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) a[i] = *(const frag8*)(cur + abase[i] + ((g * NPL + p) * 2) * K::ROWS * 16);
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DDRL_F1_KO & 4) acc[i][j][0] += (float)a[i][0] + (float)b[j][0];
            else acc[i][j] = mfma_planes(a[i], b[j], acc[i][j]);
          }
      }
    }
    if (ch + 1 < C) {
      commit_img(nxt, ch + 1);
      wait_vmcnt<0>();
    }
    __syncthreads();
  }
  float* bias = (float*)ldsb;  // both stages are free: every wave has passed the loop's final barrier
  if (tid < K::ROWS) bias[tid] = params[(tid >> 5 ? bias_off1 : bias_off0) + (tid & 31)];
  __syncthreads();
  float r255[NE];  // 1/255 of the frame normalisation and the scale of the encoder's weight planes
#pragma unroll
  for (int i = 0; i < NE; ++i) r255[i] = PIXEL_UNIT / (255.0f * plane_scale(amax[amax_idx(AMAX_W1, i)]));
  // (the scale of a1's planes comes from a bound that pack_weights derives from the weights, common.h AMAX_A1: no maximum here)
  if (DDRL_F1_KO & 2) {
    float sum = 0.0f;
    for (int i = 0; i < NE; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 1.2345f) out[tid] = sum;
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + wc * 64 + j * 32 + l31;
    if (c >= ctot) continue;
    const int b = c / 400, pix = c % 400;
    const uint32_t lanep = (uint32_t)((b * 12800 + pix + hi * (4 * 400)) * 4);
#pragma unroll
    for (int i = 0; i < NE; ++i) {  // i = encoder
      float* base = out + i * out_es;
      // sign mask of a1 (common.h Workspace::m1; bit SET = not positive): the lane shifts the signs of its 16 output channels into
      // a half-word
      unsigned bits = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = acc_row(r, hi);
        const float y = leaky_f(__builtin_fmaf(acc[i][j][r], r255[i], bias[i * 32 + oc]));  // one rounding less than mul + add
        if (!(DDRL_F1_KO & 1) || y == 1.2345f) st1_so(base + acc_row(r, 0) * 400, lanep, y);
        // y > 0 <=> its bit pattern, as a signed integer, is >= 1 <=> (pattern -sat 1) has a clear sign; alignbit shifts that sign
        // in: two VALU instructions per output (compare + select + or: three and two s_nop)
        bits = __builtin_amdgcn_alignbit(bits, (unsigned)__builtin_elementwise_sub_sat((int)__float_as_uint(y), 1), 31);
      }
      if (m1 != nullptr) ((unsigned short*)(m1 + i * m1_es))[2 * (int64_t)c + hi] = (unsigned short)bits;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv1 forward of a TRAINING launch with the weight planes resident in LDS (-DDDRL_F1_RESIDENT=0: the kernel above for every launch).
// conv_fwd1_planes_kernel copies a stacked frame's planes (16 KB) from L2 into LDS for every 256 output pixels: 6.5 GB per launch at
// B = 65,536, and its timing knock-outs put 0.6 of the kernel's 2.7 ms there.  Here one workgroup per CU keeps all four frames' planes
// (64 KB) for the whole launch and walks over the tiles: eight waves = two groups of four, each group with the tile geometry, image
// rows (4 x 10.5 KB, all four stacked frames staged at once) and epilogue of the kernel above; the groups share the barriers (they
// run the same phases), the next tile's frame bytes are in flight while the present one feeds the matrix pipe.
// ------------------------------------------------------------------------------------------------
#ifndef DDRL_F1_RESIDENT
#ifdef DDRL_PLANES_BF16
#define DDRL_F1_RESIDENT 0  // three planes per operand: the resident weights (96 KB) and two image groups exceed the LDS
#else
#define DDRL_F1_RESIDENT 1
#endif
#endif
template <int NE>
struct Fwd1R {
  static constexpr int ROWS = 32 * NE, A_BYTES = 4 * NPL * 2 * ROWS * 16, PITCH = DDRL_F1_PITCH, IMG_BYTES = 64 * PITCH;
  static constexpr int W_BYTES = 4 * A_BYTES, IMG_OFF = W_BYTES, GROUP_IMG = 4 * IMG_BYTES;   // per group: four stacked frames
  static constexpr int BIAS_OFF = IMG_OFF + 2 * GROUP_IMG;
  static constexpr size_t LDS_BYTES = BIAS_OFF + ROWS * 4;
  static_assert(!DDRL_F1_RESIDENT || LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

template <int NE>
__global__ __launch_bounds__(512) void conv_fwd1_resident_kernel(const uint8_t* __restrict__ frames, const unsigned short* __restrict__ wp1b,
                                                                 float* __restrict__ amax, const float* __restrict__ params, int64_t bias_off0,
                                                                 int64_t bias_off1, float* __restrict__ out, int64_t out_es, int n,
                                                                 unsigned* __restrict__ m1, int64_t m1_es, int C) {
  using K = Fwd1R<NE>;
  extern __shared__ __attribute__((aligned(16))) char ldsr[];
  const int tid = threadIdx.x, gq = tid >> 8, lt = tid & 255, lane = tid & 63, wc = (tid >> 6) & 3, l31 = lane & 31, hi = lane >> 5;
  char* img = ldsr + K::IMG_OFF + gq * K::GROUP_IMG;
  float* bias = (float*)(ldsr + K::BIAS_OFF);
  const int ctot = n * 400, ntiles = (ctot + 255) / 256;
  // this kernel opens every training forward: it resets the running maxima that the conv2 / conv3 epilogues raise afterwards
  if (blockIdx.x == 0 && tid < (AMAX_DH - AMAX_FIRST_ACT) * 2) amax[amax_idx(AMAX_FIRST_ACT, 0) + tid] = 0.0f;
  // weight planes of the C stacked frames: a plain copy of global memory, 1,024 quads per frame
  for (int q = tid; q < C * (K::A_BYTES / 16); q += 512) *(f4*)(ldsr + q * 16) = *(const f4*)((const char*)wp1b + (size_t)q * 16);
  if (tid < K::ROWS) bias[tid] = params[(tid >> 5 ? bias_off1 : bias_off0) + (tid & 31)];
  float r255[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) r255[i] = PIXEL_UNIT / (255.0f * plane_scale(amax[amax_idx(AMAX_W1, i)]));
  int abase[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) abase[i] = (hi * K::ROWS + i * 32 + l31) * 16;

  // tile geometry (as in the kernel above): 256 output pixels of at most two samples and the input rows they need
  struct Geo {
    int c0, b0, iy0_start, nrows0, nd0, nd_total;
    int64_t src0, src1;
  };
  auto geometry = [&](int t) {
    Geo g;
    g.c0 = t * 256;
    const int clast = min(g.c0 + 255, ctot - 1);
    g.b0 = g.c0 / 400;
    const int b1 = clast / 400, oyf0 = (g.c0 % 400) / 20;
    g.iy0_start = 4 * oyf0;
    int nrows1 = 0;
    if (b1 == g.b0) {
      g.nrows0 = 4 * ((clast % 400) / 20 - oyf0) + 8;
    } else {
      g.nrows0 = 84 - g.iy0_start;
      nrows1 = 4 * ((clast % 400) / 20) + 8;
    }
    g.nd0 = g.nrows0 * 21;
    g.nd_total = g.nd0 + nrows1 * 21;
    g.src0 = (int64_t)g.b0 * (C * 7056) + g.iy0_start * 84;
    g.src1 = (int64_t)b1 * (C * 7056);
    return g;
  };
  unsigned imreg[4][6];
  auto request = [&](const Geo& g) {  // the frame bytes of all four stacked frames of the group's tile: 24 dwords per thread
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int idx = lt + 256 * j;
      const int64_t o = (idx >= g.nd_total) ? g.src0 : (idx < g.nd0 ? g.src0 + idx * 4 : g.src1 + (idx - g.nd0) * 4);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) imreg[ch][j] = *(const unsigned*)(frames + (ch < C ? ch : C - 1) * 7056 + o);
    }
  };
  int t = 2 * (int)blockIdx.x + gq;
  Geo g = geometry(min(t, ntiles - 1));
  request(g);
  __syncthreads();  // weights and bias in place
  for (; 2 * (t / 2) < ntiles; t += 2 * (int)gridDim.x) {  // both groups of a workgroup leave together: the barriers below are shared
    const bool live = t < ntiles;
    // ---- frame bytes -> fp16 rows of the group's image (the registers hold this tile's bytes)
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int idx = lt + 256 * j;
        if (ch < C && idx < g.nd_total) *(uint2*)(img + ch * K::IMG_BYTES + (idx / 21) * K::PITCH + (idx % 21) * 8) = pixel_quad(imreg[ch][j]);
      }
    int bbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int c = g.c0 + wc * 64 + j * 32 + l31;
      if (c >= ctot) c = g.c0;
      const int b = c / 400, pix = c % 400, oy = pix / 20, ox = pix % 20;
      const int lr = (b == g.b0) ? (4 * oy - g.iy0_start) : (g.nrows0 + 4 * oy);
      bbase[j] = (lr + hi) * K::PITCH + ox * 8;
    }
    const Geo cur = g;
    __syncthreads();  // image rows visible
    // ---- the next tile's bytes travel while this one is multiplied
    const int tn = t + 2 * (int)gridDim.x;
    if (2 * (tn / 2) < ntiles) {
      g = geometry(min(tn, ntiles - 1));
      request(g);
    }
    f32x16 acc[NE][2];
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch >= C) break;
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) {
        frag8 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* q = img + ch * K::IMG_BYTES + bbase[j] + gk * (2 * K::PITCH);
          const uint2 lo = *(const uint2*)q, hi4 = *(const uint2*)(q + 8);
          b[j] = __builtin_bit_cast(frag8, (u4v){lo.x, lo.y, hi4.x, hi4.y});
        }
#pragma unroll
        for (int p = NPL - 1; p >= 0; --p) {  // smallest plane first
          frag8 a[NE];
#pragma unroll
          for (int i = 0; i < NE; ++i) a[i] = *(const frag8*)(ldsr + ch * K::A_BYTES + abase[i] + ((gk * NPL + p) * 2) * K::ROWS * 16);
#pragma unroll
          for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(a[i], b[j], acc[i][j]);
        }
      }
    }
    __syncthreads();  // every wave is done with the image rows: the next tile may overwrite them
    if (!live) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cur.c0 + wc * 64 + j * 32 + l31;
      if (c >= ctot) continue;
      const int b = c / 400, pix = c % 400;
      const uint32_t lanep = (uint32_t)((b * 12800 + pix + hi * (4 * 400)) * 4);
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        float* base = out + i * out_es;
        unsigned bits = 0u;  // sign mask of a1 (common.h Workspace::m1), as in the kernel above
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int oc = acc_row(r, hi);
          const float y = leaky_f(__builtin_fmaf(acc[i][j][r], r255[i], bias[i * 32 + oc]));
          st1_so(base + acc_row(r, 0) * 400, lanep, y);
          bits = __builtin_amdgcn_alignbit(bits, (unsigned)__builtin_elementwise_sub_sat((int)__float_as_uint(y), 1), 31);
        }
        if (m1 != nullptr) ((unsigned short*)(m1 + i * m1_es))[2 * (int64_t)c + hi] = (unsigned short)bits;
      }
    }
  }
}

template <int NE>
static void launch_fwd1_planes(const EncCall& c, bool acting, hipStream_t st) {
  const Workspace& w = *c.ws;
  const ParamLayout& L = *c.L;
  if (DDRL_F1_RESIDENT != 0 && !acting && (int64_t)c.n * 400 >= 256 * 2048) {  // at least ~8 tiles per workgroup: the 64 KB weight copy pays
    using R = Fwd1R<NE>;
    static bool configured_r = false;
    static int cus = 256;
    if (!configured_r) {
      (void)hipFuncSetAttribute((const void*)conv_fwd1_resident_kernel<NE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::LDS_BYTES);
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      configured_r = true;
    }
    hipLaunchKernelGGL(conv_fwd1_resident_kernel<NE>, dim3((unsigned)cus), dim3(512), R::LDS_BYTES, st, c.frames, w.wp1b, w.amax, c.params,
                       L.enc_base[0] + L.enc.c1b, L.enc_base[NE - 1] + L.enc.c1b, w.a1, c.max_batch * 12800, c.n, w.m1, m1_words(c.max_batch), L.C);
    return;
  }
  using K = Fwd1B<NE>;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_fwd1_planes_kernel<NE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_fwd1_planes_kernel<NE>, dim3((unsigned)(((int64_t)c.n * 400 + 255) / 256)), dim3(256), K::LDS_BYTES, st, c.frames,
                     w.wp1b, w.amax, c.params, L.enc_base[0] + L.enc.c1b, L.enc_base[NE - 1] + L.enc.c1b, w.a1, c.max_batch * 12800, c.n,
                     acting ? (unsigned*)nullptr : w.m1, m1_words(c.max_batch), L.C);
}

// ================================================================================================
// conv2 forward on the 16-bit matrix pipe, fp32-accurate (plane products, see fc2.hip fc_fwd_planes_kernel): the weights come
// as NPL planes from optim.hip (wp2b), the a1 values are split into NPL planes while they are staged, and the
// six plane products that reach 2^-18 of the largest are accumulated in fp32.  One MFMA k-group (16) = the 4 x 4 taps
// of ONE input channel: lane (pixel, h) holds taps (ky = 2h, kx = 0..3) and (ky = 2h + 1, kx = 0..3) = two runs of four
// consecutive 16-bit values in the staged image (4-byte aligned 8-byte LDS reads), so the im2col stays implicit.
// Tile = 64 output channels x 3 whole samples (243 columns in 8 column tiles of 32), k-block = 2 input channels (DDRL_F2B_KC);
// LDS holds ONE stage (image planes [plane][sample][channel][20 rows of pitch 26] 16-bit + weight planes [channel][plane][oc][16]):
// the next k-block waits in registers and is split / committed between two barriers while the CU's other
// workgroups compute.
// ================================================================================================
// Build knobs (A/B and timing-only knock-outs, see profiles/README.md v16): DDRL_F2B_ROW = image row pitch in bytes (40 = dense,
// 52 = bank-conflict free), DDRL_F2B_KO = 1 drops the residual-plane arithmetic, 2 the global loads inside the k loop
// (both give WRONG results, only their kernel times mean something).
#ifndef DDRL_F2B_ROW
#define DDRL_F2B_ROW 52
#endif
#ifndef DDRL_F2B_KO
#define DDRL_F2B_KO 0
#endif
#ifndef DDRL_F2B_KC
#define DDRL_F2B_KC 2  // input channels per k-block: 2 -> 146 VGPRs and 31 KB of LDS, three workgroups per CU (3.66 vs 3.83 ms at 4 / two)
#endif
#ifndef DDRL_F2B_WPE
#define DDRL_F2B_WPE 3  // waves per SIMD the register budget is cut for
#endif
#ifndef DDRL_F2B_STAGES
#define DDRL_F2B_STAGES 1  // 2 = double-buffered LDS, one barrier per k-block: measured 2.40 against 2.335 ms (same box, f16 planes)
#endif
struct Fwd2B {
  static constexpr int SPT = 3, KC = DDRL_F2B_KC;                 // samples per tile, input channels per k-block
  // image row pitch 26 halfwords (13 words): the 5 input-row pairs a 32-pixel column tile reads in one instruction start
  // 26 words apart = banks {0, 26, 52, 14, 40} + c, ten words each, disjoint (pitch 20: 2-way conflicts, 46 % of LDS cycles)
  static constexpr int ROW = DDRL_F2B_ROW, CH = 20 * ROW, IMG_PLANE = SPT * KC * CH;  // 12,480 B at KC = 4
  static constexpr int W_OFF = NPL * IMG_PLANE, W_BYTES = KC * NPL * 64 * 32;
  static constexpr int BIAS_OFF = W_OFF + W_BYTES;
  static constexpr int NIU = SPT * KC * 100, NIJ = (NIU + 255) / 256;      // image units of 4 pixels, per thread
  static constexpr int NWJ = W_BYTES / 16 / 256;                          // weight quads per thread (6)
  // ONE LDS stage, the next k-block committed between two barriers while the CU's other workgroups compute.  With two planes
  // instead of three a second stage fits (2 x 20.7 KB, still three workgroups per CU; -DDDRL_F2B_STAGES=2: the next k-block is
  // committed into the other stage while this one feeds the matrix pipe, one barrier per k-block) -- and loses 3 %: vector-ALU
  // work does not overlap the matrix pipe of its own SIMD (tools/mfma16_peak.hip), so the commit costs the same either way and
  // the second stage only takes LDS from the neighbours.
  static constexpr int STAGES = DDRL_F2B_STAGES, STAGE_BYTES = BIAS_OFF;
  static constexpr size_t LDS_BYTES = STAGES * STAGE_BYTES + 64 * 4;
};
struct __attribute__((packed, aligned(4))) lds_u2 {
  unsigned x, y;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DDRL_F2B_WPE, DDRL_F2B_WPE))) void conv_fwd2_planes_kernel(const float* __restrict__ a1, int64_t a1_es, const unsigned short* __restrict__ wp2b,
                                                               float* __restrict__ amax, const float* __restrict__ params, int64_t bias_off0,
                                                               int64_t bias_off1, float* __restrict__ out, int64_t out_es, int n,
                                                               unsigned* __restrict__ m2) {
  using K = Fwd2B;
  extern __shared__ __attribute__((aligned(16))) char ldsc2[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int e = blockIdx.z, b0 = blockIdx.x * K::SPT;
  const float sa = plane_scale(amax[amax_idx(AMAX_A1, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_W2, e)]));
  if (tid < 64) ((float*)(ldsc2 + K::STAGES * K::STAGE_BYTES))[tid] = params[(e ? bias_off1 : bias_off0) + tid];
  // ---- staging maps.  image unit u = tid + 256 j: sample u / 400, channel (u % 400) / 100, pixel quad u % 100
  // (row q / 5, quad q % 5).  Missing samples of the last tile read the last sample.
  const float* isrc[K::NIJ];
  int idst[K::NIJ];
#pragma unroll
  for (int j = 0; j < K::NIJ; ++j) {
    const int u = min(tid + 256 * j, K::NIU - 1);
    const int s = u / (100 * K::KC), rem = u % (100 * K::KC), q = rem % 100;
    isrc[j] = a1 + e * a1_es + (int64_t)min(b0 + s, n - 1) * 12800 + rem * 4;  // + kb * 1600
    idst[j] = (s * K::KC + rem / 100) * K::CH + (q / 5) * K::ROW + (q % 5) * 8;
  }
  const unsigned short* wsrc = wp2b + (int64_t)e * (32 * NPL * 64 * 16) + tid * 8;  // + kb * KC * NPL * 1024 + j * 2048
  // ---- operand bases
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = K::W_OFF + (i * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int c = wc * 64 + j * 32 + l31;
    if (c >= K::SPT * 81) c = 0;
    const int s = c / 81, pix = c % 81, oy = pix / 9, ox = pix % 9;
    bB[j] = s * K::KC * K::CH + (2 * oy + 2 * hi) * K::ROW + 4 * ox;
  }
  f4 ir[K::NIJ], wr[K::NWJ];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j) ir[j] = ld4(isrc[j] + kb * (400 * K::KC));
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j) wr[j] = *(const f4*)(wsrc + kb * (K::KC * NPL * 1024) + j * 2048);
  };
  auto commit = [&](char* ldsc2) {  // shadows the kernel's base: the stage to write
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j) {
      if (j + 1 < K::NIJ || tid + 256 * j < K::NIU) {
        const f4 v = ir[j];
        unsigned pa[NPL], pb[NPL];
        split_planes(v.x, v.y, sa, pa);
        split_planes(v.z, v.w, sa, pb);
        char* d = ldsc2 + idst[j];  // 4-byte aligned (odd rows start at 4 mod 8)
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(lds_u2*)(d + p * K::IMG_PLANE) = lds_u2{pa[p], pb[p]};
      }
    }
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j) *(f4*)(ldsc2 + K::W_OFF + (tid + 256 * j) * 16) = wr[j];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  constexpr int NKB = 32 / K::KC;
  auto compute = [&](const char* ldsc2) {  // shadows the kernel's base: the stage to read
#pragma unroll
    for (int kg = 0; kg < K::KC; ++kg) {
      frag8 a[NPL][2], b[NPL][2];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = *(const frag8*)(ldsc2 + aA[i] + (kg * NPL + p) * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const char* q = ldsc2 + bB[j] + p * K::IMG_PLANE + kg * K::CH;
          const lds_u2 lo = *(const lds_u2*)q, up = *(const lds_u2*)(q + K::ROW);
          b[p][j] = __builtin_bit_cast(frag8, (u4v){lo.x, lo.y, up.x, up.y});
        }
      }
      // smallest products first
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(a[PA[t]][i], b[PB[t]][j], acc[i][j]);
    }
  };
  fetch(0);
  commit(ldsc2);
  fetch(1);
  __syncthreads();
  if (K::STAGES == 1) {
    for (int kb = 0; kb < NKB; ++kb) {
      compute(ldsc2);
      __syncthreads();  // every wave is done with the stage
      if (kb + 1 < NKB) {
        commit(ldsc2);
        if (kb + 2 < NKB) fetch(kb + 2);
      }
      __syncthreads();
    }
  } else {
    static_assert(NKB % 2 == 0, "unrolled by the two stages: every LDS address stays register + immediate");
    for (int kb = 0; kb < NKB; kb += 2) {
      compute(ldsc2);                      // stage 0; stage 1 was last read before the previous barrier
      commit(ldsc2 + K::STAGE_BYTES);      // k-block kb + 1 (always exists)
      if (kb + 2 < NKB) fetch(kb + 2);
      __syncthreads();
      compute(ldsc2 + K::STAGE_BYTES);
      if (kb + 2 < NKB) {
        commit(ldsc2);
        if (kb + 3 < NKB) fetch(kb + 3);
      }
      __syncthreads();
    }
  }
  const float* bias = (const float*)(ldsc2 + K::STAGES * K::STAGE_BYTES);
  float big = 0.0f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = wc * 64 + j * 32 + l31;
    const int s = c / 81, pix = c % 81;
    if (c >= K::SPT * 81 || b0 + s >= n) continue;
    float* base = out + e * out_es + (int64_t)b0 * 5184;
    const uint32_t lb = (uint32_t)((s * 5184 + pix + hi * (4 * 81)) * 4);
    unsigned bits = 0u;  // sign mask of a2 (common.h Workspace::m2): the signs of the lane's 32 channels, shifted in register by register
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = i * 32 + acc_row(r, hi);
        const float y = leaky_f(__builtin_fmaf(acc[i][j][r], inv, bias[oc]));
        st1_so(base + (i * 32 + acc_row(r, 0)) * 81, lb, y);
        big = fmaxf(big, fabsf(y));
        bits = __builtin_amdgcn_alignbit(bits, (unsigned)__builtin_elementwise_sub_sat((int)__float_as_uint(y), 1), 31);  // conv1's epilogue
      }
    // i = 0 went in first and now sits in the upper half: swap so that bit 16 i + 15 - r belongs to register r of half i
    if (m2 != nullptr) m2[(e * (out_es / 5184) + b0 + s) * 162 + pix * 2 + hi] = (bits >> 16) | (bits << 16);
  }
  amax_update(big, amax + amax_idx(AMAX_A2, e));
}
static void launch_fwd2_planes(const EncCall& c, bool acting, hipStream_t st) {
  using K = Fwd2B;
  const Workspace& w = *c.ws;
  const ParamLayout& L = *c.L;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_fwd2_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_fwd2_planes_kernel, dim3((unsigned)((c.n + K::SPT - 1) / K::SPT), 1, (unsigned)L.NE), dim3(256), K::LDS_BYTES, st, w.a1,
                     c.max_batch * 12800, w.wp2b, w.amax, c.params, L.enc_base[0] + L.enc.c2b, L.enc_base[L.NE - 1] + L.enc.c2b, w.a2, c.max_batch * 5184,
                     c.n, acting ? (unsigned*)nullptr : w.m2);
}

// ================================================================================================
// conv3 forward as plane products (see conv_fwd2_planes_kernel).  Nine taps per input channel do not fill an MFMA k-group, so the
// k index runs over CHANNELS: the a2 block is staged channel-innermost ([plane][sample][pixel][8 channels] 16-bit, one
// 16-byte fragment per pixel) and one k-group = (two taps) x (8 channels): lane half h reads tap 2 kg + h.  Nine taps =
// 4.5 pairs: the tenth "tap" re-reads tap 8 against zero weights (10 % of the MFMAs).  Tile = 64 output channels x 5
// whole samples (245 columns in 8 column tiles), k-block = 8 input channels = 5 k-groups; one LDS stage, the next
// k-block waits in registers.  Weights: wp3b[e][k-block 8][k-group 5][plane 3][oc 64][h 2][8 channels] (optim.hip).
// ================================================================================================
#ifndef DDRL_F3B_TN
#define DDRL_F3B_TN 2
#endif
#ifndef DDRL_F3B_WPE
#define DDRL_F3B_WPE 3  // waves per SIMD the register budget is cut for: 162 VGPRs, three 50 KB workgroups per CU (2.57 vs 2.69 ms at 2)
#endif
struct Fwd3B {
  // column tiles per wave (2 x 4 fragment tiles = 10 samples per tile measured 3.00 vs 2.70 ms: not a general win); whole samples per tile
  static constexpr int TN = DDRL_F3B_TN, SPT = (128 * TN) / 49, NPX = SPT * 81;
  static constexpr int IMG_PLANE = NPX * 16;                      // 6,480 B
  static constexpr int W_OFF = NPL * IMG_PLANE, W_BYTES = 5 * NPL * 64 * 32;
  static constexpr int BIAS_OFF = W_OFF + W_BYTES;
  static constexpr int NIJ = (NPX + 255) / 256;                   // pixel units per thread (2)
  static constexpr int NWJ = (W_BYTES / 16 + 255) / 256;          // weight quads per thread (8, the last one partial)
  static constexpr size_t LDS_BYTES = BIAS_OFF + 64 * 4;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DDRL_F3B_WPE, DDRL_F3B_WPE))) void conv_fwd3_planes_kernel(const float* __restrict__ a2, int64_t a2_es, const unsigned short* __restrict__ wp3b,
                                                               float* __restrict__ amax, const float* __restrict__ params, int64_t bias_off0,
                                                               int64_t bias_off1, float* __restrict__ out, int64_t out_es, int n,
                                                               unsigned* __restrict__ m3) {
  using K = Fwd3B;
  extern __shared__ __attribute__((aligned(16))) char ldsc3[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int e = blockIdx.z, b0 = blockIdx.x * K::SPT;
  const float sa = plane_scale(amax[amax_idx(AMAX_A2, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_W3, e)]));
  if (tid < 64) ((float*)(ldsc3 + K::BIAS_OFF))[tid] = params[(e ? bias_off1 : bias_off0) + tid];
  // ---- staging maps.  pixel unit u = tid + 256 j: sample u / 81, pixel u % 81; it loads the 8 channels of the k-block
  // (stride 81 floats) and writes one 16-byte fragment per plane.  Missing samples of the last tile read the last one.
  const float* isrc[K::NIJ];
#pragma unroll
  for (int j = 0; j < K::NIJ; ++j) {
    const int u = min(tid + 256 * j, K::NPX - 1);
    const int s = u / 81, px = u % 81;
    isrc[j] = a2 + e * a2_es + (int64_t)min(b0 + s, n - 1) * 5184 + px;  // + (8 kb + c) * 81
  }
  const unsigned short* wsrc = wp3b + (int64_t)e * (8 * 5 * NPL * 64 * 16) + tid * 8;  // + kb * 5 * NPL * 1024 + j * 2048
  // ---- operand bases
  int aA[2], bB[K::TN], tapoff[5];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = K::W_OFF + (i * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    int c = wc * (32 * K::TN) + j * 32 + l31;
    if (c >= K::SPT * 49) c = 0;
    const int s = c / 49, pix = c % 49;
    bB[j] = (s * 81 + (pix / 7) * 9 + pix % 7) * 16;
  }
#pragma unroll
  for (int kg = 0; kg < 5; ++kg) {
    const int tap = min(2 * kg + hi, 8);
    tapoff[kg] = ((tap / 3) * 9 + tap % 3) * 16;
  }
  float ir[K::NIJ][8];
  f4 wr[K::NWJ];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) ir[j][c] = isrc[j][(kb * 8 + c) * 81];
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + 256 * j < K::W_BYTES / 16) wr[j] = *(const f4*)(wsrc + kb * (5 * NPL * 1024) + j * 2048);
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j) {
      if (j + 1 < K::NIJ || tid + 256 * j < K::NPX) {
        unsigned pl[4][NPL];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes(ir[j][2 * c], ir[j][2 * c + 1], sa, pl[c]);
        char* d = ldsc3 + (tid + 256 * j) * 16;
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4v*)(d + p * K::IMG_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
      }
    }
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + 256 * j < K::W_BYTES / 16) *(f4*)(ldsc3 + K::W_OFF + (tid + 256 * j) * 16) = wr[j];
  };
  f32x16 acc[2][K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  constexpr int NKB = 8;
  fetch(0);
  commit();
  fetch(1);
  __syncthreads();
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int kg = 0; kg < 5; ++kg) {
      frag8 a[NPL][2], b[NPL][K::TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = *(const frag8*)(ldsc3 + aA[i] + (kg * NPL + p) * 2048);
#pragma unroll
        for (int j = 0; j < K::TN; ++j) b[p][j] = *(const frag8*)(ldsc3 + bB[j] + tapoff[kg] + p * K::IMG_PLANE);
      }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < K::TN; ++j) acc[i][j] = mfma_planes(a[PA[t]][i], b[PB[t]][j], acc[i][j]);
    }
    __syncthreads();  // every wave is done with the stage
    if (kb + 1 < NKB) {
      commit();
      if (kb + 2 < NKB) fetch(kb + 2);
    }
    __syncthreads();
  }
  const float* bias = (const float*)(ldsc3 + K::BIAS_OFF);
  float big = 0.0f;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    const int c = wc * (32 * K::TN) + j * 32 + l31;
    const int s = c / 49, pix = c % 49;
    if (c >= K::SPT * 49 || b0 + s >= n) continue;
    float* base = out + e * out_es + (int64_t)b0 * FLAT;
    const uint32_t lb = (uint32_t)((s * FLAT + pix + hi * (4 * 49)) * 4);
    unsigned bits = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = i * 32 + acc_row(r, hi);
        const float y = leaky_f(__builtin_fmaf(acc[i][j][r], inv, bias[oc]));
        st1_so(base + (i * 32 + acc_row(r, 0)) * 49, lb, y);
        big = fmaxf(big, fabsf(y));
        bits = __builtin_amdgcn_alignbit(bits, (unsigned)__builtin_elementwise_sub_sat((int)__float_as_uint(y), 1), 31);  // conv1's epilogue
      }
    // sign mask of a3 (common.h Workspace::m3): bit 16 i + 15 - r = register r of half i is not positive
    if (m3 != nullptr) m3[(e * (out_es / FLAT) + b0 + s) * 98 + pix * 2 + hi] = (bits >> 16) | (bits << 16);
  }
  amax_update(big, amax + amax_idx(AMAX_A3, e));
}
static void launch_fwd3_planes(const EncCall& c, bool acting, hipStream_t st) {
  using K = Fwd3B;
  const Workspace& w = *c.ws;
  const ParamLayout& L = *c.L;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_fwd3_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_fwd3_planes_kernel, dim3((unsigned)((c.n + K::SPT - 1) / K::SPT), 1, (unsigned)L.NE), dim3(256), K::LDS_BYTES, st, w.a2,
                     c.max_batch * 5184, w.wp3b, w.amax, c.params, L.enc_base[0] + L.enc.c3b, L.enc_base[L.NE - 1] + L.enc.c3b, w.a3, c.max_batch * FLAT,
                     c.n, acting ? (unsigned*)nullptr : w.m3);
}

// ================================================================================================

// Training launches, and acting launches of more than DDRL_ACT_FUSED_MAX samples (they skip the sign masks only the backward reads)
void launch_conv_forward2(const EncCall& c, bool acting, hipStream_t st) {
  {
    ProfRange pr(c.prof, acting ? "ConvFwd1.act" : "ConvFwd1", st);
    if (c.L->NE == 2) {
      launch_fwd1_planes<2>(c, acting, st);
    } else {
      launch_fwd1_planes<1>(c, acting, st);
    }
  }
  {
    ProfRange pr(c.prof, acting ? "ConvFwd2.act" : "ConvFwd2", st);
    launch_fwd2_planes(c, acting, st);
  }
  {
    ProfRange pr(c.prof, acting ? "ConvFwd3.act" : "ConvFwd3", st);
    launch_fwd3_planes(c, acting, st);
  }
}

// ================================================================================================
// conv3 data gradient as plane products, gather form (see conv_fwd3_planes_kernel, whose mirror image it is):
//   dz2[b][ic][y][x] = leaky'(a2) * sum_{oc,ky,kx} dz3[b][oc][y-ky][x-kx] W3[oc][ic][ky][kx]
// dz3 is staged channel-innermost into zero-bordered 11 x 11 images (data at +2, +2; [plane][sample][pixel][16 oc] 16-bit),
// one MFMA k-group = one tap x 16 oc; tile = 64 ic x 3 whole samples (243 columns, wave w = columns 64 w .., 2 x 2 fragment
// tiles), k-block = 16 oc = 9 k-groups, 4 k-blocks, one LDS stage.
// (Requesting the a2 values of the mask before the last k-block instead of in the epilogue: no gain, 4.37 vs 4.33 ms.)
// It walks all 81 x 9 tap products of which 49 x 9 are non-zero (the scatter form, conv_dgrad3_scatter_kernel, does
// not) and still wins: 192 instead of 512 matrix-pipe cycles per 16 k.
// Weights: wd3b[e][k-block 4][tap 9][plane][ic 64][oc half 2][oc 8] (optim.hip).
// ================================================================================================
#ifndef DDRL_D3B_THREADS
#define DDRL_D3B_THREADS 256  // 512 (eight waves, 6 samples share one copy of the weights): 2.52 vs 2.43 ms -- unlike conv2's data gradient
#endif
#ifndef DDRL_D3B_TN
#define DDRL_D3B_TN 2  // 2 x 2 fragment tiles, 3 samples per tile: 4.19 vs 4.27 ms for 2 x 4 / 6 samples (re-measured under f16 planes: 2.54 vs 2.60)
#endif
struct Dgrad3B {
  // one MFMA k-group = ONE tap x 16 oc (lane half h = oc 8 h .. 8 h + 7): nine k-groups per k-block of 16 oc, no padded tenth tap
  // (tap pairs x 8 oc walked ten: executed / algorithmic 1.84 -> 1.65), the tap shift is a compile-time LDS offset, four k-blocks
  // instead of eight.  Weights: wd3b[e][k-block 4][tap 9][plane NPL][ic 64][oc half 2][oc 8] (optim.hip pack_dgrad3_planes_kernel).
  // four waves / 3 samples per workgroup, two workgroups per CU.  (-DDDRL_D3B_THREADS=512: eight waves / 6 samples share one copy of
  // the k-block's 37 KB of weights, one workgroup per CU: measured slower.)
  static constexpr int THREADS = DDRL_D3B_THREADS, TN = DDRL_D3B_TN, SPT = ((THREADS / 64) * 32 * TN) / 81;  // column tiles per wave, whole samples per tile
  static constexpr int KOC = 16, NKB = 64 / KOC, PIXB = 2 * KOC;  // oc per k-block, k-blocks, bytes per pixel and plane
  static constexpr int IMG_PLANE = SPT * 121 * PIXB;              // 11,616 B
  static constexpr int W_OFF = NPL * IMG_PLANE, W_BYTES = 9 * NPL * 64 * 32;
  static constexpr int NIU = SPT * 49, NIJ = (NIU + THREADS - 1) / THREADS;  // pixel units (16 oc each): 147 -> 1 per thread
  static constexpr int NWQ = W_BYTES / 16, NWJ = (NWQ + THREADS - 1) / THREADS;  // 2,304 weight quads -> 9 per thread
  static constexpr size_t LDS_BYTES = W_OFF + W_BYTES;
};

__global__ __launch_bounds__(Dgrad3B::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dgrad3_planes_kernel(
    const float* __restrict__ dz3, int64_t dz_es, const unsigned short* __restrict__ wd3b, float* __restrict__ amax, const unsigned* __restrict__ m2,
    float* __restrict__ out, int64_t out_es, int n) {
  using K = Dgrad3B;
  extern __shared__ __attribute__((aligned(16))) char ldsd3[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int e = blockIdx.z, b0 = blockIdx.x * K::SPT;
  const float sa = plane_scale(amax[amax_idx(AMAX_DZ3, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_W3, e)]));
  for (int i = tid; i < K::W_OFF / 16; i += K::THREADS) *(f4*)(ldsd3 + i * 16) = zero4();  // images incl. their zero borders
  // ---- staging maps.  unit u = tid + 256 j: sample u / 49, pixel u % 49 -> 16 loads of stride 49 (the k-block's 16 oc)
  const float* isrc[K::NIJ];
  int idst[K::NIJ];
#pragma unroll
  for (int j = 0; j < K::NIJ; ++j) {
    const int u = min(tid + K::THREADS * j, K::NIU - 1);
    const int s = u / 49, px = u % 49;
    isrc[j] = dz3 + e * dz_es + (int64_t)min(b0 + s, n - 1) * FLAT + px;  // + (16 kb + c) * 49
    idst[j] = (s * 121 + (px / 7 + 2) * 11 + px % 7 + 2) * K::PIXB;
  }
  const unsigned short* wsrc = wd3b + (int64_t)e * (K::NKB * 9 * NPL * 64 * 16) + tid * 8;  // + kb * 9 * NPL * 1024 + j * THREADS * 8
  // ---- operand bases
  int aA[2], bB[K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = K::W_OFF + (i * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    int c = wc * (32 * K::TN) + j * 32 + l31;
    if (c >= K::SPT * 81) c = 0;
    const int s = c / 81, pix = c % 81;
    bB[j] = (s * 121 + (pix / 9 + 2) * 11 + pix % 9 + 2) * K::PIXB + hi * 16;
  }
  float ir[K::NIJ][K::KOC];
  f4 wr[K::NWJ];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j)
#pragma unroll
      for (int c = 0; c < K::KOC; ++c) ir[j][c] = isrc[j][(kb * K::KOC + c) * 49];
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + K::THREADS * j < K::NWQ) wr[j] = *(const f4*)(wsrc + kb * (9 * NPL * 1024) + j * (K::THREADS * 8));
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j) {
      if (j + 1 < K::NIJ || tid + K::THREADS * j < K::NIU) {
        unsigned pl[K::KOC / 2][NPL];
#pragma unroll
        for (int c = 0; c < K::KOC / 2; ++c) split_planes(ir[j][2 * c], ir[j][2 * c + 1], sa, pl[c]);
        char* d = ldsd3 + idst[j];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          *(u4v*)(d + p * K::IMG_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
          *(u4v*)(d + p * K::IMG_PLANE + 16) = (u4v){pl[4][p], pl[5][p], pl[6][p], pl[7][p]};
        }
      }
    }
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + K::THREADS * j < K::NWQ) *(f4*)(ldsd3 + K::W_OFF + (tid + K::THREADS * j) * 16) = wr[j];
  };
  f32x16 acc[2][K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  fetch(0);
  __syncthreads();  // zero fill complete
  commit();
  fetch(1);
  __syncthreads();
  for (int kb = 0; kb < K::NKB; ++kb) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {  // tap (ky, kx) = (t / 3, t % 3): source pixel (y - ky, x - kx)
      constexpr int ROW11 = 11;
      const int toff = ((t / 3) * ROW11 + t % 3) * K::PIXB;
      frag8 af[NPL][2], bfr[NPL][K::TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[p][i] = *(const frag8*)(ldsd3 + aA[i] + (t * NPL + p) * 2048);
#pragma unroll
        for (int j = 0; j < K::TN; ++j) bfr[p][j] = *(const frag8*)(ldsd3 + bB[j] - toff + p * K::IMG_PLANE);
      }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int m = 0; m < NPROD; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < K::TN; ++j) acc[i][j] = mfma_planes(af[PA[m]][i], bfr[PB[m]][j], acc[i][j]);
    }
    __syncthreads();  // every wave is done with the stage
    if (kb + 1 < K::NKB) {
      commit();
      if (kb + 2 < K::NKB) fetch(kb + 2);
    }
    __syncthreads();
  }
  // dz2 = leaky'(a2) * sum
  float big = 0.0f;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    const int c = wc * (32 * K::TN) + j * 32 + l31;
    const int s = c / 81, pix = c % 81;
    if (c >= K::SPT * 81 || b0 + s >= n) continue;
    const int64_t off = (int64_t)(b0 + s) * 5184 + pix + hi * (4 * 81);
    float* op = out + e * out_es + off;
    // the signs of a2 at this lane's 32 (channel, pixel) positions: one word written by conv2's forward, whose tile layout
    // this epilogue shares (common.h Workspace::m2) -- not 32 reads of a2
    const unsigned mw = m2[(e * (out_es / 5184) + b0 + s) * 162 + pix * 2 + hi];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float g = leaky_bit(mw, 16 * i + 15 - r, acc[i][j][r] * inv);
        op[(i * 32 + acc_row(r, 0)) * 81] = g;
        big = fmaxf(big, fabsf(g));
      }
  }
  amax_update(big, amax + amax_idx(AMAX_DZ2, e));
}
static void launch_dgrad3_planes(const EncCall& c, hipStream_t st) {
  using K = Dgrad3B;
  const Workspace& w = *c.ws;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_dgrad3_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_dgrad3_planes_kernel, dim3((unsigned)((c.n + K::SPT - 1) / K::SPT), 1, (unsigned)c.L->NE), dim3(K::THREADS), K::LDS_BYTES, st,
                     w.dz3, c.max_batch * FLAT, w.wd3b, w.amax, w.m2, w.dz2, c.max_batch * 5184, c.n);
}

void launch_conv_dgrad3_2(const EncCall& c, hipStream_t st) {
  ProfRange pr(c.prof, "ConvDgrad3", st);
  launch_dgrad3_planes(c, st);
}

// ================================================================================================
// conv2 data gradient as plane products (see conv_fwd2_planes_kernel / conv_fwd3_planes_kernel):
//   da1[b][ic][2p+a][2q+c] = sum_{oc,u,v} dz2[b][oc][p-u][q-v] W2[oc][ic][2u+a][2v+c]
// A workgroup owns one row parity a, both column parities c (rows = (c, ic) = 64) and 5 whole samples
// (cols = (sample, p, q) = 500, wave w = columns 128 w .. 128 w + 127, 2 x 4 fragment tiles per wave).  dz2 is staged
// channel-innermost into zero-bordered 11 x 11 images ([plane][sample][pixel][8 oc] 16-bit, 16 B per pixel): one MFMA
// k-group = (u; v = lane half) x 8 oc, a k-block = 8 oc = 2 k-groups, 8 k-blocks.  One LDS stage; the border stays
// zero because only interior pixels are ever written.  Weights: wd2b[e][a][k-block 8][u 2][plane 3][row 64][v 2][oc 8].
// The output is the raw d(loss)/d(a1): the leaky mask is applied by its only consumer, conv1's weight gradient.
// (A first version with 5-wave workgroups of 3 samples kept only ONE workgroup resident per CU -- SQ_WAVE_CYCLES -- and
// ran at 6.7 ms.)
// ================================================================================================
// Timing-only knock-outs (-DDDRL_D2_KO=bits; results are WRONG): 1 no dz1 stores, 2 no MFMAs, 4 no dz2 loads, 8 no weight loads,
// 16 no dz2 split / LDS writes
#ifndef DDRL_D2_KO
#define DDRL_D2_KO 0
#endif
// ------------------------------------------------------------------------------------------------
// The kernel: ONE workgroup per tile for BOTH row parities: eight waves, rows = (a, c, ic) = 128
// (four 32-row fragment tiles), wave w = columns 64 w .. 64 w + 63 (4 x 2 fragment tiles: the same LDS bytes per MFMA), dz2
// staged ONCE per tile instead of once per parity (the knock-outs put that staging at a third of the two-workgroup kernel).
// ------------------------------------------------------------------------------------------------
struct Dgrad2Both {
  static constexpr int SPT = 5, THREADS = 512, TN = 2;
  static constexpr int IMG_PLANE = SPT * 121 * 16;                      // 9,680 B
  static constexpr int W_HALF = 2 * NPL * 64 * 32;                      // one parity's k-block of weights (8 KB)
  static constexpr int W_OFF = NPL * IMG_PLANE, W_BYTES = 2 * W_HALF;
  static constexpr int NIU = SPT * 81;                                  // 405 pixel units (8 oc each): threads 0 .. 404
  static constexpr int NWQ = W_BYTES / 16, NWJ = NWQ / THREADS;         // 1,024 weight quads -> 2 per thread
  static constexpr size_t LDS_BYTES = W_OFF + W_BYTES;
};

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dgrad2_both_kernel(
    const float* __restrict__ dz2, int64_t dz_es, const unsigned short* __restrict__ wd2b, float* __restrict__ amax, float* __restrict__ out,
    int64_t out_es, int n) {
  using K = Dgrad2Both;
  extern __shared__ __attribute__((aligned(16))) char ldsd2[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int e = blockIdx.y, b0 = blockIdx.x * K::SPT;
  const float sa = plane_scale(amax[amax_idx(AMAX_DZ2, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_W2, e)]));
  for (int i = tid; i < K::W_OFF / 16; i += K::THREADS) *(f4*)(ldsd2 + i * 16) = zero4();  // images incl. their zero borders
  // ---- staging maps.  unit u = tid: sample u / 81, pixel u % 81 -> 8 loads of stride 81 (the k-block's 8 oc)
  const int u = min(tid, K::NIU - 1);
  const float* isrc = dz2 + e * dz_es + (int64_t)min(b0 + u / 81, n - 1) * 5184 + u % 81;  // + (8 kb + c) * 81
  const int idst = ((u / 81) * 121 + ((u % 81) / 9 + 1) * 11 + (u % 81) % 9 + 1) * 16;
  // weight quad q = tid + 512 j of the two parities' k-blocks (W_HALF bytes each, one after the other in LDS)
  const unsigned short* wsrc[K::NWJ];
#pragma unroll
  for (int j = 0; j < K::NWJ; ++j) {
    const int q = tid + K::THREADS * j, par = q / (K::W_HALF / 16), qh = q % (K::W_HALF / 16);
    wsrc[j] = wd2b + (int64_t)(e * 2 + par) * (8 * 2 * NPL * 64 * 16) + qh * 8;  // + kb * 2 * NPL * 1024
  }
  // ---- operand bases: row tile i = 2 a + c
  int aA[4], bB[K::TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) aA[i] = K::W_OFF + (i >> 1) * K::W_HALF + ((i & 1) * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    int c = wc * (32 * K::TN) + j * 32 + l31;
    if (c >= K::SPT * 100) c = 0;
    const int s = c / 100, pq = c % 100;
    bB[j] = (s * 121 + (pq / 10 + 1) * 11 + (pq % 10 + 1) - hi) * 16;  // pixel (p, q - v) of the padded image, v = lane half
  }
  float ir[8];
  f4 wr[K::NWJ];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int c = 0; c < 8; ++c) ir[c] = isrc[(kb * 8 + c) * 81];
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j) wr[j] = *(const f4*)(wsrc[j] + kb * (2 * NPL * 1024));
  };
  auto commit = [&]() {
    if (tid < K::NIU) {
      unsigned pl[4][NPL];
#pragma unroll
      for (int c = 0; c < 4; ++c) split_planes(ir[2 * c], ir[2 * c + 1], sa, pl[c]);
      char* d = ldsd2 + idst;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u4v*)(d + p * K::IMG_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
    }
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j) *(f4*)(ldsd2 + K::W_OFF + (tid + K::THREADS * j) * 16) = wr[j];
  };
  f32x16 acc[4][K::TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  constexpr int NKB = 8;
  fetch(0);
  __syncthreads();  // zero fill complete
  commit();
  fetch(1);
  __syncthreads();
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {  // kg = u
      frag8 af[NPL][4], bfr[NPL][K::TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[p][i] = *(const frag8*)(ldsd2 + aA[i] + (kg * NPL + p) * 2048);
#pragma unroll
        for (int j = 0; j < K::TN; ++j) bfr[p][j] = *(const frag8*)(ldsd2 + bB[j] + p * K::IMG_PLANE - kg * (11 * 16));
      }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < K::TN; ++j) acc[i][j] = mfma_planes(af[PA[t]][i], bfr[PB[t]][j], acc[i][j]);
    }
    __syncthreads();  // every wave is done with the stage
    if (kb + 1 < NKB) {
      commit();
      if (kb + 2 < NKB) fetch(kb + 2);
    }
    __syncthreads();
  }
  // row tiles (2 a, 2 a + 1) = the two column parities of row parity a: horizontally adjacent pixels -> one 8-byte store
  float big = 0.0f;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    const int c = wc * (32 * K::TN) + j * 32 + l31;
    const int s = c / 100, pq = c % 100;
    if (c >= K::SPT * 100 || b0 + s >= n) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float* base = out + e * out_es + (int64_t)(b0 + s) * 12800 + (2 * (pq / 10) + a) * 20 + 2 * (pq % 10);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float g0 = acc[2 * a][j][r] * inv, g1 = acc[2 * a + 1][j][r] * inv;
        *(float2*)(base + acc_row(r, hi) * 400) = make_float2(g0, g1);
        big = fmaxf(big, fmaxf(fabsf(g0), fabsf(g1)));
      }
    }
  }
  amax_update(big, amax + amax_idx(AMAX_DZ1, e));  // da1 before the leaky mask: an upper bound of what conv1's weight gradient stages
}

void launch_conv_dgrad2_2(const EncCall& c, hipStream_t st) {
  const Workspace& w = *c.ws;
  ProfRange pr(c.prof, "ConvDgrad2", st);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_dgrad2_both_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Dgrad2Both::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_dgrad2_both_kernel, dim3((unsigned)((c.n + Dgrad2Both::SPT - 1) / Dgrad2Both::SPT), (unsigned)c.L->NE, 1),
                     dim3(Dgrad2Both::THREADS), Dgrad2Both::LDS_BYTES, st, w.dz2, c.max_batch * 5184, w.wd2b, w.amax, w.dz1, c.max_batch * 12800, c.n);
}

}  // namespace ddrl
