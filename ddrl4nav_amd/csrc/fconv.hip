// First layer of the nav encoders -- a handful of input channels under a wide kernel -- on the 16-bit matrix pipe, fp32-accurate
// (round 4; engine2.h "f16x3": two scaled fp16 planes per operand, three products per k-group on v_mfma_f32_32x32x16_f16):
//   NavPreNet1D   conv1  3 -> 64  7x7 @48, padding 1      (reference USTC_lab/nn/nav_encoder.py:96)
//   NavPreNet     conv1  1 -> 64  3x3 @48, padding 1      (nav_encoder.py:18; image_batch = 1)
//   NavPedPreNet  conv1  4 -> 64  3x3 @48, padding 1      (nav_encoder.py:56; image + three pedestrian maps)
// replacing dconv.hip's f32-input band kernels (90 TFLOP/s) / the gather kernels for the forward and the weight gradient (a first layer
// has no data gradient).  The text below describes the 7x7 layer; the 3x3 layers use the same template with one kx half (3 k-groups).
//
// With three channels a k-group cannot be "one tap x 16 channels" (pconv.hip).  Both kernels keep the input as a zero-bordered image with
// the channels innermost and padded to four, [row][column][4 channels] fp16 = 8 bytes per pixel and plane, and order the reduction index as
//   k = (ky, kx, c),  kx in 8 slots (slot 7 carries zero weights), c in 4 slots (slot 3 likewise):  K = 7 x 8 x 4 = 224 = 14 k-groups
// so that 8 consecutive k = two neighbouring pixels x 4 channels = 16 contiguous bytes of the image, 8-byte aligned.
//
// Forward:  out[b][oc][p] = sum_k Wk[oc][k] img[p + tap(k)]: rows = 64 oc (A operand: weight planes, resident in LDS for the whole launch),
// columns = output pixels (B operand: straight from the image, lane = pixel, address = pixel + compile-time tap offset).  One workgroup of
// eight waves walks whole samples: image double-buffered (the next sample's pixels are in flight / staged while this one is multiplied),
// 2 passes x 8 waves x 4 column tiles = 64 tiles for the 60.5 of a 44 x 44 output, NO barrier inside a sample.  The sample's
// power-of-two scale comes from its own largest magnitude, found while it is staged (no pre-pass).
//
// Weight gradient:  dW[oc][k] = sum_{b, p} dz[b][oc][p] img[b][p + tap(k)]: rows = 64 oc (A operand: dz, 8 consecutive pixels of an
// output row), columns = the 32 (kx, c) slots of one ky (B operand), reduction = runs of 16 pixels along an output row.  The B
// fragments come from the same channel-innermost image through the transposing LDS read (ds_read_b64_tr_b16: lane (q, pp) of a
// 16-lane group reads the 8-byte pixel at x + q + pp, the group returns 4 pixels x (4 kx x 4 c) transposed -- every address 8-byte
// aligned).  A stage = a band of 4 output rows of one sample (12 k-groups); workgroups split the (sample, band) list and leave slabs.
// Scales: each stage's largest magnitudes are found while it is staged; a workgroup multiplies under the SMALLEST scale it has met so
// far (accumulators are re-scaled by the power-of-two ratio when it drops), i.e. never coarser than one scale for the whole batch --
// without a pre-pass over the 2 GB of dz.
#include <cstdlib>

#include "engine2.h"
#include "ops.h"

#ifdef DDRL_PLANES_BF16
// The three-plane build (bf16x6) keeps the f32-input kernels for this layer (LDS budgets).  Every query answers "no", nothing launches.
namespace ddrl {
bool conv_has_first(const ConvGeom&) { return false; }
int64_t conv_first_pack_floats(const ConvGeom&) { return 0; }
int conv_first_wgrad_splits(const ConvGeom&) { return 0; }
void launch_conv_first_pack(const ConvGeom&, const float*, float*, hipStream_t) {}
void launch_conv_first_fwd(const ConvGeom&, const float*, const float*, const float*, int, float*, hipStream_t) {}
void launch_conv_first_fwd_pool(const ConvGeom&, const float*, const float*, const float*, float*, uint8_t*, float*, hipStream_t) {}
void launch_conv_first_wgrad(const ConvGeom&, const float*, const float*, float*, float*, float*, hipStream_t) {}
void launch_conv_first_wgrad_pooled(const ConvGeom&, const float*, const float*, const uint8_t*, float*, float*, float*, hipStream_t) {}
}  // namespace ddrl
#else
namespace ddrl {

namespace fconv {

using u2v = __attribute__((ext_vector_type(2))) unsigned;
using u4v = __attribute__((ext_vector_type(4))) unsigned;
using s4w = __attribute__((ext_vector_type(4))) short;

template <int CIN_, int KS_, int HIN_, int PAD_>
struct FGeo {
  static constexpr int CIN = CIN_, COUT = 64, KS = KS_, HIN = HIN_, PAD = PAD_, KK = KS * KS;
  static constexpr int OH = HIN + 2 * PAD - KS + 1, P = OH * OH, RAW = HIN * HIN, LPY = HIN + 2 * PAD;
  static constexpr int KXH = (KS + 3) / 4;                   // halves of the 8 kx slots that carry weights (7 taps: 2, 3 taps: 1)
  static constexpr int NKG = KXH * KS;                       // k-groups: (ky, kx half)
  static constexpr int W_BYTES = NKG * NPL * 64 * 32;        // [k-group][plane][oc 64][lane half 2][8 k] fp16
  // ---- forward
  static constexpr int FW_WAVES = 8, FW_TN = 4, FW_TILES = (P + 31) / 32, FW_PASSES = (FW_TILES + FW_WAVES * FW_TN - 1) / (FW_WAVES * FW_TN);
  // pooled epilogue: the upper rows of the row pairs as one stream of OH / 2 x OH pixels, two stream tiles (x 2 rows) per wave and pass
  static constexpr int OW2 = OH / 2, PW = OW2 * OW2, STREAM = OW2 * OH;
  static constexpr int FWP_PASSES = ((STREAM + 31) / 32 + FW_WAVES * (FW_TN / 2) - 1) / (FW_WAVES * (FW_TN / 2));
  static constexpr int LPX_F = OH + 8;                       // columns a fragment may touch: ox + 7 + 1
  static constexpr int IMG_F_PLANE = LPY * LPX_F * 8, IMG_F = NPL * IMG_F_PLANE;
  static constexpr int F_IMG_OFF = W_BYTES, F_BIAS_OFF = F_IMG_OFF + 2 * IMG_F, F_RED_OFF = F_BIAS_OFF + 64 * 4;
  static constexpr int F_OMAX_OFF = F_RED_OFF + 64, F_LDS = F_OMAX_OFF + 32;   // + the eight waves' largest pooled value of a sample
  static constexpr int F_UNITS = HIN * (HIN / 4), F_NJ = (F_UNITS + 511) / 512;   // (row, 4 pixels) staging units
  // ---- weight gradient
  static constexpr int BR = 4, NBANDS = OH / BR, RUNS = (OH + 15) / 16, PXR = RUNS * 16, BRPX = BR * PXR;
  static constexpr int WROWS = BR + KS - 1, LPX_W = PXR + 8;
  static constexpr int DZP = BRPX + 8;                       // dz row pitch (16-bit elements): 400 bytes, consecutive oc rows 16 bytes apart in the banks
  static constexpr int DZ_PLANE = 64 * DZP * 2, IMG_W_PLANE = WROWS * LPX_W * 8;
  static constexpr int W_IMG_OFF = NPL * DZ_PLANE, W_RED_OFF = W_IMG_OFF + NPL * IMG_W_PLANE, W_LDS = W_RED_OFF + 64;
  static constexpr int DZ_UNITS = 64 * BR * (OH / 4), DZ_NJ = DZ_UNITS / 256, IM_UNITS = WROWS * (HIN / 4);
  static constexpr int64_t SLAB = (int64_t)64 * CIN * KK + 64;
  static_assert(CIN <= 4 && KS <= 7 && (KS & 1) && HIN % 4 == 0 && OH % 4 == 0 && OH % BR == 0, "geometry");
  static_assert(DZ_UNITS % 256 == 0 && IM_UNITS <= 256 && DZ_UNITS * 4 <= W_IMG_OFF, "staging maps of the weight gradient");
  static_assert(F_LDS <= 160 * 1024 && 2 * (W_LDS + 1024) <= 160 * 1024, "LDS budget (two weight-gradient workgroups per CU)");
};

__global__ __launch_bounds__(256) void weight_amax_kernel(const float* __restrict__ w, int64_t count, float* __restrict__ slot) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  amax_update(m, slot);
}

// dst[k-group = KXH ky + kx / 4][plane][oc 64][lane half][8 k]: k = 8 half + 4 t + c  <->  kx = 4 (k-group % KXH) + 2 half + t; hdr[0] =
// largest |w| (in), hdr[1] = scale (out)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int cin, int ks, unsigned short* __restrict__ dst,
                                                   float* __restrict__ hdr) {
  const int kxh = (ks + 3) / 4, total = kxh * ks * 1024;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float scale = plane_scale(hdr[0]);
  if (i == 0) hdr[1] = scale;
  if (i >= total) return;
  const int e = i & 7, hf = (i >> 3) & 1, oc = (i >> 4) & 63, kg = i >> 10;
  const int ky = kg / kxh, kx = 4 * (kg % kxh) + 2 * hf + (e >> 2), c = e & 3;
  const float v = (kx < ks && c < cin) ? w[((oc * cin + c) * ks + ky) * ks + kx] : 0.0f;
  unsigned short pl[NPL];
  planes_of(v, scale, pl);
#pragma unroll
  for (int p = 0; p < NPL; ++p) dst[(kg * NPL + p) * 1024 + oc * 16 + hf * 8 + e] = pl[p];
}

// the power-of-two plane scale of a block from the largest magnitudes its waves left in red[0..nw): zero blocks get the largest scale
__device__ __forceinline__ float scale_from(const float* red, int nw) {
  float m = 0.0f;
  for (int i = 0; i < nw; ++i) m = fmaxf(m, red[i]);
  return m > 0.0f ? fminf(plane_scale(m), 0x1p60f) : 0x1p60f;
}
__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  return m;
}
__device__ __forceinline__ float amax4(f4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// four pixels (x .. x + 3) of CIN channel planes -> [pixel][4 channels] fp16 planes, 8 bytes per pixel and plane
template <int CIN>
__device__ __forceinline__ void commit_quad(char* img, int plane_bytes, const f4 (&v)[CIN], float scale) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    unsigned p0[NPL], p1[NPL];
    split_planes(v[0][t], CIN > 1 ? v[CIN > 1 ? 1 : 0][t] : 0.0f, scale, p0);
    if (CIN > 2) {
      split_planes(v[CIN > 2 ? 2 : 0][t], CIN > 3 ? v[CIN > 3 ? 3 : 0][t] : 0.0f, scale, p1);
    } else {
      p1[0] = p1[1] = 0u;
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) *(u2v*)(img + p * plane_bytes + t * 8) = (u2v){p0[p], p1[p]};
  }
}

// POOL: ReLU + max_pool2d(2) in the epilogue.  The column tiles of a wave then come in PAIRS over the same 32 consecutive pixels of
// the upper and the lower row of a row pair (the pixels of all upper rows form one stream of OH / 2 x OH elements; OH is even, so
// lanes 2 m and 2 m + 1 always hold the two columns of one pooling window): a window's four activations are two accumulators of a
// lane and the same two of its neighbour (one DPP move each), and neighbouring lanes still read neighbouring pixels (lane = window
// would double the stride of the fragment reads: two-way bank conflicts, measured).  `out` receives the pooled map
// [oc][OH / 2][OH / 2], `code` one decision byte per window (gconv.hip maxpool2_fwd_idx_kernel: first maximum in PyTorch's scan
// order + sign); the full-resolution activations are never written.
template <class K, bool POOL>
__global__ __launch_bounds__(512) void first_fwd_kernel(const float* __restrict__ in, int64_t in_sn, const unsigned short* __restrict__ wp,
                                                        const float* __restrict__ whdr, const float* __restrict__ bias, int act,
                                                        float* __restrict__ out, int64_t out_sn, uint8_t* __restrict__ code,
                                                        float* __restrict__ out_amax, int n) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_u(), l31 = lane & 31, hi = lane >> 5;
  float* lbias = (float*)(lds + K::F_BIAS_OFF);
  float* red = (float*)(lds + K::F_RED_OFF);  // [2][8]
  float* omax = (float*)(lds + K::F_OMAX_OFF);  // [8]: POOL with out_amax -- the sample's largest pooled value (= largest magnitude: >= 0)
  for (int i = tid; i < 2 * K::IMG_F / 16; i += 512) *(f4*)(lds + K::F_IMG_OFF + i * 16) = zero4();  // borders and the fourth channel stay zero
  for (int i = tid; i < K::W_BYTES / 16; i += 512) *(f4*)(lds + i * 16) = *(const f4*)((const char*)wp + i * 16);
  if (tid < 64) lbias[tid] = bias[tid];
  // ---- staging map: unit u = tid + 512 j -> input row u / (HIN / 4), pixels 4 (u % (HIN / 4)) ..
  int soff[K::F_NJ], sdst[K::F_NJ];
#pragma unroll
  for (int j = 0; j < K::F_NJ; ++j) {
    const int u = min(tid + 512 * j, K::F_UNITS - 1), y = u / (K::HIN / 4), x4 = u % (K::HIN / 4);
    soff[j] = y * K::HIN + x4 * 4;
    sdst[j] = ((y + K::PAD) * K::LPX_F + x4 * 4 + K::PAD) * 8;
  }
  f4 sr[K::F_NJ][K::CIN];
  auto fetch = [&](int b) {
    const float* src = in + (int64_t)b * in_sn;
#pragma unroll
    for (int j = 0; j < K::F_NJ; ++j)
#pragma unroll
      for (int c = 0; c < K::CIN; ++c) sr[j][c] = ld4(src + c * K::RAW + soff[j]);
  };
  auto leave_amax = [&](float* slot) {
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < K::F_NJ; ++j)
      if (j + 1 < K::F_NJ || tid + 512 * j < K::F_UNITS) {
#pragma unroll
        for (int c = 0; c < K::CIN; ++c) m = fmaxf(m, amax4(sr[j][c]));
      }
    m = wave_max(m);
    if (lane == 0) slot[wave] = m;
  };
  auto commit = [&](char* img, float scale) {
#pragma unroll
    for (int j = 0; j < K::F_NJ; ++j)
      if (j + 1 < K::F_NJ || tid + 512 * j < K::F_UNITS) commit_quad<K::CIN>(img + sdst[j], K::IMG_F_PLANE, sr[j], scale);
  };
  // ---- fragment bases: weights (rows = oc), pixels of this wave's column tiles per pass
  int aA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (i * 32 + l31) * 32 + hi * 16;
  const float winv = 1.0f / whdr[1];

  int b = blockIdx.x;
  if (b < n) {
    fetch(b);
    leave_amax(red);
  }
  __syncthreads();  // zero fill, weights, bias, the first sample's maxima
  float sc_cur = scale_from(red, 8);
  if (b < n) commit(lds + K::F_IMG_OFF, sc_cur);
  int bn = b + gridDim.x;
  if (bn < n) fetch(bn);
  __syncthreads();
  for (int it = 0; b < n; ++it) {
    const char* img = lds + K::F_IMG_OFF + (it & 1) * K::IMG_F;
    const float inv = winv / sc_cur;
    float* obase = out + (int64_t)b * out_sn;
    float omx = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < (POOL ? K::FWP_PASSES : K::FW_PASSES); ++pass) {
      int bB[K::FW_TN], pix[K::FW_TN];
#pragma unroll
      for (int j = 0; j < K::FW_TN; ++j) {
        if (POOL) {  // tiles (2 jj, 2 jj + 1) = stream elements q .. of the upper / lower rows
          const int q = ((pass * K::FW_WAVES + wave) * (K::FW_TN / 2) + (j >> 1)) * 32 + l31;
          pix[j] = q < K::STREAM ? (2 * (q / K::OH) + (j & 1)) * K::OH + q % K::OH : K::P;
        } else {
          pix[j] = ((pass * K::FW_WAVES + wave) * K::FW_TN + j) * 32 + l31;
        }
        const int p = pix[j] < K::P ? pix[j] : 0;
        bB[j] = ((p / K::OH) * K::LPX_F + p % K::OH + 2 * hi) * 8;
      }
      f32x16 acc[2][K::FW_TN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < K::FW_TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
      // one kernel row (two k-groups) per trip: the loop bounds how far ahead the scheduler pulls fragment reads (fully unrolled, the
      // fourteen k-groups spilled 72 - 288 registers); the row offsets are wave-uniform adds, the kx half stays an immediate
#pragma unroll 1
      for (int ky = 0; ky < K::KS; ++ky) {
        const char* wrow = lds + ky * (K::KXH * NPL * 2048);
        const char* irow = img + ky * (K::LPX_F * 8);
#pragma unroll
        for (int h4 = 0; h4 < K::KXH; ++h4) {
          frag8 af[NPL][2], bf[NPL][K::FW_TN];
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[p][i] = *(const frag8*)(wrow + aA[i] + (h4 * NPL + p) * 2048);
#pragma unroll
            for (int j = 0; j < K::FW_TN; ++j) {
              const char* s = irow + p * K::IMG_F_PLANE + bB[j] + h4 * 32;  // 8-byte aligned: two 8-byte reads
              const u2v lo = *(const u2v*)s, hh = *(const u2v*)(s + 8);
              bf[p][j] = __builtin_bit_cast(frag8, (u4v){lo.x, lo.y, hh.x, hh.y});
            }
          }
          DDRL_PLANE_PRODUCTS;
#pragma unroll
          for (int m = 0; m < NPROD; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < K::FW_TN; ++j) acc[i][j] = mfma_planes(af[PA[m]][i], bf[PB[m]][j], acc[i][j]);
        }
      }
      if (POOL) {
        uint8_t* cbase = code + (int64_t)b * (64 * K::PW);
#pragma unroll
        for (int jj = 0; jj < K::FW_TN / 2; ++jj) {
          const int top = pix[2 * jj];                                             // this lane's pixel of the upper row
          const int win = (top / K::OH / 2) * K::OW2 + (top % K::OH) / 2;
          const bool writer = top < K::P && !(lane & 1);
          // every lane takes part in the exchange; the even lanes then store under ONE predicate (a branch per store costs more than
          // the stores: measured)
          float pm[2][16];
          int pc[2][16];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int oc = i * 32 + acc_row(r, hi);
              const float v0 = fmaxf(__builtin_fmaf(acc[i][2 * jj][r], inv, lbias[oc]), 0.0f);
              const float v2 = fmaxf(__builtin_fmaf(acc[i][2 * jj + 1][r], inv, lbias[oc]), 0.0f);
              const float v1 = lane_swap1(v0), v3 = lane_swap1(v2);               // the window's right column, from the odd lane
              float m = v0;
              int am = 0;
              if (v1 > m) { m = v1; am = 1; }
              if (v2 > m) { m = v2; am = 2; }
              if (v3 > m) { m = v3; am = 3; }
              pm[i][r] = m;
              pc[i][r] = am | (m > 0.0f ? 4 : 0);
            }
          if (writer) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) omx = fmaxf(omx, pm[i][r]);
          }
          if (writer) {
            float* op = obase + 4 * hi * K::PW + win;
            uint8_t* cp = cbase + 4 * hi * K::PW + win;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int ro = (i * 32 + (r & 3) + 8 * (r >> 2)) * K::PW;
                op[ro] = pm[i][r];
                cp[ro] = (uint8_t)pc[i][r];
              }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < K::FW_TN; ++j) {
          if (pix[j] >= K::P) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int oc = i * 32 + acc_row(r, hi);
              float v = __builtin_fmaf(acc[i][j][r], inv, lbias[oc]);
              if (act == 1) v = fmaxf(v, 0.0f);
              obase[(int64_t)oc * K::P + pix[j]] = v;
            }
        }
      }
    }
    // ---- the next sample: its maxima, then its planes into the other image (released by the barrier that ended the previous round)
    const bool more = bn < n;
    if (more) leave_amax(red + 8 * ((it + 1) & 1));
    if (POOL && out_amax != nullptr) {   // this workgroup owns the whole sample: its eight waves' maxima meet behind the barrier below
      omx = wave_max(omx);
      if (lane == 0) omax[wave] = omx;
    }
    __syncthreads();
    if (POOL && out_amax != nullptr && tid == 0)
      amax_raise(fmaxf(fmaxf(fmaxf(omax[0], omax[1]), fmaxf(omax[2], omax[3])), fmaxf(fmaxf(omax[4], omax[5]), fmaxf(omax[6], omax[7]))), out_amax + b);
    if (more) {
      sc_cur = scale_from(red + 8 * ((it + 1) & 1), 8);
      commit(lds + K::F_IMG_OFF + ((it + 1) & 1) * K::IMG_F, sc_cur);
    }
    b = bn;
    bn += gridDim.x;
    if (bn < n) fetch(bn);
    __syncthreads();
  }
}

__device__ __forceinline__ frag8 tr_frag(const char* lds, int off_lo, int off_hi) {
  typedef s4w __attribute__((address_space(3))) * lds_s4;
  const s4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_lo));
  const s4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s8w;
  return __builtin_bit_cast(frag8, (s8w)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// UNPOOL: `dz` is d(pooled) [64][OH / 2][OH / 2] of the layer's ReLU + max_pool2d(2) and `ucode` its decision bytes; the staging
// forms d(pre-activation) on the fly (the pooled gradient at the window's first maximum under the ReLU's sign, zero elsewhere), so
// the 2 GB full-resolution gradient of a 4,096-sample batch is neither written nor read.
template <class K, bool UNPOOL>
__global__ __launch_bounds__(256, 2) void first_wgrad_kernel(const float* __restrict__ in, int64_t in_sn, const float* __restrict__ dz,
                                                             int64_t dz_sn, const uint8_t* __restrict__ ucode, float* __restrict__ part, int n,
                                                             int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_u(), l31 = lane & 31, hi = lane >> 5;
  float* red = (float*)(lds + K::W_RED_OFF);  // [0..3] dz maxima of the waves, [4..7] image maxima
  const int split = blockIdx.x;
  const int nst = n * K::NBANDS, per = (nst + nsplit - 1) / nsplit;
  const int st_begin = min(nst, split * per), st_end = min(nst, st_begin + per);
  for (int i = tid; i < K::W_RED_OFF / 16; i += 256) *(f4*)(lds + i * 16) = zero4();  // the padded pixels of the dz rows and the image borders stay zero
  // ---- staging maps
  int doff[K::DZ_NJ], ddst[K::DZ_NJ], dpos[K::DZ_NJ];
#pragma unroll
  for (int j = 0; j < K::DZ_NJ; ++j) {
    const int u = tid + 256 * j, oc = u / (K::BR * (K::OH / 4)), rem = u % (K::BR * (K::OH / 4)), r = rem / (K::OH / 4), x4 = rem % (K::OH / 4);
    if (UNPOOL) {  // two windows per quad: row pair r / 2 of the band, columns 2 x4, 2 x4 + 1
      doff[j] = oc * K::PW + (r >> 1) * K::OW2 + x4 * 2;   // + (oy0 / 2) * OW2
      dpos[j] = (r & 1) * 2;
    } else {
      doff[j] = oc * K::P + r * K::OH + x4 * 4;            // + oy0 * OH
    }
    ddst[j] = (oc * K::DZP + r * K::PXR + x4 * 4) * 2;
  }
  const int iu = min(tid, K::IM_UNITS - 1), irow = iu / (K::HIN / 4), ix4 = iu % (K::HIN / 4);
  const int idst = K::W_IMG_OFF + (irow * K::LPX_W + K::PAD + ix4 * 4) * 8;
  f4 dr[K::DZ_NJ], ir[K::CIN];
  float bsum[K::DZ_NJ];
#pragma unroll
  for (int j = 0; j < K::DZ_NJ; ++j) bsum[j] = 0.0f;
  bool irow_ok = false;
  auto fetch = [&](int st) {
    const int b = st / K::NBANDS, oy0 = (st % K::NBANDS) * K::BR;
    if (UNPOOL) {
      const int64_t o = (int64_t)b * dz_sn + (oy0 / 2) * K::OW2;
#pragma unroll
      for (int j = 0; j < K::DZ_NJ; ++j) {
        const f2 g = *(const f2*)(dz + o + doff[j]);                  // (OW2 and 2 x4 are even: 8-byte aligned)
        const unsigned cc = *(const unsigned short*)(ucode + o + doff[j]);
        dr[j] = (f4){g.x, g.y, __uint_as_float(cc), 0.0f};            // routed at commit, once the loads have landed
      }
    } else {
      const float* dsrc = dz + (int64_t)b * dz_sn + oy0 * K::OH;
#pragma unroll
      for (int j = 0; j < K::DZ_NJ; ++j) dr[j] = ld4(dsrc + doff[j]);
    }
    const int iy = oy0 - K::PAD + irow;
    irow_ok = iy >= 0 && iy < K::HIN;
    const float* isrc = in + (int64_t)b * in_sn + min(max(iy, 0), K::HIN - 1) * K::HIN + ix4 * 4;
#pragma unroll
    for (int c = 0; c < K::CIN; ++c) ir[c] = ld4(isrc + c * K::RAW);
  };
  auto route = [&]() {  // UNPOOL: (g0, g1, codes) -> the four gradients of the quad
    if (UNPOOL) {
#pragma unroll
      for (int j = 0; j < K::DZ_NJ; ++j) {
        const unsigned cc = __float_as_uint(dr[j].z), c0 = cc & 7u, c1 = (cc >> 8) & 7u;
        const unsigned p0 = 4u | (unsigned)dpos[j];
        const float g0 = dr[j].x, g1 = dr[j].y;
        dr[j] = (f4){c0 == p0 ? g0 : 0.0f, c0 == p0 + 1 ? g0 : 0.0f, c1 == p0 ? g1 : 0.0f, c1 == p0 + 1 ? g1 : 0.0f};
      }
    }
  };
  auto leave_amax = [&]() {
    float md = 0.0f, mi = 0.0f;
    route();
#pragma unroll
    for (int j = 0; j < K::DZ_NJ; ++j) md = fmaxf(md, amax4(dr[j]));
    if (irow_ok && tid < K::IM_UNITS) {
#pragma unroll
      for (int c = 0; c < K::CIN; ++c) mi = fmaxf(mi, amax4(ir[c]));
    }
    md = wave_max(md);
    mi = wave_max(mi);
    if (lane == 0) {
      red[wave] = md;
      red[4 + wave] = mi;
    }
  };
  auto commit = [&](float sd, float sa) {
#pragma unroll
    for (int j = 0; j < K::DZ_NJ; ++j) {
      unsigned p0[NPL], p1[NPL];
      split_planes(dr[j].x, dr[j].y, sd, p0);
      split_planes(dr[j].z, dr[j].w, sd, p1);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u2v*)(lds + p * K::DZ_PLANE + ddst[j]) = (u2v){p0[p], p1[p]};
      bsum[j] += (dr[j].x + dr[j].y) + (dr[j].z + dr[j].w);
    }
    if (tid < K::IM_UNITS) {
      f4 v[K::CIN];
#pragma unroll
      for (int c = 0; c < K::CIN; ++c) v[c] = irow_ok ? ir[c] : zero4();  // rows above / below the image
      commit_quad<K::CIN>(lds + idst, K::IMG_W_PLANE, v, sa);
    }
  };
  // ---- fragment addresses.  dz: lane = (oc, pixel half).  Image: 16-lane group g16 -> kx half (g16 & 1), pixel half (g16 >> 1);
  // inside the group lane 4 q + pp reads the pixel at x + q + pp (first read) / + 4 (second): rows q = pixels, chunks pp = kx
  int aA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = ((i * 32 + l31) * K::DZP + 8 * hi) * 2;
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  constexpr int KYW = (K::KS + 3) / 4;  // kernel rows per wave (7 taps: 2, the last wave has one; 3 taps: 1, the last wave idles)
  const int ky0 = KYW * wave;
  const bool two = KYW == 2 && ky0 + 1 < K::KS;
  const int bL = K::W_IMG_OFF + (ky0 * K::LPX_W + 8 * (g16 >> 1) + q + 4 * (g16 & 1) + pp) * 8;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][s][r] = 0.0f;
  float sd_run = 0x1p60f, sa_run = 0x1p60f;
  if (st_begin < st_end) fetch(st_begin);
  for (int st = st_begin; st < st_end; ++st) {
    leave_amax();
    __syncthreads();  // the previous stage's fragments are read, this stage's maxima are in place
    const float sd = fminf(sd_run, scale_from(red, 4)), sa = fminf(sa_run, scale_from(red + 4, 4));
    if (sd * sa != sd_run * sa_run) {  // wave-uniform: a larger magnitude arrived, the sums so far move to the coarser scale (power of two: exact)
      const float ratio = (sd * sa) / (sd_run * sa_run);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][s][r] *= ratio;
      sd_run = sd;
      sa_run = sa;
    }
    commit(sd, sa);
    if (st + 1 < st_end) fetch(st + 1);
    __syncthreads();
    if (ky0 < K::KS) {
#pragma unroll
      for (int r = 0; r < K::BR; ++r)
#pragma unroll
        for (int run = 0; run < K::RUNS; ++run) {
          frag8 af[NPL][2], bf[NPL][2];
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[p][i] = *(const frag8*)(lds + p * K::DZ_PLANE + aA[i] + (r * K::PXR + run * 16) * 2);
            const int o = bL + p * K::IMG_W_PLANE + (r * K::LPX_W + run * 16) * 8;
            bf[p][0] = tr_frag(lds, o, o + 32);
            if (two) bf[p][1] = tr_frag(lds, o + K::LPX_W * 8, o + K::LPX_W * 8 + 32);
          }
          DDRL_PLANE_PRODUCTS;
#pragma unroll
          for (int m = 0; m < NPROD; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[i][0] = mfma_planes(af[PA[m]][i], bf[PB[m]][0], acc[i][0]);
              if (two) acc[i][1] = mfma_planes(af[PA[m]][i], bf[PB[m]][1], acc[i][1]);
            }
        }
    }
  }
  // ---- epilogue: slab[oc][c][ky][kx] (torch layout), then the bias partial
  float* slab = part + (int64_t)split * K::SLAB;
  const float inv = 1.0f / (sd_run * sa_run);
  const int kx = l31 >> 2, c = l31 & 3;
  if (ky0 < K::KS && kx < K::KS && c < K::CIN) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && !two) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[((i * 32 + acc_row(r, hi)) * K::CIN + c) * K::KK + (ky0 + s) * K::KS + kx] = acc[i][s][r] * inv;
    }
  }
  __syncthreads();
  float* bred = (float*)lds;  // [unit]: unit u = tid + 256 j belongs to oc = u / (BR * OH / 4)
#pragma unroll
  for (int j = 0; j < K::DZ_NJ; ++j) bred[tid + 256 * j] = bsum[j];
  __syncthreads();
  if (tid < 64) {
    constexpr int UPO = K::BR * (K::OH / 4);
    float s = 0.0f;
    for (int u = 0; u < UPO; ++u) s += bred[tid * UPO + u];
    slab[(int64_t)64 * K::CIN * K::KK + tid] = s;
  }
}

}  // namespace fconv

// ---- host side ---------------------------------------------------------------------------------------------------------------------
using FN1dC1 = fconv::FGeo<3, 7, 48, 1>;   // NavPreNet1D.conv1
using FNavC1 = fconv::FGeo<1, 3, 48, 1>;   // NavPreNet.conv1 (image_batch = 1): 3 k-groups, 9 of their 48 k carry weights
using FPedC1 = fconv::FGeo<4, 3, 48, 1>;   // NavPedPreNet.conv1 (image + 3 pedestrian maps): 36 of 48

enum FirstId { kFNone = 0, kFN1d, kFNav, kFPed };

static FirstId first_id(const ConvGeom& g) {
#ifdef DDRL_PLANES_BF16
  return kFNone;  // diagnostic three-plane build: first layers on the generic gather kernels (gconv.hip) as well
#else
  if (g.stride != 1 || g.h != 48 || g.w != 48 || g.kh != g.kw || g.pad_h != 1 || g.pad_w != 1 || g.cout != 64) return kFNone;
  if (g.kh == 7 && g.cin == 3) return kFN1d;
  if (g.kh == 3 && g.cin == 1) return kFNav;
  if (g.kh == 3 && g.cin == 4) return kFPed;
  return kFNone;
#endif
}

bool conv_has_first(const ConvGeom& g) { return first_id(g) != kFNone; }

// dispatch over the compile-time geometries
#define DDRL_FIRST_DISPATCH(g, CALL)        \
  switch (first_id(g)) {                    \
    case kFN1d: { using K = FN1dC1; CALL; } break; \
    case kFNav: { using K = FNavC1; CALL; } break; \
    case kFPed: { using K = FPedC1; CALL; } break; \
    default: break;                         \
  }

// floats of the packed region: the forward's weight planes + a 64-float header
int64_t conv_first_pack_floats(const ConvGeom& g) {
  int64_t f = 0;
  DDRL_FIRST_DISPATCH(g, f = K::W_BYTES / 4 + 64);
  return f;
}

void launch_conv_first_pack(const ConvGeom& g, const float* w, float* region, hipStream_t st) {
  float* hdr = region + conv_first_pack_floats(g) - 64;
  (void)hipMemsetAsync(hdr, 0, 64 * sizeof(float), st);
  hipLaunchKernelGGL(fconv::weight_amax_kernel, dim3(8), dim3(256), 0, st, w, (int64_t)g.cout * g.cin * g.kh * g.kw, hdr);
  hipLaunchKernelGGL(fconv::pack_kernel, dim3((((g.kh + 3) / 4) * g.kh * 1024 + 255) / 256), dim3(256), 0, st, w, g.cin, g.kh, (unsigned short*)region, hdr);
}

template <class K, bool POOL>
static void run_first_fwd(const ConvGeom& g, const float* in, const float* region, const float* bias, int act, float* out, int64_t out_sn, uint8_t* code,
                          float* out_amax, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)fconv::first_fwd_kernel<K, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, K::F_LDS);
    configured = true;
  }
  const int grid = g.n < 256 ? g.n : 256;  // one persistent workgroup per CU walks the samples
  hipLaunchKernelGGL((fconv::first_fwd_kernel<K, POOL>), dim3(grid), dim3(512), K::F_LDS, st, in, g.in_sn, (const unsigned short*)region,
                     region + K::W_BYTES / 4, bias, act, out, out_sn, code, out_amax, g.n);
}

void launch_conv_first_fwd(const ConvGeom& g, const float* in, const float* region, const float* bias, int act, float* out, hipStream_t st) {
  DDRL_FIRST_DISPATCH(g, (run_first_fwd<K, false>(g, in, region, bias, act, out, g.out_sn, nullptr, nullptr, st)));
}

// conv + ReLU + max_pool2d(2): pooled [n][64][OH / 2][OH / 2] (dense), code = one decision byte per window
void launch_conv_first_fwd_pool(const ConvGeom& g, const float* in, const float* region, const float* bias, float* pooled, uint8_t* code,
                                float* out_amax, hipStream_t st) {
  DDRL_FIRST_DISPATCH(g, (run_first_fwd<K, true>(g, in, region, bias, 1, pooled, (int64_t)64 * K::PW, code, out_amax, st)));
}

int conv_first_wgrad_splits(const ConvGeom& g) {
  int nbands = 0;
  DDRL_FIRST_DISPATCH(g, nbands = K::NBANDS);
  if (nbands == 0) return 0;
  const int nst = g.n * nbands;
  int s = 512;                                  // two workgroups per CU
  const int cap = (nst + nbands - 1) / nbands;  // at least a sample's worth of bands per workgroup
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

template <class K, bool UNPOOL>
static void run_first_wgrad(const ConvGeom& g, const float* in, const float* dz, int64_t dz_sn, const uint8_t* ucode, float* part, float* dw, float* db,
                            hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)fconv::first_wgrad_kernel<K, UNPOOL>, hipFuncAttributeMaxDynamicSharedMemorySize, K::W_LDS);
    configured = true;
  }
  const int S = conv_first_wgrad_splits(g);
  hipLaunchKernelGGL((fconv::first_wgrad_kernel<K, UNPOOL>), dim3(S), dim3(256), K::W_LDS, st, in, g.in_sn, dz, dz_sn, ucode, part, g.n, S);
  launch_reduce_slabs2(part, S, K::SLAB, (int64_t)64 * K::CIN * K::KK, dw, 64, db, st);
}

void launch_conv_first_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st) {
  DDRL_FIRST_DISPATCH(g, (run_first_wgrad<K, false>(g, in, dz, g.out_sn, nullptr, part, dw, db, st)));
}

// the same from d(pooled) [n][64][OH / 2][OH / 2] + decision bytes
void launch_conv_first_wgrad_pooled(const ConvGeom& g, const float* in, const float* dpool, const uint8_t* ucode, float* part, float* dw, float* db,
                                    hipStream_t st) {
  DDRL_FIRST_DISPATCH(g, (run_first_wgrad<K, true>(g, in, dpool, (int64_t)64 * K::PW, ucode, part, dw, db, st)));
}

}  // namespace ddrl
#endif  // DDRL_PLANES_BF16
