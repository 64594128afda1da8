// Direct stride-1 convolutions of the heavy nav-encoder layers on the 16-bit matrix pipe, fp32-accurate (round 4).
//
// The design of the Atari conv3 kernels (conv2.hip: conv_fwd3_planes_kernel / conv_dgrad3_planes_kernel) as a template over the
// compile-time geometry, replacing the f32-input MFMA templates of dconv.hip for the forward and the data gradient of
//   NavPreNet1D  conv2 64 -> 128 5x5 @22, conv3 128 -> 256 3x3 @10      (reference USTC_lab/nn/nav_encoder.py:96-98)
//   NavPreNet / NavPedPreNet  conv2 64 -> 128 3x3 @24, conv3 128 -> 256 3x3 @12   (nav_encoder.py:18-20, 56-58)
//
//   out[b][r][oy][ox] = sum_{c, ky, kx} Wk[r][c][ky][kx] * in[b][c][oy + ky - PAD][ox + kx - PAD]
//
// Plane scheme (engine2.h "f16x3"): every fp32 operand as two scaled fp16 planes, x S = h0 + h1 to 22 bits, three products per
// k-group on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  The weights arrive pre-split from the pack kernel below with ONE
// power-of-two scale per layer; the input is split while it is staged with ONE power-of-two scale PER SAMPLE, taken from the sample's
// largest magnitude (engine2.h scale_of_amax), so that a sample whose activations or gradients are orders of magnitude below the
// batch's largest keeps its 22 bits; the epilogue multiplies every output column by 1 / (S_sample S_w).  The magnitudes come from
// the PRODUCER of the tensor: the pooling epilogue / the data gradient below leave the largest magnitude of every sample they write
// in `out_amax` (round 5; the pre-passes over the tensors -- sample_amax_kernel, 1-2 GB of reads per PPO iteration of the nav net --
// cost 2.5 of its 21 ms, profiles/README.md r05); a tensor that arrives from elsewhere takes the pre-pass.
//
// rows = 64 output channels per workgroup (MFMA A operand = weight planes), columns = the output pixels of NS WHOLE samples
// (B operand), k-block = 16 input channels, one MFMA k-group = ONE tap x 16 channels: the input block is staged
// channel-innermost into zero-bordered images ([plane][sample][padded pixel][16 channels] fp16, 32 bytes per pixel) that stay for
// all taps, the weights in chunks of TAPC taps ([tap][plane][row 64][lane half 2][8 channels]); every operand is one 16-byte LDS
// read at lane_base + immediate (the tap shift is a compile-time offset inside a chunk).  The data gradient of a stride-1
// convolution is the same computation on dz with flipped, transposed kernels and PAD' = KS - 1 - PAD (the pack kernel does the
// flipping).  Geometry-dependent tile shapes (waves x column tiles per wave) are chosen so that NS samples' pixels fill the 32-wide
// column tiles as well as the register budget allows (table at the bottom): 2 x 400 = 8 x 100 = 25 of 28 tiles, 576 = 4 x 144 = 18 of 18.
#include <cstdlib>

#include "engine2.h"
#include "ops.h"

#ifdef DDRL_PLANES_BF16
// The three-plane build (bf16x6) keeps the f32-input kernels of dconv.hip for these layers: three planes per operand do not fit the LDS
// budgets of the tiles below.  Every query answers "no", nothing launches.
namespace ddrl {
bool conv_has_planes(const ConvGeom&) { return false; }
bool conv_planes_has_pool(const ConvGeom&) { return false; }
int64_t conv_planes_pack_floats(const ConvGeom&) { return 0; }
int conv_planes_wgrad_splits(const ConvGeom&) { return 0; }
void launch_sample_amax(const float*, int64_t, int, int, float*, hipStream_t, int) {}
void launch_conv_planes_pack(const ConvGeom&, const float*, float*, float*, hipStream_t) {}
void launch_conv_planes_fwd(const ConvGeom&, const float*, const float*, float*, const float*, int, float*, hipStream_t) {}
void launch_conv_planes_dgrad(const ConvGeom&, const float*, const float*, float*, float*, hipStream_t) {}
void launch_conv_planes_wgrad(const ConvGeom&, const float*, const float*, float*, float*, float*, hipStream_t) {}
void launch_conv_planes_fwd_pool(const ConvGeom&, const float*, const float*, float*, const float*, const float*, float*, uint8_t*, float*, hipStream_t) {}
void launch_conv_planes_dgrad_pooled(const ConvGeom&, const float*, const uint8_t*, const float*, float*, const float*, float*, float*, hipStream_t) {}
void launch_conv_planes_wgrad_pooled(const ConvGeom&, const float*, const float*, const uint8_t*, const float*, const float*, float*, float*, float*,
                                     hipStream_t) {}
}  // namespace ddrl
#else
namespace ddrl {

namespace pconv {

using u4v = __attribute__((ext_vector_type(4))) unsigned;

template <int CIN_, int COUT_, int KS_, int HIN_, int PAD_, int NS_, int WAVES_, int TN_, int TAPC_>
struct Geo {
  static constexpr int CIN = CIN_, COUT = COUT_, KS = KS_, HIN = HIN_, PAD = PAD_, NS = NS_, WAVES = WAVES_, TN = TN_, TAPC = TAPC_;
  static constexpr int THREADS = 64 * WAVES, KK = KS * KS, NTC = KK / TAPC, KOC = 16, NCB = CIN / KOC, NKB = NCB * NTC;
  static constexpr int OH = HIN + 2 * PAD - KS + 1, P = OH * OH, LP = HIN + 2 * PAD, LPP = LP * LP, RAW = HIN * HIN;
  static constexpr int COLS = NS * P, PIXB = 2 * KOC;                      // bytes per pixel and plane
  static constexpr int IMG_PLANE = NS * LPP * PIXB, W_OFF = NPL * IMG_PLANE, W_BYTES = TAPC * NPL * 64 * 32;
  static constexpr int BIAS_OFF = W_OFF + W_BYTES, SC_OFF = BIAS_OFF + 64 * 4, OMAX_OFF = SC_OFF + ((NS * 4 + 15) / 16) * 16;
  static constexpr size_t LDS_BYTES = OMAX_OFF + ((NS * 4 + 15) / 16) * 16;   // + the largest |output| of every sample of the tile
  static constexpr int NIU = NS * RAW, NIJ = (NIU + THREADS - 1) / THREADS;   // pixel units (16 channels each) per tile / thread
  static constexpr int NWQ = W_BYTES / 16, NWJ = (NWQ + THREADS - 1) / THREADS;
  static_assert(CIN % KOC == 0 && COUT % 64 == 0 && KK % TAPC == 0, "channel blocks of 16, row tiles of 64, whole tap chunks");
  static_assert(TAPC == KK || TAPC == KS, "a weight chunk is the whole kernel or one kernel row");
  static_assert(COLS <= WAVES * TN * 32, "the tile's columns must fit the waves' column tiles");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Largest magnitude of every sample (the pre-pass for tensors whose producer does not leave it).  One workgroup per sample.
__global__ __launch_bounds__(256) void sample_amax_kernel(const float* __restrict__ in, int64_t in_sn, int elems, float* __restrict__ amax,
                                                          int accumulate = 0) {
  __shared__ float red[4];
  const float* src = in + (int64_t)blockIdx.x * in_sn;
  float m = 0.0f;
  for (int i = threadIdx.x * 4; i < elems; i += 1024) {
    const f4 v = ld4(src + i);  // elems is a multiple of 4, the base 16-byte aligned (api_ops.hip direct_ok)
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (accumulate) amax_raise(m, amax + blockIdx.x);   // an `out_amax` filled by a pass over the output: raised, like the epilogues do
    else amax[blockIdx.x] = m;
  }
}

// ---- weights: largest magnitude, then the planes in the order the k index walks ----------------------------------------------
// (the forward and the data-gradient region take the same scale: one pass over the weights leaves it in both headers)
__global__ __launch_bounds__(256) void weight_amax_kernel(const float* __restrict__ w, int64_t count, float* __restrict__ slot, float* __restrict__ slot2) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  amax_update(m, slot);
  amax_update(m, slot2);
}
// dst[row tile][cb][tap][plane][row 64][lane half 2][8 channels] (16-bit); hdr[0] = largest |w| (in), hdr[1] = scale (out).
// dgrad = 0: rows = cout, channels = cin, taps as they are.  dgrad = 1: rows = cin, channels = cout, taps flipped.
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ w, int cin, int cout, int kk, int dgrad,
                                                          unsigned short* __restrict__ dst, float* __restrict__ hdr) {
  const int rows = dgrad ? cin : cout, chans = dgrad ? cout : cin;
  const int64_t total = (int64_t)rows * chans * kk;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float scale = plane_scale(hdr[0]);
  if (i == 0) hdr[1] = scale;
  if (i >= total) return;
  // i -> (row tile, cb, tap, row, half, e) without the plane: planes are 1024 shorts apart inside a (tap) block of NPL * 1024
  const int e = (int)(i & 7), hf = (int)((i >> 3) & 1), r = (int)((i >> 4) & 63);
  const int64_t q = i >> 10;  // (row tile, cb, tap)
  const int ncb = chans / 16;
  const int tap = (int)(q % kk), cb = (int)((q / kk) % ncb), rt = (int)(q / ((int64_t)kk * ncb));
  const int row = rt * 64 + r, ch = cb * 16 + hf * 8 + e;
  const float v = dgrad ? w[((int64_t)ch * cin + row) * kk + (kk - 1 - tap)] : w[((int64_t)row * cin + ch) * kk + tap];
  unsigned short pl[NPL];
  planes_of(v, scale, pl);
  unsigned short* d = dst + q * (NPL * 1024) + r * 16 + hf * 8 + e;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 1024] = pl[p];
}

// POOL (forward of a layer that is followed by ReLU + max_pool2d(2); TN = 4): the column tiles of a wave come in PAIRS over the
// same 32 consecutive pixels of the upper and the lower row of a row pair (the upper rows of the NS samples form one stream of
// NS x OH / 2 x OH pixels; OH is even, so lanes 2 m and 2 m + 1 hold the two columns of one pooling window and neighbouring lanes
// still read neighbouring pixels), the epilogue takes the window's maximum over two accumulators of the lane and the same two of its
// neighbour (one DPP move each) and writes the pooled map [oc][OH / 2][OH / 2] + one decision byte per window (gconv.hip
// maxpool2_fwd_idx_kernel); the full-resolution activations are never written.
// UNPOOL (data gradient of such a layer): `in` is d(pooled) [CIN][HIN / 2][HIN / 2] and `ucode` the layer's decision bytes; the
// staging forms d(pre-activation) on the fly -- the pooled gradient at the window's first maximum under the ReLU's sign, zero
// elsewhere (gconv.hip maxpool2_bwd_idx_kernel) -- so the full-resolution gradient is never written or read either.
#ifndef DDRL_PC_OCC2
#define DDRL_PC_OCC2 1  // four-wave geometries: ask for two waves per SIMD (<= 256 registers), i.e. two RESIDENT workgroups per CU
#endif
template <class K, bool POOL = false, bool UNPOOL = false>
__global__ __launch_bounds__(K::THREADS, (DDRL_PC_OCC2 && K::THREADS <= 256) ? 2 : 1) void direct_planes_kernel(const float* __restrict__ in, int64_t in_sn, const unsigned short* __restrict__ wp,
                                                                   const float* __restrict__ whdr, const float* __restrict__ amax,
                                                                   const float* __restrict__ bias, int act, float* __restrict__ out,
                                                                   int64_t out_sn, uint8_t* __restrict__ code,
                                                                   const uint8_t* __restrict__ ucode, float* __restrict__ out_amax, int n) {
  extern __shared__ __attribute__((aligned(16))) char ldsp[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b0 = blockIdx.x * K::NS, rt = blockIdx.y;
  for (int i = tid; i < K::W_OFF / 16; i += K::THREADS) *(f4*)(ldsp + i * 16) = zero4();  // images incl. their zero borders
  if (tid < 64) ((float*)(ldsp + K::BIAS_OFF))[tid] = bias != nullptr ? bias[rt * 64 + tid] : 0.0f;
  if (tid < K::NS) {
    ((float*)(ldsp + K::SC_OFF))[tid] = scale_of_amax(amax[min(b0 + tid, n - 1)]);
    ((float*)(ldsp + K::OMAX_OFF))[tid] = 0.0f;
  }
  // ---- staging maps.  unit u = tid + THREADS j: sample u / RAW, raw pixel u % RAW -> 16 loads of stride RAW (the k-block's channels)
  // UNPOOL: a unit is a pooling WINDOW (its four pixels share the 16 gradients and decision bytes of the k-block's channels)
  constexpr int CST = UNPOOL ? K::RAW / 4 : K::RAW;  // channel stride of `in`
  constexpr int NIU = UNPOOL ? K::NS * K::RAW / 4 : K::NIU, NIJ = (NIU + K::THREADS - 1) / K::THREADS;
  const float* isrc[NIJ];
  const uint8_t* usrc[NIJ];
  int idst[NIJ];
  float isc[NIJ];
#pragma unroll
  for (int j = 0; j < NIJ; ++j) {
    const int u = min(tid + K::THREADS * j, NIU - 1);
    if (UNPOOL) {
      const int s = u / CST, w = u % CST, wy = w / (K::HIN / 2), wx = w % (K::HIN / 2);
      const int b = min(b0 + s, n - 1);
      isrc[j] = in + (int64_t)b * in_sn + w;  // + (16 cb + c) * RAW / 4
      usrc[j] = ucode + (int64_t)b * ((int64_t)K::CIN * CST) + w;
      idst[j] = (s * K::LPP + (2 * wy + K::PAD) * K::LP + 2 * wx + K::PAD) * K::PIXB;  // pixel (0, 0) of the window
      isc[j] = scale_of_amax(amax[b]);
    } else {
      const int s = u / K::RAW, px = u % K::RAW;
      const int b = min(b0 + s, n - 1);
      isrc[j] = in + (int64_t)b * in_sn + px;  // + (16 cb + c) * RAW
      idst[j] = (s * K::LPP + (px / K::HIN + K::PAD) * K::LP + px % K::HIN + K::PAD) * K::PIXB;
      isc[j] = scale_of_amax(amax[b]);
    }
  }
  const unsigned short* wsrc = wp + (int64_t)rt * ((int64_t)K::NCB * K::KK * NPL * 1024) + tid * 8;  // + kb * TAPC * NPL * 1024 + j * THREADS * 8
  // ---- operand bases
  int aA[2], bB[K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = K::W_OFF + (i * 32 + l31) * 32 + hi * 16;
  constexpr int OW2 = K::OH / 2, PW = OW2 * OW2, SPS = OW2 * K::OH;  // POOL: stream elements (upper-row pixels) per sample
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    if (POOL) {  // tiles (2 jj, 2 jj + 1) = stream elements q .. of the upper / lower rows
      const int q0 = (wc * (K::TN / 2) + (j >> 1)) * 32 + l31, q = q0 < K::NS * SPS ? q0 : 0;
      const int s = q / SPS, e = q % SPS;
      bB[j] = (s * K::LPP + (2 * (e / K::OH) + (j & 1)) * K::LP + e % K::OH) * K::PIXB + hi * 16;
    } else {
      int c = wc * (32 * K::TN) + j * 32 + l31;
      if (c >= K::COLS) c = 0;
      const int s = c / K::P, pix = c % K::P;
      bB[j] = (s * K::LPP + (pix / K::OH) * K::LP + pix % K::OH) * K::PIXB + hi * 16;
    }
  }
  float ir[NIJ][K::KOC];
  uint8_t ic[UNPOOL ? NIJ : 1][K::KOC];
  f4 wr[K::NWJ];
  auto fetch_img = [&](int cb) {
#pragma unroll
    for (int j = 0; j < NIJ; ++j)
#pragma unroll
      for (int c = 0; c < K::KOC; ++c) {
        ir[j][c] = isrc[j][(int64_t)(cb * K::KOC + c) * CST];
        if (UNPOOL) ic[j][c] = usrc[j][(int64_t)(cb * K::KOC + c) * CST];
      }
  };
  auto fetch_w = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + K::THREADS * j < K::NWQ) wr[j] = *(const f4*)(wsrc + (int64_t)kb * (K::TAPC * NPL * 1024) + j * (K::THREADS * 8));
  };
  auto commit_px = [&](char* d, const float (&v)[K::KOC], float sc) {  // one pixel: 16 channels -> 32 bytes per plane
    unsigned pl[K::KOC / 2][NPL];
#pragma unroll
    for (int c = 0; c < K::KOC / 2; ++c) split_planes(v[2 * c], v[2 * c + 1], sc, pl[c]);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      *(u4v*)(d + p * K::IMG_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
      *(u4v*)(d + p * K::IMG_PLANE + 16) = (u4v){pl[4][p], pl[5][p], pl[6][p], pl[7][p]};
    }
  };
  auto commit_img = [&]() {
#pragma unroll
    for (int j = 0; j < NIJ; ++j) {
      if (j + 1 < NIJ || tid + K::THREADS * j < NIU) {
        if (UNPOOL) {  // the window's gradient goes to its first maximum, under the ReLU's sign; the other three pixels get zeros
#pragma unroll
          for (int pos = 0; pos < 4; ++pos) {
            float v[K::KOC];
#pragma unroll
            for (int c = 0; c < K::KOC; ++c) v[c] = (ic[j][c] & 7) == (4 | pos) ? ir[j][c] : 0.0f;
            commit_px(ldsp + idst[j] + ((pos >> 1) * K::LP + (pos & 1)) * K::PIXB, v, isc[j]);
          }
        } else {
          commit_px(ldsp + idst[j], ir[j], isc[j]);
        }
      }
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + K::THREADS * j < K::NWQ) *(f4*)(ldsp + K::W_OFF + (tid + K::THREADS * j) * 16) = wr[j];
  };
  f32x16 acc[2][K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < K::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  fetch_img(0);
  fetch_w(0);
  __syncthreads();  // zero fill complete
  commit_img();
  commit_w();
  if (K::NCB > 1) fetch_img(1);
  if (K::NKB > 1) fetch_w(1);
  __syncthreads();
  for (int cb = 0; cb < K::NCB; ++cb) {
    for (int tc = 0; tc < K::NTC; ++tc) {
      const int kb = cb * K::NTC + tc;
      const int rowoff = (K::TAPC == K::KK) ? 0 : tc * (K::LP * K::PIXB);  // a chunk = one kernel row: ky = tc
#pragma unroll
      for (int t = 0; t < K::TAPC; ++t) {
        const int toff = (K::TAPC == K::KK) ? ((t / K::KS) * K::LP + t % K::KS) * K::PIXB : t * K::PIXB;
        frag8 af[NPL][2], bfr[NPL][K::TN];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[p][i] = *(const frag8*)(ldsp + aA[i] + (t * NPL + p) * 2048);
#pragma unroll
          for (int j = 0; j < K::TN; ++j) bfr[p][j] = *(const frag8*)(ldsp + bB[j] + rowoff + toff + p * K::IMG_PLANE);
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
        commit_w();
        if (kb + 2 < K::NKB) fetch_w(kb + 2);
        if (tc + 1 == K::NTC) {  // the next chunk opens the next channel block
          commit_img();
          if (cb + 2 < K::NCB) fetch_img(cb + 2);
        }
      }
      __syncthreads();
    }
  }
  // ---- epilogue: un-scale per column (sample), bias, ReLU
  const float* lbias = (const float*)(ldsp + K::BIAS_OFF);
  const float* lsc = (const float*)(ldsp + K::SC_OFF);
  float* omax = (float*)(ldsp + K::OMAX_OFF);   // zeroed at the start; the barriers of the k loop lie in between
  const float winv = 1.0f / whdr[1];
  if constexpr (POOL) {
    static_assert(K::TN % 2 == 0 && K::WAVES * (K::TN / 2) * 32 >= K::NS * SPS && K::OH % 2 == 0, "tile pairs over the stream of upper-row pixels");
#pragma unroll
    for (int jj = 0; jj < K::TN / 2; ++jj) {
      const int q = (wc * (K::TN / 2) + jj) * 32 + l31;
      const int s = q / SPS, e = q % SPS, wl = (e / K::OH) * OW2 + (e % K::OH) / 2;
      const bool writer = q < K::NS * SPS && b0 + s < n && !(lane & 1);
      const float inv = winv / lsc[min(s, K::NS - 1)];
      // every lane takes part in the exchange; the even lanes then store under ONE predicate
      float pm[2][16];
      int pc[2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + acc_row(r, hi);
          const float v0 = fmaxf(__builtin_fmaf(acc[i][2 * jj][r], inv, lbias[row]), 0.0f);
          const float v2 = fmaxf(__builtin_fmaf(acc[i][2 * jj + 1][r], inv, lbias[row]), 0.0f);
          const float v1 = lane_swap1(v0), v3 = lane_swap1(v2);  // the window's right column, from the odd lane
          float m = v0;
          int am = 0;
          if (v1 > m) { m = v1; am = 1; }
          if (v2 > m) { m = v2; am = 2; }
          if (v3 > m) { m = v3; am = 3; }
          pm[i][r] = m;
          pc[i][r] = am | (m > 0.0f ? 4 : 0);
        }
      if (out_amax != nullptr && writer) {   // the pooled values are >= 0: their largest is the sample's largest magnitude
        float lm = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) lm = fmaxf(lm, pm[i][r]);
        lds_amax_raise(lm, omax + s);
      }
      if (writer) {
        float* op = out + (int64_t)(b0 + s) * out_sn + (int64_t)(rt * 64 + 4 * hi) * PW + wl;
        uint8_t* cp = code + (int64_t)(b0 + s) * ((int64_t)K::COUT * PW) + (int64_t)(rt * 64 + 4 * hi) * PW + wl;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (i * 32 + (r & 3) + 8 * (r >> 2)) * PW;
            op[ro] = pm[i][r];
            cp[ro] = (uint8_t)pc[i][r];
          }
      }
    }
    if (out_amax != nullptr) {   // wave-uniform
      __syncthreads();
      if (tid < K::NS && b0 + tid < n) amax_raise(omax[tid], out_amax + b0 + tid);   // COUT / 64 row tiles raise the same slot
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    const int c = wc * (32 * K::TN) + j * 32 + l31;
    const int s = c / K::P, pix = c % K::P;
    if (c >= K::COLS || b0 + s >= n) continue;
    const float inv = winv / lsc[s];
    float* op = out + (int64_t)(b0 + s) * out_sn + (int64_t)(rt * 64) * K::P + pix;
    float lm = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + acc_row(r, hi);
        float v = __builtin_fmaf(acc[i][j][r], inv, lbias[row]);
        if (act == 1) v = fmaxf(v, 0.0f);
        op[(int64_t)row * K::P] = v;
        lm = fmaxf(lm, fabsf(v));
      }
    if (out_amax != nullptr) lds_amax_raise(lm, omax + s);
  }
  if (out_amax != nullptr) {   // wave-uniform
    __syncthreads();
    if (tid < K::NS && b0 + tid < n) amax_raise(omax[tid], out_amax + b0 + tid);
  }
}

// ================================================================================================
// Weight gradient of the same layers as plane products -- the design of the Atari conv3 / conv2 weight gradients (wgrad2.hip:
// both blocks staged channel-innermost, MFMA fragments through the transposing LDS read ds_read_b64_tr_b16) over the compile-time
// geometry:
//   part[split][oc][ic][tap] = sum over the split's samples and output pixels of  dz[b][oc][pix] * in[b][ic][oy + ky - PAD][ox + kx - PAD]
// GEMM rows = 64 oc per workgroup, columns = (tap, ic) of an ICW-wide block of input channels, reduction index kappa = output pixel
// of the staged block: NB whole samples, or a BAND of BR output rows of one sample (the input band is BR + KS - 1 padded rows).
// dz as [kappa][64 oc] (128-byte rows, half-swap swizzle), the input as [ic half][padded pixel][32 ic] (64-byte rows; cells
// outside the image are written as zeros by the staging); a lane's 8 k-values are 8 consecutive kappa = rows kappa (dz) and rows
// rho(kappa) + tap offset (input).  Waves: ICW = 64: (oc half, ic half) x all KK taps (3x3: 9 fragment tiles, as the Atari
// kernels); ICW = 32: (oc half, tap half) x 13 / 12 taps (5x5).  Both operands get ONE power-of-two scale for the whole batch (the
// largest of the per-sample maxima the pre-pass leaves): a sum over samples is accurate in the absolute sense, relative to its
// largest contribution.  The bias gradient (sum of dz) rides along in fp32 from the staging registers (ic block 0 only).
// ================================================================================================
#ifndef DDRL_PW_KO
#define DDRL_PW_KO 0  // wgrad_planes_kernel timing knock-outs (1: staging only, 2: matrix work only; tools/build_variant.sh)
#endif
// (Round 5 also measured the taps of a stage as a software-pipelined sequence -- fragments of tap t + 1 requested before the MFMAs of
// tap t, pinned with sched_group_barrier --: no gain, and one pinned region over a whole stage took hipcc 11 minutes; profiles/README.md.)
using s4w = __attribute__((ext_vector_type(4))) short;
__device__ __forceinline__ frag8 tr_frag(const char* lds, int off_lo, int off_hi) {
  typedef s4w __attribute__((address_space(3))) * lds_s4;
  const s4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_lo));
  const s4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s8w;
  return __builtin_bit_cast(frag8, (s8w)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int CIN_, int COUT_, int KS_, int HIN_, int PAD_, int NB_, int BR_, int ICW_>
struct WGeo {
  static constexpr int CIN = CIN_, COUT = COUT_, KS = KS_, HIN = HIN_, PAD = PAD_, NB = NB_, BR = BR_, ICW = ICW_;
  static constexpr int KK = KS * KS, OH = HIN + 2 * PAD - KS + 1, P = OH * OH, LP = HIN + 2 * PAD, RAW = HIN * HIN, KT = CIN * KK;
  static constexpr int NBANDS = OH / BR, KAPPA = NB * BR * OH, NKG = (KAPPA + 15) / 16, AROWS = NKG * 16;
  static constexpr int BRW = BR + KS - 1, BPX = NB * BRW * LP, NH = ICW / 32, BP = 64;
  static constexpr int A_PLANE = AROWS * 128, B_HALF = BPX * BP, B_PLANE = NH * B_HALF, B_OFF = NPL * A_PLANE;
  static constexpr int LDS_BYTES = NPL * (A_PLANE + B_PLANE);
  static constexpr int NT = ICW == 64 ? KK : (KK + 1) / 2;                   // fragment tiles (taps) per wave
  static constexpr int A_UNITS = KAPPA * 8, B_UNITS = BPX * (ICW / 8);       // (row, 8-channel group) staging units
  static constexpr int NA = (A_UNITS + 255) / 256, NBU = (B_UNITS + 255) / 256;
  static constexpr int64_t SLAB = (int64_t)COUT * KT + COUT;
  static_assert(OH % BR == 0 && (NB == 1 || BR == OH), "bands of whole output rows of one sample, or whole samples");
  static_assert(ICW == 32 || ICW == 64, "input-channel block");
  static_assert(CIN % ICW == 0 && COUT % 64 == 0 && LDS_BYTES <= 160 * 1024, "blocks / LDS budget");
  static_assert(KAPPA * 8 * 8 * 4 <= LDS_BYTES, "the bias reduction reuses the stage");
};

// UNPOOL: `dz` is d(pooled) [COUT][OH / 2][OH / 2] of a layer that is followed by ReLU + max_pool2d(2) and `ucode` its decision
// bytes; the staging forms d(pre-activation) on the fly (direct_planes_kernel's UNPOOL).
template <class K, bool UNPOOL = false>
__global__ __launch_bounds__(256) void wgrad_planes_kernel(const float* __restrict__ in, int64_t in_sn, const float* __restrict__ dz, int64_t dz_sn,
                                                           const uint8_t* __restrict__ ucode, const float* __restrict__ sc_in,
                                                           const float* __restrict__ sc_dz, float* __restrict__ part, int n, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char ldsw[];
  __shared__ float s_min[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wi = wave >> 1, wx = wave & 1;
  constexpr int NICB = K::CIN / K::ICW;
  const int icb = blockIdx.x % NICB, split = blockIdx.x / NICB, oct = blockIdx.y;
  // one scale per operand for the whole batch, from the largest of the per-sample magnitudes (sc_dz / sc_in hold magnitudes)
  {
    float ma = 0.0f, mb = 0.0f;
    for (int i = tid; i < n; i += 256) {
      ma = fmaxf(ma, sc_dz[i]);
      mb = fmaxf(mb, sc_in[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ma = fmaxf(ma, __shfl_xor(ma, off, 64));
      mb = fmaxf(mb, __shfl_xor(mb, off, 64));
    }
    if (lane == 0) {
      s_min[0][wave] = ma;
      s_min[1][wave] = mb;
    }
  }
  __syncthreads();
  const float sd = scale_of_amax(fmaxf(fmaxf(s_min[0][0], s_min[0][1]), fmaxf(s_min[0][2], s_min[0][3])));
  const float sa = scale_of_amax(fmaxf(fmaxf(s_min[1][0], s_min[1][1]), fmaxf(s_min[1][2], s_min[1][3])));
  const float inv = 1.0f / (sd * sa);
  const int nsg = (n + K::NB - 1) / K::NB, nst = nsg * K::NBANDS;
  const int per = (nst + nsplit - 1) / nsplit;
  const int st_begin = split * per, st_end = min(nst, st_begin + per);
  // zero rows of the dz image (kappa >= KAPPA): written once
  for (int i = tid; i < NPL * (K::AROWS - K::KAPPA) * 8; i += 256) {
    const int pl = i / ((K::AROWS - K::KAPPA) * 8), r = i % ((K::AROWS - K::KAPPA) * 8);
    *(u4v*)(ldsw + pl * K::A_PLANE + K::KAPPA * 128 + r * 16) = (u4v){0u, 0u, 0u, 0u};
  }
  // ---- staging maps.  UNPOOL: an A unit is (8 channels, one pooling WINDOW): its four pixels share the 8 gradients and decision bytes
  constexpr int ACS = UNPOOL ? K::P / 4 : K::P;                       // channel stride of `dz`
  constexpr int ABAND = UNPOOL ? (K::BR / 2) * (K::OH / 2) : K::BR * K::OH;  // band shift inside a channel
  constexpr int A_UNITS = UNPOOL ? K::KAPPA * 2 : K::A_UNITS, NA = (A_UNITS + 255) / 256;
  int64_t aoff[NA], boff[K::NBU];  // offsets inside a sample (floats), without the band shift
  int awr[NA], awr1[UNPOOL ? NA : 1], bwr[K::NBU], asmp[NA], bsmp[K::NBU], bry[K::NBU];
  unsigned bxok = 0u;                 // bit t: the unit's column is inside the image
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int u = min(tid + 256 * t, A_UNITS - 1);
    if (UNPOOL) {
      static_assert(!UNPOOL || (K::BR % 2 == 0 && K::OH % 2 == 0), "bands of whole row pairs");
      constexpr int WPS = (K::BR / 2) * (K::OH / 2), NW = K::KAPPA / 4;  // windows per sample band / per stage
      const int c8 = u / NW, win = u % NW, bl = win / WPS, wr = win % WPS, wy = wr / (K::OH / 2), wx = wr % (K::OH / 2);
      asmp[t] = bl;
      aoff[t] = (int64_t)(oct * 64 + c8 * 8) * ACS + wr;                      // + band * (BR / 2) * (OH / 2), + sample * dz_sn, + c * P / 4
      const int k0 = bl * (K::BR * K::OH) + 2 * wy * K::OH + 2 * wx, k1 = k0 + K::OH;  // even: k and k + 1 share the swizzle
      awr[t] = k0 * 128 + ((c8 * 16) ^ (((k0 >> 1) & 1) * 64));
      awr1[t] = k1 * 128 + ((c8 * 16) ^ (((k1 >> 1) & 1) * 64));
    } else {
      const int c8 = u / K::KAPPA, kap = u % K::KAPPA, bl = kap / (K::BR * K::OH), w = kap % (K::BR * K::OH);
      asmp[t] = bl;
      aoff[t] = (int64_t)(oct * 64 + c8 * 8) * K::P + w;                      // + band * BR * OH, + sample * dz_sn, + c * P
      awr[t] = kap * 128 + ((c8 * 16) ^ (((kap >> 1) & 1) * 64));
    }
  }
#pragma unroll
  for (int t = 0; t < K::NBU; ++t) {
    const int u = min(tid + 256 * t, K::B_UNITS - 1);
    const int c8 = u / K::BPX, rho = u % K::BPX, bl = rho / (K::BRW * K::LP), r = rho % (K::BRW * K::LP);
    const int ry = r / K::LP, rx = r % K::LP, ix = rx - K::PAD;
    bsmp[t] = bl;
    bry[t] = ry - K::PAD;                                                     // input row = band * BR + bry
    if (ix >= 0 && ix < K::HIN) bxok |= 1u << t;
    boff[t] = (int64_t)(icb * K::ICW + c8 * 8) * K::RAW + min(max(ix, 0), K::HIN - 1);  // + iy * HIN, + sample * in_sn, + c * RAW
    bwr[t] = K::B_OFF + (c8 >> 2) * K::B_HALF + rho * K::BP + (c8 & 3) * 16;
  }
  // ---- fragment addresses (wgrad2.hip): 16-lane group g16: columns 16 (g16 & 1) .. +15 of the 32-channel fragment, k-values
  // 8 (g16 >> 1) .. +7; inside the group lane 4 q + pp supplies row q (first read) / q + 4 (second), chunk pp
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = (q >> 1) & 1;
  const int a_lane = (8 * (g16 >> 1) + q) * 128 + (((wi ^ sw) * 64) + (g16 & 1) * 32 + pp * 8);
  const int b_lane = K::B_OFF + (K::ICW == 64 ? wx * K::B_HALF : 0) + (g16 & 1) * 32 + pp * 8;
  int brow[K::NKG][2];
#pragma unroll
  for (int g = 0; g < K::NKG; ++g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kap = 16 * g + 8 * (g16 >> 1) + q + 4 * r;
      const int bl = kap / (K::BR * K::OH), w = kap % (K::BR * K::OH);
      brow[g][r] = kap < K::KAPPA ? ((bl * K::BRW + w / K::OH) * K::LP + w % K::OH) * K::BP : 0;  // padded kappa: any row (dz is zero there)
    }
  float ar[NA][8], br[K::NBU][8], bsum[NA][8];
  uint8_t ac[UNPOOL ? NA : 1][8];
#pragma unroll
  for (int t = 0; t < NA; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) bsum[t][c] = 0.0f;
  auto fetch = [&](int st) {
    const int sg = st / K::NBANDS, band = st % K::NBANDS;
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int64_t o = (int64_t)min(sg * K::NB + asmp[t], n - 1) * dz_sn + aoff[t] + band * ABAND;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        ar[t][c] = dz[o + (int64_t)c * ACS];
        if (UNPOOL) ac[t][c] = ucode[o + (int64_t)c * ACS];
      }
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      const int iy = min(max(band * K::BR + bry[t], 0), K::HIN - 1);
      const float* src = in + (int64_t)min(sg * K::NB + bsmp[t], n - 1) * in_sn + boff[t] + iy * K::HIN;
#pragma unroll
      for (int c = 0; c < 8; ++c) br[t][c] = src[(int64_t)c * K::RAW];
    }
  };
  auto commit = [&](int st) {
    const int sg = st / K::NBANDS, band = st % K::NBANDS;
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      if (t + 1 < NA || tid + 256 * t < A_UNITS) {
        if (sg * K::NB + asmp[t] >= n) {  // missing sample of a ragged last stage: contributes zero
#pragma unroll
          for (int c = 0; c < 8; ++c) ar[t][c] = 0.0f;
        }
        auto put = [&](char* d, const float (&v)[8]) {
          unsigned pl[4][NPL];
#pragma unroll
          for (int c = 0; c < 4; ++c) split_planes(v[2 * c], v[2 * c + 1], sd, pl[c]);
#pragma unroll
          for (int p = 0; p < NPL; ++p) *(u4v*)(d + p * K::A_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
        };
        if (UNPOOL) {  // the window's gradient goes to its first maximum, under the ReLU's sign; the other three pixels get zeros
#pragma unroll
          for (int pos = 0; pos < 4; ++pos) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (ac[t][c] & 7) == (4 | pos) ? ar[t][c] : 0.0f;
            put(ldsw + ((pos >> 1) ? awr1[t] : awr[t]) + (pos & 1) * 128, v);
          }
#pragma unroll
          for (int c = 0; c < 8; ++c) bsum[t][c] += (ac[t][c] & 4) ? ar[t][c] : 0.0f;
        } else {
          put(ldsw + awr[t], ar[t]);
#pragma unroll
          for (int c = 0; c < 8; ++c) bsum[t][c] += ar[t][c];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      if (t + 1 < K::NBU || tid + 256 * t < K::B_UNITS) {
        const int iy = band * K::BR + bry[t];
        const bool ok = ((bxok >> t) & 1u) && iy >= 0 && iy < K::HIN && sg * K::NB + bsmp[t] < n;
        unsigned pl[4][NPL];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes(ok ? br[t][2 * c] : 0.0f, ok ? br[t][2 * c + 1] : 0.0f, sa, pl[c]);
        char* d = ldsw + bwr[t];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4v*)(d + p * K::B_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
      }
    }
  };
  f32x16 acc[K::NT];
#pragma unroll
  for (int t = 0; t < K::NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  if (st_begin < st_end) {
    fetch(st_begin);
    commit(st_begin);
    if (st_begin + 1 < st_end) fetch(st_begin + 1);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
#if DDRL_PW_KO != 1   // timing-only knock-out 1: no fragment reads, no matrix instructions (results are WRONG)
#pragma unroll
      for (int g = 0; g < K::NKG; ++g) {
        frag8 a[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) a[p] = tr_frag(ldsw, a_lane + p * K::A_PLANE + g * 2048, a_lane + p * K::A_PLANE + g * 2048 + 512);
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < K::NT; ++t) {
          const int tap = K::ICW == 64 ? t : wx * K::NT + t;   // ICW = 32: the wave's half of the taps (wx is wave-uniform)
          if (K::ICW == 32 && tap >= K::KK) continue;
          const int toff = ((tap / K::KS) * K::LP + tap % K::KS) * K::BP;
          frag8 b[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) b[p] = tr_frag(ldsw, b_lane + p * K::B_PLANE + brow[g][0] + toff, b_lane + p * K::B_PLANE + brow[g][1] + toff);
#pragma unroll
          for (int m = 0; m < NPROD; ++m) acc[t] = mfma_planes(a[PA[m]], b[PB[m]], acc[t]);
        }
      }
#endif
      __syncthreads();  // every wave is done with the stage
#if DDRL_PW_KO != 2   // timing-only knock-out 2: the first stage is multiplied again and again (no loads, no splits, no LDS stores)
      if (st + 1 < st_end) {
        commit(st + 1);
        if (st + 2 < st_end) fetch(st + 2);
      }
#endif
      __syncthreads();
    }
  }
  // ---- epilogue: slab[oc][ic][tap] (torch layout), then the bias partial (ic block 0)
  float* slab = part + (int64_t)split * K::SLAB;
#pragma unroll
  for (int t = 0; t < K::NT; ++t) {
    const int tap = K::ICW == 64 ? t : wx * K::NT + t;
    if (K::ICW == 32 && tap >= K::KK) continue;
    const int ic = icb * K::ICW + (K::ICW == 64 ? wx * 32 : 0) + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(int64_t)(oct * 64 + wi * 32 + acc_row(r, hi)) * K::KT + ic * K::KK + tap] = acc[t][r] * inv;
  }
  if (icb != 0) return;  // block-uniform
  __syncthreads();
  float* red = (float*)ldsw;  // [unit][8]
#pragma unroll
  for (int t = 0; t < NA; ++t)
    if (t + 1 < NA || tid + 256 * t < A_UNITS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[(tid + 256 * t) * 8 + c] = bsum[t][c];
    }
  __syncthreads();
  if (tid < 64) {  // oc = oct * 64 + tid: units (c8 = tid / 8) * UPC .. + UPC - 1, channel tid % 8
    constexpr int UPC = A_UNITS / 8;  // units per channel group: the stage's pixels, or its windows (UNPOOL)
    float sacc = 0.0f;
    for (int k = 0; k < UPC; ++k) sacc += red[((tid >> 3) * UPC + k) * 8 + (tid & 7)];
    slab[(int64_t)K::COUT * K::KT + oct * 64 + tid] = sacc;
  }
}

}  // namespace pconv

// ---- dispatch table (same layers as dconv.hip's forward / data-gradient instantiations) -------------------------------------------
// Tile shapes.  The 18-tile geometries (576 = 4 x 144 columns) run 6 waves x 3 tiles (96 accumulators), two workgroups per CU; for the
// 400- / 100-pixel layers see the A/B note below (a 5-wave workgroup would fit 25 column tiles exactly, but two of its waves share a
// SIMD and the compiler then has 256 registers per wave for 160 accumulators + fragments + prefetch).
//                        CIN  COUT KS HIN PAD NS WAVES TN TAPC
// Same-box A/B (tools/ab_nav.py, profiles/README.md r04): whole samples filling the column tiles better (2 x 400 = 8 x 100 columns on
// four waves x 7 tiles: 25 of 28 used, 224 accumulators, 94-111 KB of LDS -> ONE workgroup per CU) against one / four samples on
// four waves x 4 tiles (400 of 512 columns, 128 accumulators, 57-62 KB -> TWO workgroups per CU): the second is faster although it
// issues 14 % more matrix instructions (5x5 forward 4.29 -> 4.12 ms, 3x3 data gradient 1.81 -> 1.59 ms per 4,096 samples): at one
// workgroup per CU nothing covers the barriers and commits of a k-block.  -DDDRL_PC_WIDE=1 selects the wide tiles.
// Later in the round: the 3x3 @10 layer with FIVE samples per workgroup (500 of 512 columns, 59 KB; nav iteration 23.39 -> 23.05 ms).
// Then the register count: the four-wave kernels compiled to 320-370 registers, so that in spite of their LDS footprint only ONE
// workgroup per CU was ever resident.  With the allocator asked for two waves per SIMD (launch bounds; 234-256 registers, no spills but
// 38 in the 5x5 data gradient) two really are: nav iteration 22.71 -> 21.89 ms.  Under it the 5x5 forward on four waves x two tile pairs
// (3.41 ms) beats the seven-wave form that had won before (3.54; 4.04 -> 3.70 against the one-resident-workgroup four-wave form).
#ifndef DDRL_PC_WIDE
#define DDRL_PC_WIDE 0
#endif
#if DDRL_PC_WIDE
using PN1dC2F = pconv::Geo<64, 128, 5, 22, 1, 2, 4, 7, 5>;    // 2 x 400 columns = 25 of 28 column tiles
using PN1dC3F = pconv::Geo<128, 256, 3, 10, 1, 8, 4, 7, 3>;   // 8 x 100 columns
using PN1dC3D = pconv::Geo<256, 128, 3, 10, 1, 8, 4, 7, 3>;
#else
#ifndef DDRL_PC2_W7
#define DDRL_PC2_W7 0  // 1: seven waves x one tile pair (400 of 448 columns); see the A/B notes below
#endif
#if DDRL_PC2_W7
using PN1dC2F = pconv::Geo<64, 128, 5, 22, 1, 1, 7, 2, 5>;    // 400 of 448 columns, seven waves x one tile pair
#else
using PN1dC2F = pconv::Geo<64, 128, 5, 22, 1, 1, 4, 4, 5>;    // 400 of 512 columns
#endif
#ifndef DDRL_PC3_NS
#define DDRL_PC3_NS 5  // samples per workgroup of the 3x3 @10 layer: 5 x 100 = 500 of 512 columns (4: 400 of 512)
#endif
using PN1dC3F = pconv::Geo<128, 256, 3, 10, 1, DDRL_PC3_NS, 4, 4, 3>;
using PN1dC3D = pconv::Geo<256, 128, 3, 10, 1, DDRL_PC3_NS, 4, 4, 3>;
#endif
using PN1dC2D = pconv::Geo<128, 64, 5, 20, 3, 1, 4, 4, 5>;    // 484 of 512 columns
using PNavC2F = pconv::Geo<64, 128, 3, 24, 1, 1, 6, 3, 3>;    // 576 columns = 18 column tiles
using PNavC2D = pconv::Geo<128, 64, 3, 24, 1, 1, 6, 3, 3>;
using PNavC3F = pconv::Geo<128, 256, 3, 12, 1, 4, 6, 3, 3>;   // 4 x 144 columns
using PNavC3D = pconv::Geo<256, 128, 3, 12, 1, 4, 6, 3, 3>;
// the same forwards with ReLU + max-pool in the epilogue: tile PAIRS over the 288-pixel stream of upper rows (9 stream tiles) need an even
// number of column tiles per wave: five waves x (2 pairs) = 10 stream tiles, 128 accumulators (six waves x 3 tiles cannot pair up)
using PNavC2FP = pconv::Geo<64, 128, 3, 24, 1, 1, 5, 4, 3>;
using PNavC3FP = pconv::Geo<128, 256, 3, 12, 1, 4, 5, 4, 3>;
// AtariPreNet's conv3 as an operator (64 -> 64 3x3 @9, no padding; the fused Atari path has its own kernels in conv2.hip): 5 samples =
// 245 of 256 columns; data gradient: dz 7x7 with PAD' = 2 -> 9x9 = 81 columns per sample, 3 samples = 243 of 256
using PAtC3F = pconv::Geo<64, 64, 3, 9, 0, 5, 4, 2, 9>;
using PAtC3D = pconv::Geo<64, 64, 3, 7, 2, 3, 4, 2, 9>;

enum PlanesId { kPNone = -1, kPN1dC2, kPN1dC3, kPNavC2, kPNavC3, kPAtC3 };

static PlanesId planes_id(const ConvGeom& g) {
#ifdef DDRL_PLANES_BF16
  // the three-plane diagnostic build (accuracy attribution only, never the shipped library) has no specialised plane kernels: every
  // nav layer then runs on the generic gather kernels of gconv.hip (f32 inputs, three bf16 planes: several times slower)
  return kPNone;
#else
  if (g.stride != 1 || g.h != g.w || g.kh != g.kw || g.pad_h != g.pad_w) return kPNone;
  const auto is = [&](int cin, int cout, int ks, int h) { return g.cin == cin && g.cout == cout && g.kh == ks && g.h == h; };
  if (g.pad_h == 0) return is(64, 64, 3, 9) ? kPAtC3 : kPNone;
  if (g.pad_h != 1) return kPNone;
  if (is(64, 128, 5, 22)) return kPN1dC2;
  if (is(128, 256, 3, 10)) return kPN1dC3;
  if (is(64, 128, 3, 24)) return kPNavC2;
  if (is(128, 256, 3, 12)) return kPNavC3;
  return kPNone;
#endif
}

bool conv_has_planes(const ConvGeom& g) { return planes_id(g) != kPNone; }

// per-sample power-of-two plane scales of x[n][elems] (sample stride sn): what the plane kernels compute in their pre-pass
void launch_sample_amax(const float* x, int64_t sn, int elems, int n, float* amax, hipStream_t st, int accumulate) {
  hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)n), dim3(256), 0, st, x, sn, elems, amax, accumulate);
}

// floats of ONE packed region (forward or data gradient): the planes (2 bytes x 2 planes per weight = 4 bytes) + a 64-float header
int64_t conv_planes_pack_floats(const ConvGeom& g) { return (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2 + 64; }

void launch_conv_planes_pack(const ConvGeom& g, const float* w, float* wpf, float* wpd, hipStream_t st) {
  const int kk = g.kh * g.kw;
  const int64_t total = (int64_t)g.cout * g.cin * kk;
  const int64_t planes = total * NPL / 2;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  (void)hipMemsetAsync(wpf + planes, 0, 64 * sizeof(float), st);
  (void)hipMemsetAsync(wpd + planes, 0, 64 * sizeof(float), st);
  const unsigned ablocks = (unsigned)((total + 4095) / 4096 < 512 ? (total + 4095) / 4096 : 512);  // ~16 elements per thread
  hipLaunchKernelGGL(pconv::weight_amax_kernel, dim3(ablocks), dim3(256), 0, st, w, total, wpf + planes, wpd + planes);
  for (int dg = 0; dg < 2; ++dg) {
    float* region = dg ? wpd : wpf;
    float* hdr = region + planes;
    hipLaunchKernelGGL(pconv::pack_planes_kernel, dim3(blocks), dim3(256), 0, st, w, g.cin, g.cout, kk, dg, (unsigned short*)region, hdr);
  }
}

template <class K>
static void run_planes(const float* in, int64_t in_sn, const float* region, int64_t planes, float* scales, const float* bias, int act,
                       float* out, int64_t out_sn, int n, hipStream_t st, float* out_amax = nullptr) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::direct_planes_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)n), dim3(256), 0, st, in, in_sn, K::CIN * K::RAW, scales, 0);
  hipLaunchKernelGGL((pconv::direct_planes_kernel<K, false>), dim3((unsigned)((n + K::NS - 1) / K::NS), K::COUT / 64, 1), dim3(K::THREADS), K::LDS_BYTES, st, in,
                     in_sn, (const unsigned short*)region, region + planes, scales, bias, act, out, out_sn, (uint8_t*)nullptr, (const uint8_t*)nullptr, out_amax, n);
}

template <class K>
static void run_planes_pool(const float* in, int64_t in_sn, const float* region, int64_t planes, float* scales, const float* given, const float* bias,
                            float* pooled, uint8_t* code, float* out_amax, int n, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::direct_planes_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  if (!given) hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)n), dim3(256), 0, st, in, in_sn, K::CIN * K::RAW, scales, 0);
  hipLaunchKernelGGL((pconv::direct_planes_kernel<K, true>), dim3((unsigned)((n + K::NS - 1) / K::NS), K::COUT / 64, 1), dim3(K::THREADS), K::LDS_BYTES, st, in,
                     in_sn, (const unsigned short*)region, region + planes, given ? given : scales, bias, 1, pooled, (int64_t)K::COUT * (K::P / 4), code, (const uint8_t*)nullptr, out_amax, n);
}

// data gradient straight from d(pooled) + decision bytes (K = the layer's data-gradient geometry: CIN = dz channels, HIN = dz size).
// The per-sample scales come from d(pooled) itself: max |d(pooled)| bounds max |dz| (the routing only drops elements).
template <class K>
static void run_planes_unpool(const float* dpool, const uint8_t* ucode, const float* region, int64_t planes, float* scales, const float* given,
                              float* din, int64_t din_sn, float* out_amax, int n, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::direct_planes_kernel<K, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  constexpr int64_t PSN = (int64_t)K::CIN * K::RAW / 4;
  if (!given) hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)n), dim3(256), 0, st, dpool, PSN, (int)PSN, scales, 0);
  hipLaunchKernelGGL((pconv::direct_planes_kernel<K, false, true>), dim3((unsigned)((n + K::NS - 1) / K::NS), K::COUT / 64, 1), dim3(K::THREADS), K::LDS_BYTES,
                     st, dpool, PSN, (const unsigned short*)region, region + planes, given ? given : scales, (const float*)nullptr, 0, din, din_sn, (uint8_t*)nullptr, ucode, out_amax, n);
}

// conv + ReLU + max_pool2d(2) in one launch for the layers whose tiles allow it (four column tiles per wave)
bool conv_planes_has_pool(const ConvGeom& g) {
#if DDRL_PC_WIDE
  return false;
#else
  const PlanesId id = planes_id(g);
  return id == kPN1dC2 || id == kPN1dC3 || id == kPNavC2 || id == kPNavC3;
#endif
}

void launch_conv_planes_fwd_pool(const ConvGeom& g, const float* in, const float* wpf, float* scales, const float* given, const float* bias,
                                 float* pooled, uint8_t* code, float* out_amax, hipStream_t st) {
#if !DDRL_PC_WIDE
  const int64_t planes = (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes_pool<PN1dC2F>(in, g.in_sn, wpf, planes, scales, given, bias, pooled, code, out_amax, g.n, st); break;
    case kPN1dC3: run_planes_pool<PN1dC3F>(in, g.in_sn, wpf, planes, scales, given, bias, pooled, code, out_amax, g.n, st); break;
    case kPNavC2: run_planes_pool<PNavC2FP>(in, g.in_sn, wpf, planes, scales, given, bias, pooled, code, out_amax, g.n, st); break;
    case kPNavC3: run_planes_pool<PNavC3FP>(in, g.in_sn, wpf, planes, scales, given, bias, pooled, code, out_amax, g.n, st); break;
    default: break;
  }
#endif
}

void launch_conv_planes_fwd(const ConvGeom& g, const float* in, const float* wpf, float* scales, const float* bias, int act, float* out,
                            hipStream_t st) {
  const int64_t planes = (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes<PN1dC2F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPN1dC3: run_planes<PN1dC3F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPNavC2: run_planes<PNavC2F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPNavC3: run_planes<PNavC3F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPAtC3: run_planes<PAtC3F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    default: break;
  }
}

void launch_conv_planes_dgrad(const ConvGeom& g, const float* dz, const float* wpd, float* scales, float* din, hipStream_t st) {
  const int64_t planes = (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes<PN1dC2D>(dz, g.out_sn, wpd, planes, scales, nullptr, 0, din, g.in_sn, g.n, st); break;
    case kPN1dC3: run_planes<PN1dC3D>(dz, g.out_sn, wpd, planes, scales, nullptr, 0, din, g.in_sn, g.n, st); break;
    case kPNavC2: run_planes<PNavC2D>(dz, g.out_sn, wpd, planes, scales, nullptr, 0, din, g.in_sn, g.n, st); break;
    case kPNavC3: run_planes<PNavC3D>(dz, g.out_sn, wpd, planes, scales, nullptr, 0, din, g.in_sn, g.n, st); break;
    case kPAtC3: run_planes<PAtC3D>(dz, g.out_sn, wpd, planes, scales, nullptr, 0, din, g.in_sn, g.n, st); break;
    default: break;
  }
}


// ---- weight gradients ----------------------------------------------------------------------------------------------------------------
//                          CIN  COUT KS HIN PAD NB BR ICW
// (5x5 A/B, round 4: three tap groups of 9 / 9 / 7 taps across workgroups with (oc half, ic half) waves -- 144 accumulators, two
// workgroups per CU, the dz band staged three times -- measured 6.3 against 4.45 ms: not kept)
using PN1dC2W = pconv::WGeo<64, 128, 5, 22, 1, 1, 4, 32>;    // bands of 4 output rows: 80 kappa = 5 k-groups; wave = (oc half, 13 / 12 taps)
using PN1dC3W = pconv::WGeo<128, 256, 3, 10, 1, 2, 10, 64>;  // 2 whole samples: 200 kappa of 208
using PNavC2W = pconv::WGeo<64, 128, 3, 24, 1, 1, 4, 64>;    // bands of 4 rows: 96 kappa = 6 k-groups
using PNavC3W = pconv::WGeo<128, 256, 3, 12, 1, 1, 12, 64>;  // one whole sample: 144 kappa = 9 k-groups
using PAtC3W = pconv::WGeo<64, 64, 3, 9, 0, 2, 7, 64>;       // AtariPreNet conv3 as an operator: 2 whole samples, 98 kappa of 112

template <class K>
static int wgrad_splits_of(int n) {
  const int tiles = (K::CIN / K::ICW) * (K::COUT / 64);
  const int nst = ((n + K::NB - 1) / K::NB) * K::NBANDS;
  int s = (768 + tiles - 1) / tiles;            // about three workgroups per CU in flight, where LDS and registers allow them
  const int cap = (nst + 7) / 8;                // at least eight stages per workgroup (prologue / epilogue / slab traffic)
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

int conv_planes_wgrad_splits(const ConvGeom& g) {
  switch (planes_id(g)) {
    case kPN1dC2: return wgrad_splits_of<PN1dC2W>(g.n);
    case kPN1dC3: return wgrad_splits_of<PN1dC3W>(g.n);
    case kPNavC2: return wgrad_splits_of<PNavC2W>(g.n);
    case kPNavC3: return wgrad_splits_of<PNavC3W>(g.n);
    case kPAtC3: return wgrad_splits_of<PAtC3W>(g.n);
    default: return 0;
  }
}

template <class K>
static void run_planes_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* scales, int S, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::wgrad_planes_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  float* sc_in = scales;
  float* sc_dz = scales + g.n;
  hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)g.n), dim3(256), 0, st, in, g.in_sn, K::CIN * K::RAW, sc_in, 0);
  hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)g.n), dim3(256), 0, st, dz, g.out_sn, K::COUT * K::P, sc_dz, 0);
  hipLaunchKernelGGL(pconv::wgrad_planes_kernel<K>, dim3((unsigned)((K::CIN / K::ICW) * S), K::COUT / 64, 1), dim3(256), K::LDS_BYTES, st, in, g.in_sn,
                     dz, g.out_sn, (const uint8_t*)nullptr, sc_in, sc_dz, part, g.n, S);
}

template <class K>
static void run_planes_wgrad_pooled(const ConvGeom& g, const float* in, const float* dpool, const uint8_t* ucode, float* part, float* scales,
                                    const float* given_in, const float* given_dp, int S, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::wgrad_planes_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  float* sc_in = scales;
  float* sc_dz = scales + g.n;
  constexpr int64_t PSN = (int64_t)K::COUT * K::P / 4;
  if (!given_in) hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)g.n), dim3(256), 0, st, in, g.in_sn, K::CIN * K::RAW, sc_in, 0);
  if (!given_dp) hipLaunchKernelGGL(pconv::sample_amax_kernel, dim3((unsigned)g.n), dim3(256), 0, st, dpool, PSN, (int)PSN, sc_dz, 0);  // max |d(pooled)| bounds max |dz|
  hipLaunchKernelGGL((pconv::wgrad_planes_kernel<K, true>), dim3((unsigned)((K::CIN / K::ICW) * S), K::COUT / 64, 1), dim3(256), K::LDS_BYTES, st, in,
                     g.in_sn, dpool, PSN, ucode, given_in ? given_in : sc_in, given_dp ? given_dp : sc_dz, part, g.n, S);
}

// weight / data gradient of a pooled layer straight from d(pooled) + decision bytes (the layers of conv_planes_has_pool)
void launch_conv_planes_wgrad_pooled(const ConvGeom& g, const float* in, const float* dpool, const uint8_t* ucode, const float* given_in,
                                     const float* given_dp, float* part, float* dw, float* db, hipStream_t st) {
#if !DDRL_PC_WIDE
  const int S = conv_planes_wgrad_splits(g);
  const int KT = g.cin * g.kh * g.kw;
  const int64_t slab = (int64_t)g.cout * KT + g.cout;
  float* scales = part + (int64_t)S * slab;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes_wgrad_pooled<PN1dC2W>(g, in, dpool, ucode, part, scales, given_in, given_dp, S, st); break;
    case kPN1dC3: run_planes_wgrad_pooled<PN1dC3W>(g, in, dpool, ucode, part, scales, given_in, given_dp, S, st); break;
    case kPNavC2: run_planes_wgrad_pooled<PNavC2W>(g, in, dpool, ucode, part, scales, given_in, given_dp, S, st); break;
    case kPNavC3: run_planes_wgrad_pooled<PNavC3W>(g, in, dpool, ucode, part, scales, given_in, given_dp, S, st); break;
    default: return;
  }
  launch_reduce_slabs2(part, S, slab, (int64_t)g.cout * KT, dw, g.cout, db, st);
#endif
}

void launch_conv_planes_dgrad_pooled(const ConvGeom& g, const float* dpool, const uint8_t* ucode, const float* wpd, float* scales, const float* given,
                                     float* din, float* out_amax, hipStream_t st) {
#if !DDRL_PC_WIDE
  const int64_t planes = (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes_unpool<PN1dC2D>(dpool, ucode, wpd, planes, scales, given, din, g.in_sn, out_amax, g.n, st); break;
    case kPN1dC3: run_planes_unpool<PN1dC3D>(dpool, ucode, wpd, planes, scales, given, din, g.in_sn, out_amax, g.n, st); break;
    case kPNavC2: run_planes_unpool<PNavC2D>(dpool, ucode, wpd, planes, scales, given, din, g.in_sn, out_amax, g.n, st); break;
    case kPNavC3: run_planes_unpool<PNavC3D>(dpool, ucode, wpd, planes, scales, given, din, g.in_sn, out_amax, g.n, st); break;
    default: break;
  }
#endif
}

// part: S slabs of COUT * KT + COUT floats, then 2 n floats of scratch for the per-sample scales
void launch_conv_planes_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st) {
  const int S = conv_planes_wgrad_splits(g);
  const int KT = g.cin * g.kh * g.kw;
  const int64_t slab = (int64_t)g.cout * KT + g.cout;
  float* scales = part + (int64_t)S * slab;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes_wgrad<PN1dC2W>(g, in, dz, part, scales, S, st); break;
    case kPN1dC3: run_planes_wgrad<PN1dC3W>(g, in, dz, part, scales, S, st); break;
    case kPNavC2: run_planes_wgrad<PNavC2W>(g, in, dz, part, scales, S, st); break;
    case kPNavC3: run_planes_wgrad<PNavC3W>(g, in, dz, part, scales, S, st); break;
    case kPAtC3: run_planes_wgrad<PAtC3W>(g, in, dz, part, scales, S, st); break;
    default: return;
  }
  launch_reduce_slabs2(part, S, slab, (int64_t)g.cout * KT, dw, g.cout, db, st);
}

}  // namespace ddrl
#endif  // DDRL_PLANES_BF16
