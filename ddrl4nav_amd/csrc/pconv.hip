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
// power-of-two scale per layer; the input is split while it is staged with ONE power-of-two scale PER SAMPLE (its largest
// magnitude, left by sample_scale_kernel in a pre-pass over the input: 2 % of the convolution's time), so that a sample whose
// activations or gradients are orders of magnitude below the batch's largest keeps its 22 bits; the epilogue multiplies every
// output column by 1 / (S_sample S_w).
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
  static constexpr int BIAS_OFF = W_OFF + W_BYTES, SC_OFF = BIAS_OFF + 64 * 4;
  static constexpr size_t LDS_BYTES = SC_OFF + ((NS * 4 + 15) / 16) * 16;
  static constexpr int NIU = NS * RAW, NIJ = (NIU + THREADS - 1) / THREADS;   // pixel units (16 channels each) per tile / thread
  static constexpr int NWQ = W_BYTES / 16, NWJ = (NWQ + THREADS - 1) / THREADS;
  static_assert(CIN % KOC == 0 && COUT % 64 == 0 && KK % TAPC == 0, "channel blocks of 16, row tiles of 64, whole tap chunks");
  static_assert(TAPC == KK || TAPC == KS, "a weight chunk is the whole kernel or one kernel row");
  static_assert(COLS <= WAVES * TN * 32, "the tile's columns must fit the waves' column tiles");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Largest magnitude of every sample -> the power-of-two scale of its fp16 planes.  One workgroup per sample.
__global__ __launch_bounds__(256) void sample_scale_kernel(const float* __restrict__ in, int64_t in_sn, int elems, float* __restrict__ scales) {
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
  if (threadIdx.x == 0) scales[blockIdx.x] = plane_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---- weights: largest magnitude, then the planes in the order the k index walks ----------------------------------------------
__global__ __launch_bounds__(256) void weight_amax_kernel(const float* __restrict__ w, int64_t count, float* __restrict__ slot) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  amax_update(m, slot);
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

template <class K>
__global__ __launch_bounds__(K::THREADS) void direct_planes_kernel(const float* __restrict__ in, int64_t in_sn, const unsigned short* __restrict__ wp,
                                                                   const float* __restrict__ whdr, const float* __restrict__ scales,
                                                                   const float* __restrict__ bias, int act, float* __restrict__ out,
                                                                   int64_t out_sn, int n) {
  extern __shared__ __attribute__((aligned(16))) char ldsp[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b0 = blockIdx.x * K::NS, rt = blockIdx.y;
  for (int i = tid; i < K::W_OFF / 16; i += K::THREADS) *(f4*)(ldsp + i * 16) = zero4();  // images incl. their zero borders
  if (tid < 64) ((float*)(ldsp + K::BIAS_OFF))[tid] = bias != nullptr ? bias[rt * 64 + tid] : 0.0f;
  if (tid < K::NS) ((float*)(ldsp + K::SC_OFF))[tid] = scales[min(b0 + tid, n - 1)];
  // ---- staging maps.  unit u = tid + THREADS j: sample u / RAW, raw pixel u % RAW -> 16 loads of stride RAW (the k-block's channels)
  const float* isrc[K::NIJ];
  int idst[K::NIJ];
  float isc[K::NIJ];
#pragma unroll
  for (int j = 0; j < K::NIJ; ++j) {
    const int u = min(tid + K::THREADS * j, K::NIU - 1);
    const int s = u / K::RAW, px = u % K::RAW;
    const int b = min(b0 + s, n - 1);
    isrc[j] = in + (int64_t)b * in_sn + px;  // + (16 cb + c) * RAW
    idst[j] = (s * K::LPP + (px / K::HIN + K::PAD) * K::LP + px % K::HIN + K::PAD) * K::PIXB;
    isc[j] = scales[b];
  }
  const unsigned short* wsrc = wp + (int64_t)rt * ((int64_t)K::NCB * K::KK * NPL * 1024) + tid * 8;  // + kb * TAPC * NPL * 1024 + j * THREADS * 8
  // ---- operand bases
  int aA[2], bB[K::TN];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = K::W_OFF + (i * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    int c = wc * (32 * K::TN) + j * 32 + l31;
    if (c >= K::COLS) c = 0;
    const int s = c / K::P, pix = c % K::P;
    bB[j] = (s * K::LPP + (pix / K::OH) * K::LP + pix % K::OH) * K::PIXB + hi * 16;
  }
  float ir[K::NIJ][K::KOC];
  f4 wr[K::NWJ];
  auto fetch_img = [&](int cb) {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j)
#pragma unroll
      for (int c = 0; c < K::KOC; ++c) ir[j][c] = isrc[j][(int64_t)(cb * K::KOC + c) * K::RAW];
  };
  auto fetch_w = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::NWJ; ++j)
      if (j + 1 < K::NWJ || tid + K::THREADS * j < K::NWQ) wr[j] = *(const f4*)(wsrc + (int64_t)kb * (K::TAPC * NPL * 1024) + j * (K::THREADS * 8));
  };
  auto commit_img = [&]() {
#pragma unroll
    for (int j = 0; j < K::NIJ; ++j) {
      if (j + 1 < K::NIJ || tid + K::THREADS * j < K::NIU) {
        unsigned pl[K::KOC / 2][NPL];
#pragma unroll
        for (int c = 0; c < K::KOC / 2; ++c) split_planes(ir[j][2 * c], ir[j][2 * c + 1], isc[j], pl[c]);
        char* d = ldsp + idst[j];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          *(u4v*)(d + p * K::IMG_PLANE) = (u4v){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
          *(u4v*)(d + p * K::IMG_PLANE + 16) = (u4v){pl[4][p], pl[5][p], pl[6][p], pl[7][p]};
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
  const float winv = 1.0f / whdr[1];
#pragma unroll
  for (int j = 0; j < K::TN; ++j) {
    const int c = wc * (32 * K::TN) + j * 32 + l31;
    const int s = c / K::P, pix = c % K::P;
    if (c >= K::COLS || b0 + s >= n) continue;
    const float inv = winv / lsc[s];
    float* op = out + (int64_t)(b0 + s) * out_sn + (int64_t)(rt * 64) * K::P + pix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + acc_row(r, hi);
        float v = __builtin_fmaf(acc[i][j][r], inv, lbias[row]);
        if (act == 1) v = fmaxf(v, 0.0f);
        op[(int64_t)row * K::P] = v;
      }
  }
}

}  // namespace pconv

// ---- dispatch table (same layers as dconv.hip's forward / data-gradient instantiations) -------------------------------------------
// Tile shapes.  A 5-wave workgroup would fit 25 column tiles exactly, but two of its waves share a SIMD and the compiler then has
// 256 registers per wave for 160 accumulators + fragments + prefetch: four waves x 7 column tiles (224 accumulators of a 512-register
// budget, 25 of 28 tiles used) instead.  The 18-tile geometries run 6 waves x 3 tiles (96 accumulators), two workgroups per CU.
//                        CIN  COUT KS HIN PAD NS WAVES TN TAPC
using PN1dC2F = pconv::Geo<64, 128, 5, 22, 1, 2, 4, 7, 5>;    // 2 x 400 columns = 25 of 28 column tiles
using PN1dC2D = pconv::Geo<128, 64, 5, 20, 3, 1, 4, 4, 5>;    // 484 of 512 columns
using PN1dC3F = pconv::Geo<128, 256, 3, 10, 1, 8, 4, 7, 3>;   // 8 x 100 columns
using PN1dC3D = pconv::Geo<256, 128, 3, 10, 1, 8, 4, 7, 3>;
using PNavC2F = pconv::Geo<64, 128, 3, 24, 1, 1, 6, 3, 3>;    // 576 columns = 18 column tiles
using PNavC2D = pconv::Geo<128, 64, 3, 24, 1, 1, 6, 3, 3>;
using PNavC3F = pconv::Geo<128, 256, 3, 12, 1, 4, 6, 3, 3>;   // 4 x 144 columns
using PNavC3D = pconv::Geo<256, 128, 3, 12, 1, 4, 6, 3, 3>;

enum PlanesId { kPNone = -1, kPN1dC2, kPN1dC3, kPNavC2, kPNavC3 };

static PlanesId planes_id(const ConvGeom& g) {
#ifdef DDRL_PLANES_BF16
  return kPNone;  // the three-plane build keeps the f32-input kernels of dconv.hip
#else
  static const bool off = [] { const char* e = getenv("DDRL_NAV_F32"); return e && e[0] == '1'; }();  // A/B switch: the f32-input kernels
  if (off) return kPNone;
  if (g.stride != 1 || g.h != g.w || g.kh != g.kw || g.pad_h != g.pad_w || g.pad_h != 1) return kPNone;
  const auto is = [&](int cin, int cout, int ks, int h) { return g.cin == cin && g.cout == cout && g.kh == ks && g.h == h; };
  if (is(64, 128, 5, 22)) return kPN1dC2;
  if (is(128, 256, 3, 10)) return kPN1dC3;
  if (is(64, 128, 3, 24)) return kPNavC2;
  if (is(128, 256, 3, 12)) return kPNavC3;
  return kPNone;
#endif
}

bool conv_has_planes(const ConvGeom& g) { return planes_id(g) != kPNone; }

// floats of ONE packed region (forward or data gradient): the planes (2 bytes x 2 planes per weight = 4 bytes) + a 64-float header
int64_t conv_planes_pack_floats(const ConvGeom& g) { return (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2 + 64; }

void launch_conv_planes_pack(const ConvGeom& g, const float* w, float* wpf, float* wpd, hipStream_t st) {
  const int kk = g.kh * g.kw;
  const int64_t total = (int64_t)g.cout * g.cin * kk;
  const int64_t planes = total * NPL / 2;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  for (int dg = 0; dg < 2; ++dg) {
    float* region = dg ? wpd : wpf;
    float* hdr = region + planes;
    (void)hipMemsetAsync(hdr, 0, 64 * sizeof(float), st);
    hipLaunchKernelGGL(pconv::weight_amax_kernel, dim3(64), dim3(256), 0, st, w, total, hdr);
    hipLaunchKernelGGL(pconv::pack_planes_kernel, dim3(blocks), dim3(256), 0, st, w, g.cin, g.cout, kk, dg, (unsigned short*)region, hdr);
  }
}

template <class K>
static void run_planes(const float* in, int64_t in_sn, const float* region, int64_t planes, float* scales, const float* bias, int act,
                       float* out, int64_t out_sn, int n, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)pconv::direct_planes_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(pconv::sample_scale_kernel, dim3((unsigned)n), dim3(256), 0, st, in, in_sn, K::CIN * K::RAW, scales);
  hipLaunchKernelGGL(pconv::direct_planes_kernel<K>, dim3((unsigned)((n + K::NS - 1) / K::NS), K::COUT / 64, 1), dim3(K::THREADS), K::LDS_BYTES, st, in,
                     in_sn, (const unsigned short*)region, region + planes, scales, bias, act, out, out_sn, n);
}

void launch_conv_planes_fwd(const ConvGeom& g, const float* in, const float* wpf, float* scales, const float* bias, int act, float* out,
                            hipStream_t st) {
  const int64_t planes = (int64_t)g.cout * g.cin * g.kh * g.kw * NPL / 2;
  switch (planes_id(g)) {
    case kPN1dC2: run_planes<PN1dC2F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPN1dC3: run_planes<PN1dC3F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPNavC2: run_planes<PNavC2F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
    case kPNavC3: run_planes<PNavC3F>(in, g.in_sn, wpf, planes, scales, bias, act, out, g.out_sn, g.n, st); break;
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
    default: break;
  }
}

}  // namespace ddrl
