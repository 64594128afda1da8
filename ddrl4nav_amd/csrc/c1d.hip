// The laser branch of NavPreNet1D -- Conv1d(1, 32, 5, stride 2) on 960 beams and Conv1d(32, 32, 3, stride 2) on its 478 outputs, both
// un-activated (reference USTC_lab/nn/nav_encoder.py:99-100, 115-121) -- as bandwidth-shaped fp32 vector kernels (round 4).
//
// These layers hold 6 GFLOP per 4,096 samples (8 us of matrix-pipe time) against 375 MB of activations: what bounds them is moving the
// activations once, and the generic gather kernels of gconv.hip (im2col decode per element, 64-row MFMA tiles for 32 output channels)
// ran them at 12-27 TFLOP/s = 5-8x their bandwidth time.  Here a lane owns ONE output position (forward) or one PAIR of input positions
// (data gradient: the stride-2 pattern of a 3-tap kernel is k in {0, 2} for even and k = 1 for odd positions) and keeps all 32 channels of
// it in registers; the weights are read through the scalar cache from layouts the pack kernel transposes for it ([c][k][oc] / [oc][k][c]:
// the 32 values of an inner loop are contiguous, wave-uniform loads), so an inner loop is 32 v_fmac with an SGPR operand and no LDS at
// all (round 4's first attempt kept inputs and weights in LDS: 8-way bank conflicts on the weight reads, slower than the gather
// kernels: profiles/README.md).  The weight gradient tiles (16 oc x 4 c x 3 taps) per wave over chunks of 64 positions, reduces across
// the lanes once per wave and leaves slabs (fixed order: deterministic).  fp32 throughout, one fused multiply-add per term.
#include <cstdlib>

#include "engine2.h"
#include "kernels.h"
#include "ops.h"

namespace ddrl {

namespace c1d {

// wt[c][k][oc] (forward), wd[oc][k][c] (data gradient) from torch's w[oc][c][k]
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int cin, int cout, int kw, float* __restrict__ wt,
                                                   float* __restrict__ wd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * cin * kw) return;
  const int k = i % kw, c = (i / kw) % cin, oc = i / (kw * cin);
  wt[(c * kw + k) * cout + oc] = w[i];
  wd[(oc * kw + k) * cin + c] = w[i];
}

// out[b][oc][x] = act(bias[oc] + sum_{c, k} w[oc][c][k] in[b][c][S x + k]);  grid (ceil(OW / 256), n)
template <int CIN, int COUT, int KW, int S>
__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ in, int64_t in_sn, const float* __restrict__ wt,
                                                  const float* __restrict__ bias, int act, float* __restrict__ out, int64_t out_sn, int W, int OW,
                                                  float* __restrict__ out_amax) {
  const int x0 = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const bool ok = x0 < OW;
  const int x = ok ? x0 : OW - 1;   // lanes past the row compute a duplicate and store nothing (every lane reaches the wave reduction)
  const float* src = in + (int64_t)b * in_sn + S * x;
  float acc[COUT];
#pragma unroll
  for (int oc = 0; oc < COUT; ++oc) acc[oc] = 0.0f;  // the bias joins at the end: terms decades below it must not be rounded against it one by one
#pragma unroll 2
  for (int c = 0; c < CIN; ++c) {
    float a[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) a[k] = src[(int64_t)c * W + k];
#pragma unroll
    for (int k = 0; k < KW; ++k)
#pragma unroll
      for (int oc = 0; oc < COUT; ++oc) acc[oc] = __builtin_fmaf(wt[(c * KW + k) * COUT + oc], a[k], acc[oc]);
  }
  float* dst = out + (int64_t)b * out_sn + x;
  float lm = 0.0f;
#pragma unroll
  for (int oc = 0; oc < COUT; ++oc) {
    float v = acc[oc] + bias[oc];
    if (act == 1) v = fmaxf(v, 0.0f);
    if (ok) dst[(int64_t)oc * OW] = v;
    lm = fmaxf(lm, fabsf(v));
  }
  if (out_amax != nullptr) {   // the sample's largest |output| for the dense layer that reads the flattened rows (engine2.h amax_raise)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lm = fmaxf(lm, __shfl_xor(lm, off, 64));
    if ((threadIdx.x & 63) == 0) amax_raise(lm, out_amax + b);
  }
}

// 3 taps, stride 2:  din[b][c][2 m]     = sum_oc (w[oc][c][0] dz[b][oc][m] + w[oc][c][2] dz[b][oc][m - 1])
//                    din[b][c][2 m + 1] = sum_oc  w[oc][c][1] dz[b][oc][m]            grid (ceil(ceil(W / 2) / 256), n)
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void dgrad32_kernel(const float* __restrict__ dz, int64_t dz_sn, const float* __restrict__ wd,
                                                      float* __restrict__ din, int64_t din_sn, int W, int OW) {
  const int m = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (2 * m >= W) return;
  const float* src = dz + (int64_t)b * dz_sn;
  float ev[CIN], od[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) ev[c] = od[c] = 0.0f;
  const bool ok0 = m < OW, ok1 = m >= 1 && m - 1 < OW;
#pragma unroll 2
  for (int oc = 0; oc < COUT; ++oc) {
    const float d0 = ok0 ? src[(int64_t)oc * OW + m] : 0.0f, d1 = ok1 ? src[(int64_t)oc * OW + m - 1] : 0.0f;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      ev[c] = __builtin_fmaf(wd[(oc * 3 + 0) * CIN + c], d0, ev[c]);
      ev[c] = __builtin_fmaf(wd[(oc * 3 + 2) * CIN + c], d1, ev[c]);
      od[c] = __builtin_fmaf(wd[(oc * 3 + 1) * CIN + c], d0, od[c]);
    }
  }
  float* dst = din + (int64_t)b * din_sn + 2 * m;
  const bool has_odd = 2 * m + 1 < W;
  if (has_odd && !((W | din_sn) & 1) && !((uintptr_t)din & 7)) {  // (even, odd) as one 8-byte store: full lines per wave instead of every other dword
#pragma unroll
    for (int c = 0; c < CIN; ++c) *(float2*)(dst + (int64_t)c * W) = make_float2(ev[c], od[c]);
  } else {
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      dst[(int64_t)c * W] = ev[c];
      if (has_odd) dst[(int64_t)c * W + 1] = od[c];
    }
  }
}

// part[split][oc][c][k] = sum over the split's (b, x) of dz[b][oc][x] in[b][c][2 x + k];  part[split][COUT CIN 3 + oc] = sum dz[b][oc][x].
// A wave owns the tile (16 oc, 4 c, 3 taps) = blockIdx.y and walks chunks of 64 positions (lane = position); grid (splits, tiles).
constexpr int WG_OC = 16, WG_C = 4;
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void wgrad32_kernel(const float* __restrict__ in, int64_t in_sn, const float* __restrict__ dz, int64_t dz_sn,
                                                      float* __restrict__ part, int W, int OW, int n, int nsplit) {
  __shared__ float red[4][WG_OC * WG_C * 3 + WG_OC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NCG = CIN / WG_C;
  const int og = blockIdx.y / NCG, cg = blockIdx.y % NCG, split = blockIdx.x;
  const int cpx = (OW + 63) / 64, nch = n * cpx;  // chunks per sample, chunks in all
  const int per = (nch + nsplit - 1) / nsplit, ch0 = min(nch, split * per), ch1 = min(nch, ch0 + per);
  float acc[WG_OC][WG_C][3], bs[WG_OC];
#pragma unroll
  for (int i = 0; i < WG_OC; ++i) {
    bs[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < WG_C; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[i][j][k] = 0.0f;
  }
  for (int ch = ch0 + wave; ch < ch1; ch += 4) {
    const int b = ch / cpx, x = (ch % cpx) * 64 + lane;
    const bool ok = x < OW;
    const int xs = ok ? x : 0;
    const float* dsrc = dz + (int64_t)b * dz_sn + (int64_t)(og * WG_OC) * OW + xs;
    const float* isrc = in + (int64_t)b * in_sn + (int64_t)(cg * WG_C) * W + 2 * xs;
    float d[WG_OC], a[WG_C][3];
#pragma unroll
    for (int i = 0; i < WG_OC; ++i) d[i] = dsrc[(int64_t)i * OW];
#pragma unroll
    for (int j = 0; j < WG_C; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) a[j][k] = isrc[(int64_t)j * W + k];
#pragma unroll
    for (int i = 0; i < WG_OC; ++i) {
      const float di = ok ? d[i] : 0.0f;
      bs[i] += di;
#pragma unroll
      for (int j = 0; j < WG_C; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[i][j][k] = __builtin_fmaf(di, a[j][k], acc[i][j][k]);
    }
  }
  // across the lanes (fixed order), then across the four waves
  auto lanes = [&](float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
#pragma unroll
  for (int i = 0; i < WG_OC; ++i) {
#pragma unroll
    for (int j = 0; j < WG_C; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float v = lanes(acc[i][j][k]);
        if (lane == 0) red[wave][(i * WG_C + j) * 3 + k] = v;
      }
    const float v = lanes(bs[i]);
    if (lane == 0) red[wave][WG_OC * WG_C * 3 + i] = v;
  }
  __syncthreads();
  float* slab = part + (int64_t)split * ((int64_t)COUT * CIN * 3 + COUT);
  const int t = threadIdx.x;
  if (t < WG_OC * WG_C * 3) {
    const int i = t / (WG_C * 3), j = (t / 3) % WG_C, k = t % 3;
    slab[((og * WG_OC + i) * CIN + cg * WG_C + j) * 3 + k] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  } else if (t < WG_OC * WG_C * 3 + WG_OC && cg == 0) {
    slab[COUT * CIN * 3 + og * WG_OC + (t - WG_OC * WG_C * 3)] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}

}  // namespace c1d

// ---- host side ---------------------------------------------------------------------------------------------------------------------
enum C1dId { kC1None = 0, kC1First, kC1Second };

static C1dId c1d_id(const ConvGeom& g) {
  if (g.h != 1 || g.kh != 1 || g.stride != 2 || g.pad_h != 0 || g.pad_w != 0 || g.cout != 32) return kC1None;
  if (g.cin == 1 && g.kw == 5) return kC1First;
  if (g.cin == 32 && g.kw == 3) return kC1Second;
  return kC1None;
}

bool conv_has_c1d(const ConvGeom& g) { return c1d_id(g) != kC1None; }
bool conv_has_c1d_backward(const ConvGeom& g) { return c1d_id(g) == kC1Second; }  // the first layer keeps thin_wgrad_kernel and has no data gradient
// floats of the two packed regions: wt[c][k][oc], wd[oc][k][c]
int64_t conv_c1d_pack_floats(const ConvGeom& g) { return (int64_t)g.cout * g.cin * g.kw; }

void launch_conv_c1d_pack(const ConvGeom& g, const float* w, float* wt, float* wd, hipStream_t st) {
  const int total = g.cout * g.cin * g.kw;
  hipLaunchKernelGGL(c1d::pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, g.cin, g.cout, g.kw, wt, wd);
}

void launch_conv_c1d_fwd(const ConvGeom& g, const float* in, const float* wt, const float* bias, int act, float* out, float* out_amax, hipStream_t st) {
  const dim3 grid((g.ow + 255) / 256, g.n);
  if (c1d_id(g) == kC1First)
    hipLaunchKernelGGL((c1d::fwd_kernel<1, 32, 5, 2>), grid, dim3(256), 0, st, in, g.in_sn, wt, bias, act, out, g.out_sn, g.w, g.ow, out_amax);
  else
    hipLaunchKernelGGL((c1d::fwd_kernel<32, 32, 3, 2>), grid, dim3(256), 0, st, in, g.in_sn, wt, bias, act, out, g.out_sn, g.w, g.ow, out_amax);
}

void launch_conv_c1d_dgrad(const ConvGeom& g, const float* dz, const float* wd, float* din, hipStream_t st) {
  const dim3 grid(((g.w + 1) / 2 + 255) / 256, g.n);
  hipLaunchKernelGGL((c1d::dgrad32_kernel<32, 32>), grid, dim3(256), 0, st, dz, g.out_sn, wd, din, g.in_sn, g.w, g.ow);
}

int conv_c1d_wgrad_splits(const ConvGeom& g) {
  if (!conv_has_c1d_backward(g)) return 0;
  const int nch = g.n * ((g.ow + 63) / 64);
  int s = 64;                      // x 16 tiles = 1,024 workgroups of four waves
  const int cap = (nch + 15) / 16; // at least four chunks per wave
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

void launch_conv_c1d_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db, hipStream_t st) {
  const int S = conv_c1d_wgrad_splits(g);
  constexpr int TILES = (32 / c1d::WG_OC) * (32 / c1d::WG_C);
  hipLaunchKernelGGL((c1d::wgrad32_kernel<32, 32>), dim3(S, TILES), dim3(256), 0, st, in, g.in_sn, dz, g.out_sn, part, g.w, g.ow, g.n, S);
  const int64_t slab = (int64_t)32 * 32 * 3 + 32;
  launch_reduce_slabs2(part, S, slab, (int64_t)32 * 32 * 3, dw, 32, db, st);
}

}  // namespace ddrl
