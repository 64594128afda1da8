// Bit-identity check of the lane-exchange reductions of csrc/ppo_math.h against the ds_bpermute (__shfl_xor) butterfly they replace:
//   (1) wave_sum / wave_max in every lane;  (2) the 32-value transposing reduction + the DPP row gather heads_loss uses (value i + 4 j of
//   "sample" i lands in lane 32 (i & 1) + 16 (i >> 1) as g[j]), against 32 separate butterflies.
// Distinct random values in every lane and register, so that a wrong exchange partner cannot cancel out.
//   hipcc -O3 --offload-arch=gfx950 -Iddrl4nav_amd/csrc tools/wave_butterfly.hip -o /tmp/wb && /tmp/wb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ppo_math.h"

using namespace ddrl;

__device__ __forceinline__ float shfl_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float shfl_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// x: [wave][32 values][64 lanes]; ref / got: [wave][32][64] (sum per value in every lane / gathered totals, see below)
__global__ void check_kernel(const float* x, float* ref, float* got, float* ref_max, float* got_max, float* got_sum) {
  const int lane = threadIdx.x, wv = blockIdx.x;
  const float* xs = x + (size_t)wv * 32 * 64;
  float v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = xs[k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 32; ++k) ref[((size_t)wv * 32 + k) * 64 + lane] = shfl_sum(v[k]);
  ref_max[(size_t)wv * 64 + lane] = shfl_max(fabsf(v[0]));
  got_max[(size_t)wv * 64 + lane] = wave_max(fabsf(v[0]));
  got_sum[(size_t)wv * 64 + lane] = wave_sum(v[0]);
  // value index k = i + 4 j, i = i0 + 2 i1
  const bool bit3 = lane & 8, bit2 = lane & 4, bit1 = lane & 2;
  float zz[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float y[2];
#pragma unroll
    for (int i1 = 0; i1 < 2; ++i1) y[i1] = swap_add32(v[2 * i1 + 4 * j], v[2 * i1 + 1 + 4 * j]);
    zz[j] = swap_add16(y[0], y[1]);
  }
  float w4[4], w2[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) w4[q] = fold_add8(zz[2 * q], zz[2 * q + 1], bit3);
#pragma unroll
  for (int q = 0; q < 2; ++q) w2[q] = fold_add4(w4[2 * q], w4[2 * q + 1], bit2);
  float tot = fold_add2(w2[0], w2[1], bit1);
  tot += dpp_lane<0xB1>(tot, tot);
  float g[8];
  g[0] = tot;
  g[1] = dpp_lane<0x108>(tot, tot), g[2] = dpp_lane<0x104>(tot, tot), g[3] = dpp_lane<0x10C>(tot, tot);
  g[4] = dpp_lane<0x102>(tot, tot), g[5] = dpp_lane<0x10A>(tot, tot), g[6] = dpp_lane<0x106>(tot, tot);
  g[7] = dpp_lane<0x10E>(tot, tot);
  // lane l stores what it gathered as "sample" i(l) = bit5 + 2 bit4; only the rows' first lanes are meaningful
  const int i = (lane >> 5) | ((lane >> 3) & 2);
#pragma unroll
  for (int j = 0; j < 8; ++j) got[((size_t)wv * 32 + i + 4 * j) * 64 + lane] = g[j];
}

int main() {
  const int waves = 512, n = waves * 32 * 64;
  float *x, *ref, *got, *rm, *gm, *gs;
  float* h = (float*)malloc(n * 4);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 20)) * ((i % 7) == 0 ? 1e-3f : 1.0f);
  }
  if (hipMalloc(&x, n * 4) || hipMalloc(&ref, n * 4) || hipMalloc(&got, n * 4) || hipMalloc(&rm, waves * 64 * 4) ||
      hipMalloc(&gm, waves * 64 * 4) || hipMalloc(&gs, waves * 64 * 4))
    return 2;
  if (hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice) || hipMemset(got, 0, n * 4)) return 2;
  check_kernel<<<waves, 64>>>(x, ref, got, rm, gm, gs);
  float *hr = (float*)malloc(n * 4), *hg = (float*)malloc(n * 4), *a = (float*)malloc(waves * 256), *b = (float*)malloc(waves * 256),
        *c = (float*)malloc(waves * 256);
  if (hipMemcpy(hr, ref, n * 4, hipMemcpyDeviceToHost) || hipMemcpy(hg, got, n * 4, hipMemcpyDeviceToHost) ||
      hipMemcpy(a, rm, waves * 256, hipMemcpyDeviceToHost) || hipMemcpy(b, gm, waves * 256, hipMemcpyDeviceToHost) ||
      hipMemcpy(c, gs, waves * 256, hipMemcpyDeviceToHost))
    return 2;
  int bad_sum = 0, bad_max = 0, bad_tr = 0;
  for (int w = 0; w < waves; ++w) {
    for (int l = 0; l < 64; ++l) {
      if (memcmp(&a[w * 64 + l], &b[w * 64 + l], 4)) ++bad_max;
      if (memcmp(&hr[(w * 32 + 0) * 64 + l], &c[w * 64 + l], 4)) ++bad_sum;
    }
    for (int k = 0; k < 32; ++k) {
      const int i = k & 3, owner = 32 * (i & 1) + 16 * (i >> 1);
      if (memcmp(&hr[(w * 32 + k) * 64 + owner], &hg[(w * 32 + k) * 64 + owner], 4)) ++bad_tr;
    }
  }
  printf("wave_sum mismatching lanes %d, wave_max %d of %d; transposing reduction mismatching values %d of %d\n", bad_sum, bad_max,
         waves * 64, bad_tr, waves * 32);
  return (bad_sum || bad_max || bad_tr) ? 1 : 0;
}
