// What the 16-bit matrix pipe of this GPU sustains under the socket's power limit, and what vector-ALU / LDS work next to
// it costs: loops of independent v_mfma_f32_32x32x16_f16 (2 workgroups x 4 waves per CU on all CUs), each timed for
// several hundred ms.  Companion of tools/mfma_peak.hip (f32-input MFMA) for the plane-product kernels (engine2.h).
//   regs      : operands in registers (8 rotating pseudo-random fragments per lane)
//   lds       : every MFMA operand pair read from LDS (2 x 2 tiles per k-step: 4 ds_read_b128 per 4 MFMAs)
//   lds+valu V: plus V vector-ALU instructions per MFMA (v_pk_fma_f32 / v_cvt_pk_f16_f32 alternating: the split's mix)
// hipcc -O3 --offload-arch=gfx950 tools/mfma16_peak.hip -o /tmp/mfma16_peak && /tmp/mfma16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

__global__ __launch_bounds__(256) void regs_loop(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  h8v a[8], b[8];
  unsigned s0 = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s0 = s0 * 1664525u + 1013904223u;
      a[u][e] = (_Float16)((float)(s0 >> 8) / 16777216.0f - 0.5f);
      s0 = s0 * 1664525u + 1013904223u;
      b[u][e] = (_Float16)(((float)(s0 >> 8) / 16777216.0f - 0.5f) * 1e-2f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + i) & 7], b[(u + 2 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 8 k-steps per iteration; per k-step 2 A fragments + 2 B fragments (ds_read_b128 at lane_base + immediate) -> 2 x 2 MFMAs
template <int NV>
__global__ __launch_bounds__(256) void lds_loop(float* out, int iters) {
  __shared__ u4v lds[4096];  // 64 KB
  for (int i = threadIdx.x; i < 4096; i += 256) {
    unsigned s = i * 2654435761u;
    u4v v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s = s * 1664525u + 1013904223u;
      const _Float16 lo = (_Float16)(((float)(s >> 8) / 16777216.0f - 0.5f) * 0.05f);
      s = s * 1664525u + 1013904223u;
      const _Float16 hi = (_Float16)(((float)(s >> 8) / 16777216.0f - 0.5f) * 0.05f);
      v[e] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    lds[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = lane, bbase = 2048 + wave * 64 + lane;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  typedef __attribute__((ext_vector_type(2))) float f2v;
  f2v v[4];
  unsigned pk[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = f2v{0.001f * (lane + q), 0.002f * (lane + q)};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      h8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(h8v, lds[abase + i * 64 + s * 128]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(h8v, lds[bbase + j * 256 + (s & 3) * 512 - (s >> 2) * 0]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4 * NV; ++q) {
        if (q & 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q & 3]) : "v"(v[q & 3][0]), "v"(v[(q + 1) & 3][1]));
        else asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 3]) : "v"(v[(q + 1) & 3]));
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += v[q][0] + v[q][1] + (float)pk[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double run(void (*k)(float*, int), float* d, int blocks, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * (double)iters * 32 * reps;  // 32 MFMAs per wave per iteration
  return mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2;
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  printf("%s, %d CUs, 2 workgroups x 4 waves per CU; dense f16 MFMA peak of the guide: 2500 TFLOP/s at 2.4 GHz\n", p.name, p.multiProcessorCount);
  const int iters = 20000, reps = 8;  // ~0.1-0.3 s per line
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d\n", pass);
    printf("  regs                 %8.1f TFLOP/s\n", run(regs_loop, d, blocks, iters, reps));
    printf("  lds                  %8.1f TFLOP/s\n", run(lds_loop<0>, d, blocks, iters, reps));
    printf("  lds + 1 valu / mfma  %8.1f TFLOP/s\n", run(lds_loop<1>, d, blocks, iters, reps));
    printf("  lds + 2 valu / mfma  %8.1f TFLOP/s\n", run(lds_loop<2>, d, blocks, iters, reps));
    printf("  lds + 4 valu / mfma  %8.1f TFLOP/s\n", run(lds_loop<4>, d, blocks, iters, reps));
    printf("  lds + 8 valu / mfma  %8.1f TFLOP/s\n", run(lds_loop<8>, d, blocks, iters, reps));
  }
  hipFree(d);
  return 0;
}
