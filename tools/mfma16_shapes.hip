// v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 under the socket's power limit (MI355X_MICROARCH.md "DVFS give-back" item 7:
// the 16x16x32 shape held a higher clock and delivered 1.12-1.15 x the FLOP/s in bare bf16 loops).  Same operand traffic per FLOP in
// both forms: a wave computes a 64 x 64 output block per k = 32 -- 2 x 2 tiles of 32 x 32 (two k-steps of 16: 8 fragment reads, 8 MFMAs)
// or 4 x 4 tiles of 16 x 16 (one k-step of 32: 8 fragment reads, 16 MFMAs) -- 64 accumulator registers either way.
//   regs : operands in registers        lds : every fragment a ds_read_b128 at lane_base + immediate
//   +V   : V vector-ALU instructions per 32,768 FLOP-pairs (= per 32x32x16 MFMA, per two 16x16x32), the plane split's mix
// hipcc -O3 --offload-arch=gfx950 tools/mfma16_shapes.hip -o /tmp/mfma16_shapes && /tmp/mfma16_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;
typedef __attribute__((ext_vector_type(2))) float f2v;

__device__ __forceinline__ void fill_lds(u4v* lds, int n) {
  for (int i = threadIdx.x; i < n; i += 256) {
    unsigned s = i * 2654435761u + blockIdx.x;
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
}

template <int NV>
__device__ __forceinline__ void valu(f2v (&v)[4], unsigned (&pk)[4]) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (q & 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q & 3]) : "v"(v[q & 3][0]), "v"(v[(q + 1) & 3][1]));
    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 3][0]) : "v"(v[(q + 1) & 3][1]));
  }
}

// ---- 32 x 32 x 16: per k = 32: 2 k-steps x (2 A + 2 B fragment reads, 4 MFMAs)
template <int NV, bool LDS>
__global__ __launch_bounds__(256) void loop32(float* out, int iters) {
  __shared__ u4v lds[4096];
  fill_lds(lds, 4096);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = lane, bbase = 2048 + wave * 64 + lane;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  f2v v[4];
  unsigned pk[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = f2v{0.001f * (lane + q), 0.002f * (lane + q)};
  h8v ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) ra[i] = __builtin_bit_cast(h8v, lds[abase + i * 64]), rb[i] = __builtin_bit_cast(h8v, lds[bbase + i * 256]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {  // 8 k-steps of 16 = 4 x (k = 32)
      h8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = LDS ? __builtin_bit_cast(h8v, lds[abase + i * 64 + s * 128]) : ra[(i + s) & 1];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = LDS ? __builtin_bit_cast(h8v, lds[bbase + j * 256 + (s & 3) * 512]) : rb[(j + s) & 1];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
          valu<NV>(v, pk);
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

// ---- 16 x 16 x 32: per k = 32: 4 A + 4 B fragment reads, 16 MFMAs
template <int NV, bool LDS>
__global__ __launch_bounds__(256) void loop16(float* out, int iters) {
  __shared__ u4v lds[4096];
  fill_lds(lds, 4096);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = lane, bbase = 2048 + wave * 64 + lane;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;
  f2v v[4];
  unsigned pk[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = f2v{0.001f * (lane + q), 0.002f * (lane + q)};
  h8v ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(h8v, lds[abase + i * 64]), rb[i] = __builtin_bit_cast(h8v, lds[bbase + i * 256]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // 4 x (k = 32)
      h8v a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = LDS ? __builtin_bit_cast(h8v, lds[abase + i * 64 + s * 256]) : ra[(i + s) & 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = LDS ? __builtin_bit_cast(h8v, lds[bbase + j * 256 + (s & 1) * 1024]) : rb[(j + s) & 3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
          if ((i * 4 + j) & 1) valu<NV>(v, pk);   // NV per TWO of these = per 32x32x16's worth of FLOP
        }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
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
  // both loops: 32 x (32x32x16 worth of FLOP) per wave per iteration = 4 x (k = 32) x 64 x 64 outputs
  const double flop = (double)blocks * 4 * (double)iters * reps * 4.0 * 2.0 * 64 * 64 * 32;
  return flop / (ms * 1e-3) / 1e12;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2;
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  printf("%s, %d CUs, 2 workgroups x 4 waves per CU; TFLOP/s of f16 MFMA work (dense peak 2500)\n", p.name, p.multiProcessorCount);
  const int iters = 20000, reps = 8;
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d                     32x32x16   16x16x32   ratio\n", pass);
#define LINE(name, NV, L)                                                             \
  {                                                                                   \
    const double x = run(loop32<NV, L>, d, blocks, iters, reps), y = run(loop16<NV, L>, d, blocks, iters, reps); \
    printf("  %-22s %8.1f   %8.1f   %.3f\n", name, x, y, y / x);                      \
  }
    LINE("regs", 0, false)
    LINE("lds", 0, true)
    LINE("lds + 2 valu", 2, true)
    LINE("lds + 4 valu", 4, true)
    LINE("lds + 6 valu", 6, true)
  }
  hipFree(d);
  return 0;
}
