// Sustained f32-input MFMA rate of this GPU: a register-only loop of independent
// v_mfma_f32_32x32x2_f32 (no LDS, no memory), 2 waves per SIMD on every CU, timed in windows.
// Companion to DESIGN.md section 3.2 ("busy x clock"): what the part sustains when nothing but the
// matrix pipe is working.   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  // mode 0: one constant operand pair (little bit toggling); mode 1: eight pseudo-random operand pairs per
  // lane, rotated every MFMA (data-dependent switching as in a real GEMM)
  float a[8], b[8];
  unsigned s0 = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    s0 = s0 * 1664525u + 1013904223u;
    a[u] = a0 == 0.0f ? ((float)(s0 >> 8) / 16777216.0f - 0.5f) : a0;
    s0 = s0 * 1664525u + 1013904223u;
    b[u] = a0 == 0.0f ? ((float)(s0 >> 8) / 16777216.0f - 0.5f) * 1e-3f : b0;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 2 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 2: the operand pattern of the library's kernels -- per k-step 2 + 2 ds_read_b32 at lane_base + immediate
// feeding a 2 x 2 block of MFMAs, 16 k-steps per "k-block", no global memory, no barriers
__global__ __launch_bounds__(256) void mfma_lds_loop(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = ((float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f) * 0.05f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = (lane >> 5) * 64 + (lane & 31), bbase = 4096 + (lane >> 5) * 260 + wave * 64 + (lane & 31);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = lds[abase + i * 32 + 2 * s * 128];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds[bbase + j * 32 + 2 * s * 4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 3: mode 2 plus NV independent VALU fmas per k-step (4 MFMAs): does vector-ALU work overlap the matrix pipe?
template <int NV>
__global__ __launch_bounds__(256) void mfma_valu_loop(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = ((float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f) * 0.05f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = (lane >> 5) * 64 + (lane & 31), bbase = 4096 + (lane >> 5) * 260 + wave * 64 + (lane & 31);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.001f * (lane + q);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = lds[abase + i * 32 + 2 * s * 128];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds[bbase + j * 32 + 2 * s * 4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
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
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 4: mode 2 plus NLD independent 1 KB global loads per 32 MFMAs and wave (streaming, never waited for
// inside the loop, scalar-base addressing so that no VALU work is added): does HBM traffic by itself slow the
// matrix pipe (power / clock, shared paths)?
template <int NLD>
__global__ __launch_bounds__(256) void mfma_stream_loop(float* out, const char* src, unsigned long long src_bytes, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = ((float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f) * 0.05f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int abase = (lane >> 5) * 64 + (lane & 31), bbase = 4096 + (lane >> 5) * 260 + wave * 64 + (lane & 31);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const unsigned voff = lane * 16;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * 4;
  const unsigned long long wid = (unsigned long long)blockIdx.x * 4 + (unsigned)__builtin_amdgcn_readfirstlane(wave);
  unsigned long long pos = wid * 1024;  // wave-uniform (blockIdx / wave id only)
  // "+v": the destination stays allocated to the asm statements for the whole loop -- the loads complete
  // asynchronously, so their register must never be handed to anything else
  f4v sink = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
      const unsigned long long ab = (unsigned long long)(src + pos);
      const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab >> 32));
      const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ab);  // (unsigned: no sign extension below)
      const unsigned long long sb = ((unsigned long long)hi32 << 32) | (unsigned long long)lo32;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(sink) : "v"(voff), "s"(sb) : "memory");
      pos += nwaves * 1024;
      if (pos + 1024 > src_bytes) pos = wid * 1024;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = lds[abase + i * 32 + 2 * s * 128];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds[bbase + j * 32 + 2 * s * 4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (NLD > 0) s += sink[0] * 0.0f;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NLD>
static void run_stream(float* out, const char* src, unsigned long long src_bytes, int wgs, int iters, double flop, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int w = 0; w < 6; ++w) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_stream_loop<NLD>, dim3(wgs), dim3(256), 0, 0, out, src, src_bytes, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (w >= 2 && ms < best) best = ms;
  }
  const double bytes = (double)wgs * 4 * iters * NLD * 1024.0;
  printf("%d x 1 KB global load per 32 MFMAs: %8.2f ms  %7.1f TFLOP/s   %6.2f TB/s streamed\n", NLD, best, flop / (best * 1e-3) / 1e12,
         bytes / (best * 1e-3) / 1e12);
}

template <int NV>
static void run_valu(float* out, int wgs, int iters, double flop, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int w = 0; w < 6; ++w) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_valu_loop<NV>, dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (w >= 2 && ms < best) best = ms;
  }
  printf("%3d VALU fma per 4 MFMAs: %8.2f ms  %7.1f TFLOP/s   (fully serialised would be %5.1f)\n", NV, best, flop / (best * 1e-3) / 1e12,
         153.7 * 256.0 / (256.0 + 4.0 * NV));
}

int main() {
  const int wgs = 256 * 2, iters = 20000;  // 2 workgroups of 4 waves per CU; 32 MFMAs per iteration per wave
  float* out;
  if (hipMalloc((void**)&out, wgs * 256 * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double flop_per_launch = (double)wgs * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);
  hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, 1000, 1.0f, 1e-3f);
  hipDeviceSynchronize();
  for (int mode = 0; mode < 2; ++mode) {
  printf("%s operands\nwindow  ms        TFLOP/s\n", mode ? "pseudo-random" : "constant");
  for (int w = 0; w < 40; ++w) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, iters, mode ? 0.0f : 1.0f, 1e-3f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (w < 3 || w % 10 == 9) printf("%3d   %8.2f   %7.1f\n", w, ms, flop_per_launch / (ms * 1e-3) / 1e12);
  }
  }
  printf("operands from LDS (2 + 2 ds_read_b32 per 4 MFMAs)\nwindow  ms        TFLOP/s\n");
  for (int w = 0; w < 40; ++w) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_lds_loop, dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (w < 3 || w % 10 == 9) printf("%3d   %8.2f   %7.1f\n", w, ms, flop_per_launch / (ms * 1e-3) / 1e12);
  }
  {
    const unsigned long long src_bytes = 4ull << 30;
    char* src = nullptr;
    if (hipMalloc((void**)&src, src_bytes) == hipSuccess) {
      hipMemset(src, 0, src_bytes);
      run_stream<0>(out, src, src_bytes, wgs, iters, flop_per_launch, e0, e1);
      run_stream<1>(out, src, src_bytes, wgs, iters, flop_per_launch, e0, e1);
      run_stream<2>(out, src, src_bytes, wgs, iters, flop_per_launch, e0, e1);
      run_stream<4>(out, src, src_bytes, wgs, iters, flop_per_launch, e0, e1);
      hipFree(src);
    }
  }
  run_valu<0>(out, wgs, iters, flop_per_launch, e0, e1);
  run_valu<4>(out, wgs, iters, flop_per_launch, e0, e1);
  run_valu<8>(out, wgs, iters, flop_per_launch, e0, e1);
  run_valu<16>(out, wgs, iters, flop_per_launch, e0, e1);
  run_valu<32>(out, wgs, iters, flop_per_launch, e0, e1);
  return 0;
}
