// What the 16-bit matrix pipe sustains when the work AROUND it looks like the plane-product kernels of this library: per MFMA
// (v_mfma_f32_32x32x16_f16) V vector-ALU instructions and B bytes streamed from HBM, every operand pair read from LDS.
// Companion of tools/mfma16_peak.hip; 2 workgroups x 4 waves per CU, each line timed for ~0.2 s, two passes.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma16_mix.hip -o /tmp/mfma16_mix && /tmp/mfma16_mix
// The library's kernels (profiles/r03_*_pmc_per_kernel.csv): 2.5-7.6 vector-ALU instructions and 120-150 bytes of HBM traffic per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

// NV vector-ALU instructions per MFMA; LD 16-byte-per-lane loads (1 KB per wave) per iteration of 32 MFMAs, ST the same for stores
template <int NV, int LD, int ST, int OP = 0>
__global__ __launch_bounds__(256) void mix_loop(float* out, const char* src, char* dst, size_t per_wave, int iters, size_t window) {
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
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const char* sp = src + gw * per_wave + lane * 16;
  char* dp = dst + gw * per_wave + lane * 16;
  size_t off = 0;
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
  u4v sink = {1u, 2u, 3u, 4u};
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = f2v{0.001f * (lane + q), 0.002f * (lane + q)};
  for (int it = 0; it < iters; ++it) {
    // the memory traffic of the iteration: requested first, never waited for (the hardware's 64-deep counter is the only back pressure)
#pragma unroll
    for (int l = 0; l < LD; ++l) {
      if (OP >= 100) {  // narrow form: 4 bytes per lane (256 B per instruction), lanes contiguous
        const char* p = src + gw * per_wave + lane * 4 + off + l * 256;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink[0]) : "v"(p) : "memory");
      } else {
        const char* p = sp + off + l * 1024;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(p) : "memory");  // "+": the registers stay reserved while loads are in flight
      }
    }
#pragma unroll
    for (int l = 0; l < ST; ++l) {
      char* p = dp + off + l * 1024;
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(sink) : "memory");
    }
    off += (LD > ST ? LD : ST) * 1024;
    if (off + 8 * 1024 > window) off = 0;  // window = per_wave: streams from HBM; a few KB: the same lines again and again (L2 hits)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      h8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(h8v, lds[abase + i * 64 + s * 128]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(h8v, lds[bbase + j * 256 + (s & 3) * 512]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4 * NV; ++q) {
        if (OP == 0) {
          if (q & 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q & 3]) : "v"(v[q & 3][0]), "v"(v[(q + 1) & 3][1]));
          else asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 3]) : "v"(v[(q + 1) & 3]));
        } else if (OP == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(pk[q & 3]) : "v"(pk[(q + 1) & 3]));
        else if (OP == 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pk[q & 3]) : "v"(pk[(q + 1) & 3]), "v"(pk[(q + 2) & 3]), "v"(pk[(q + 3) & 3]));
        else if (OP == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[q & 3][0]) : "v"(pk[(q + 1) & 3]));
        else if (OP == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(pk[q & 3]) : "v"(v[q & 3][0]), "v"(v[(q + 1) & 3][1]));
        else if (OP == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[q & 3]) : "v"(v[(q + 1) & 3]), "v"(v[(q + 2) & 3]));
        else if (OP == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[q & 3][0]) : "v"(v[(q + 1) & 3][1]), "v"(v[(q + 2) & 3][0]));
        else if (OP == 7) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[q & 3][0]) : "v"(v[(q + 1) & 3][1]), "v"(v[(q + 2) & 3][0]));
        else if (OP == 8) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[q & 3]) : "v"(v[q & 3][0]), "v"(v[(q + 1) & 3][1]));
        else if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 3]) : "v"(v[(q + 1) & 3]));
        else if (OP == 10) asm volatile("v_add_u32 %0, %1, %2" : "=v"(pk[q & 3]) : "v"(pk[(q + 1) & 3]), "v"(pk[(q + 2) & 3]));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
#pragma unroll
  for (int q = 0; q < 4; ++q) s += v[q][0] + v[q][1] + (float)pk[q];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(sink[0] & 1u);
}

typedef void (*kern_t)(float*, const char*, char*, size_t, int, size_t);
static void run(const char* name, kern_t k, int ld, int st, float* d, const char* src, char* dst, size_t per_wave, int blocks, int iters, int reps, size_t window = 0) {
  if (window == 0) window = per_wave;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, src, dst, per_wave, iters, window);
  const hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    printf("  %s: %s\n", name, hipGetErrorString(err));
    exit(1);
  }
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, src, dst, per_wave, iters, window);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4, mfma = waves * (double)iters * 32 * reps;
  const double bytes = waves * (double)iters * reps * (ld + st) * 1024.0;
  printf("  %-44s %8.1f TFLOP/s   %5.2f TB/s   %6.1f B per MFMA   (%.0f ms)\n", name, mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12,
         bytes / (ms * 1e-3) / 1e12, (ld + st) * 1024.0 / 32.0, ms);
  fflush(stdout);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 2;
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  const size_t per_wave = (size_t)1 << 20;  // 1 MB per wave and direction: 2 GB read + 2 GB written, far beyond L2 and the 256 MB memory-side cache
  char *src, *dst;
  if (hipMalloc(&src, per_wave * blocks * 4) != hipSuccess || hipMalloc(&dst, per_wave * blocks * 4) != hipSuccess) return 1;
  hipMemset(src, 1, per_wave * blocks * 4);
  hipMemset(dst, 0, per_wave * blocks * 4);
  printf("%s, %d CUs, 2 workgroups x 4 waves per CU, v_mfma_f32_32x32x16_f16 with both operands from LDS\n", p.name, p.multiProcessorCount);
  const bool quick = argc > 1;  // `mfma16_mix quick`: two lines, once (for a rocprofv3 --pmc pass: does SQ_INSTS_VALU count the MFMAs?)
  const int iters = quick ? 2000 : 12000, reps = quick ? 1 : 6;
#define RUN(NV, LD, ST, label) run(label, mix_loop<NV, LD, ST>, LD, ST, d, src, dst, per_wave, blocks, iters, reps)
#define RUNOP(OP, label) run(label, mix_loop<4, 0, 0, OP>, 0, 0, d, src, dst, per_wave, blocks, iters, reps)
  if (quick) {
    RUN(0, 0, 0, "MFMA + LDS operands only");
    RUN(3, 0, 0, "+ 3 VALU per MFMA");
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d\n", pass);
    RUN(0, 0, 0, "MFMA + LDS operands only");
    RUN(0, 2, 0, "+ 64 B read per MFMA");
    RUN(0, 4, 0, "+ 128 B read per MFMA");
    RUN(0, 2, 2, "+ 64 B read + 64 B written per MFMA");
    RUN(0, 4, 4, "+ 128 B read + 128 B written per MFMA");
    RUN(3, 0, 0, "+ 3 VALU per MFMA");
    RUN(3, 2, 2, "+ 3 VALU, 64 B read + 64 B written per MFMA");
    RUN(3, 4, 0, "+ 3 VALU, 128 B read per MFMA");
    RUN(6, 2, 2, "+ 6 VALU, 64 B read + 64 B written per MFMA");
    RUN(6, 4, 0, "+ 6 VALU, 128 B read per MFMA");
    // the same reads served by L2 (every wave cycles through a 16 KB window: 32 MB in all) -- what an L2 -> LDS / register weight copy costs
    run("+ 128 B read per MFMA from L2", mix_loop<0, 4, 0>, 4, 0, d, src, dst, per_wave, blocks, iters, reps, 16384);
    run("+ 256 B read per MFMA from L2", mix_loop<0, 8, 0>, 8, 0, d, src, dst, per_wave, blocks, iters, reps, 16384);
    run("+ 3 VALU, 128 B read per MFMA from L2", mix_loop<3, 4, 0>, 4, 0, d, src, dst, per_wave, blocks, iters, reps, 16384);
    // 4-byte-per-lane loads: the same number of INSTRUCTIONS as the 128 B line above moves a quarter of the bytes
    run("+ 4 dword loads per 32 MFMAs (32 B / MFMA)", mix_loop<0, 4, 0, 100>, 1, 0, d, src, dst, per_wave, blocks, iters, reps);
    run("+ 16 dword loads per 32 MFMAs (128 B / MFMA)", mix_loop<0, 16, 0, 100>, 4, 0, d, src, dst, per_wave, blocks, iters, reps);
    // which vector-ALU instructions cost what: 4 of one kind per MFMA, no memory traffic
    RUNOP(1, "+ 4 v_mov_b32 per MFMA");
    RUNOP(10, "+ 4 v_add_u32 per MFMA");
    RUNOP(2, "+ 4 v_perm_b32 per MFMA");
    RUNOP(3, "+ 4 v_cvt_f32_ubyte1 per MFMA");
    RUNOP(7, "+ 4 v_max_f32 per MFMA");
    RUNOP(6, "+ 4 v_fma_f32 per MFMA");
    RUNOP(8, "+ 4 v_cvt_pk_f16_f32 per MFMA");
    RUNOP(4, "+ 4 v_fma_mixlo_f16 per MFMA");
    RUNOP(5, "+ 4 v_pk_mul_f32 per MFMA");
    RUNOP(9, "+ 4 v_pk_fma_f32 per MFMA");
  }
  return 0;
}
