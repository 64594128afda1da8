// What a HALF-DEPTH matrix instruction costs on gfx950: v_mfma_f32_32x32x8_f16 (the CDNA1-3 form, k = 8) against
// v_mfma_f32_32x32x16_f16 (k = 16), operands in registers, 2 workgroups x 4 waves per CU, each line ~0.3 s back to back so that the
// socket's power limit (not the issue rate alone) sets the clock.  Question behind it (round 6): the plane kernels pad their reduction
// index to multiples of 16 (conv1's weight gradient 40 -> 48 pixels, conv3's forward 9 -> 10 taps, ...); if the k = 8 form costs half
// of a k = 16 one, a ragged tail of <= 8 can run on it.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_k8.hip -o /tmp/mfma_k8 && /tmp/mfma_k8
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// per iteration: N16 k = 16 instructions and N8 k = 8 instructions on each of 4 accumulators
template <int N16, int N8>
__global__ __launch_bounds__(256) void loop(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  h8v a[2], b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[i][e] = (_Float16)(0.01f * ((lane * 7 + e * 3 + i) % 61) - 0.3f);
      b[i][e] = (_Float16)(0.02f * ((lane * 5 + e * 11 + i) % 53) - 0.5f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < N16; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + s) & 1], b[(i >> 1) ^ (s & 1)], acc[i], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < N8; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const h8v af = a[(i + s) & 1], bf = b[(i >> 1) ^ (s & 1)];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16((h4v){af[0], af[1], af[2], af[3]}, (h4v){bf[4], bf[5], bf[6], bf[7]}, acc[i], 0, 0, 0);
      }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
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
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, blocks = cus * 2;
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  const int iters = 4000, reps = 12;
  struct { const char* name; void (*k)(float*, int); int n16, n8; } rows[] = {
      {"6 x k16            ", loop<6, 0>, 6, 0}, {"12 x k8            ", loop<0, 12>, 0, 12}, {"5 x k16 + 1 x k8   ", loop<5, 1>, 5, 1},
      {"5 x k16            ", loop<5, 0>, 5, 0}, {"2 x k16 + 1 x k8   ", loop<2, 1>, 2, 1},  {"3 x k16            ", loop<3, 0>, 3, 0},
      {"2 x k16            ", loop<2, 0>, 2, 0}};
  printf("%s, %d CUs, 2 workgroups x 4 waves per CU, operands in registers; ms per launch of %d iterations\n", prop.name, cus, iters);
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d\n", pass);
    for (auto& r : rows) {
      const double ms = run(r.k, d, blocks, iters, reps);
      const double insts = (double)blocks * 4 * iters * 4 * (r.n16 + r.n8);            // wave instructions
      const double flop = (double)blocks * 4 * iters * 4 * (r.n16 * 32768.0 + r.n8 * 16384.0);
      const double simd_cycles_at_2p4 = ms * 1e-3 * 2.4e9 / ((double)iters * 4 * (r.n16 + r.n8) * 2);  // two waves share a SIMD
      printf("  %s %8.3f ms   %7.1f TFLOP/s   %.2f ns per wave-instruction per SIMD (= %.1f cycles at 2.4 GHz)\n", r.name, ms, flop / ms * 1e-9,
             ms * 1e6 / ((double)iters * 4 * (r.n16 + r.n8) * 2), simd_cycles_at_2p4);
      (void)insts;
    }
  }
  return 0;
}
