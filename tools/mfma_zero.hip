// What an MFMA on (mostly) ZERO operands costs under the socket's power cap: the padded k-groups and zero borders of the plane kernels
//  multiply zeros in one operand.  v_mfma_f32_32x32x16_f16, operands in registers, 2 workgroups x 4 waves per CU;
// every line ~0.3 s.  A = random / 15 of 16 k zero / all zero; B random.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_zero.hip -o /tmp/mfma_zero && /tmp/mfma_zero
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: A random; 1: A has ONE non-zero k of 16 (lane half 0, element 0); 2: A all zero; 3: every third instruction as mode 1
template <int MODE>
__global__ __launch_bounds__(256) void loop(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  h8v a[2], az[2], b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 va = (_Float16)(0.01f * ((lane * 7 + e * 3 + i) % 61) - 0.3f);
      a[i][e] = va;
      az[i][e] = (MODE == 2) ? (_Float16)0.0f : ((lane < 32 && e == 0) ? va : (_Float16)0.0f);
      b[i][e] = (_Float16)(0.02f * ((lane * 5 + e * 11 + i) % 53) - 0.5f);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool z = MODE == 1 || MODE == 2 || (MODE == 3 && s % 3 == 2);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(z ? az[(i + s) & 1] : a[(i + s) & 1], b[(i >> 1) ^ (s & 1)], acc[i], 0, 0, 0);
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
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 2;
  float* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  const int iters = 4000, reps = 12;
  struct { const char* name; void (*k)(float*, int); } rows[] = {{"A random                    ", loop<0>}, {"A: one non-zero k of 16     ", loop<1>},
                                                                  {"A all zero                  ", loop<2>}, {"every third as 'one of 16'  ", loop<3>}};
  printf("%d CUs, 2 workgroups x 4 waves per CU, 24 MFMAs per iteration and wave, %d iterations; ms per launch\n", prop.multiProcessorCount, iters);
  for (int pass = 0; pass < 2; ++pass) {
    printf("pass %d\n", pass);
    double base = 0.0;
    for (auto& r : rows) {
      const double ms = run(r.k, d, blocks, iters, reps);
      if (base == 0.0) base = ms;
      printf("  %s %8.3f ms   %.3f of the random-operand time\n", r.name, ms, ms / base);
    }
  }
  return 0;
}
