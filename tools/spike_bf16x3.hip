// Research spike (NOT part of libddrl_hip.so): an fp32-accurate GEMM on the bf16 matrix rate.
//
//   C[M][N] = A[M][K] * B[K][N]   with every fp32 operand split exactly into three bf16 terms
//   x = hi + mid + lo, and six of the nine cross products kept:
//   hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   (dropped terms <= 2^-24 relative),
//   each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Purpose: measure what DESIGN.md section 5.1 calls the road past the f32-MFMA ceiling -- rate
// (TFLOP/s of fp32-equivalent work) and error against a float64 reference, next to the error of a
// plain fp32 fma chain -- on the shape of the encoder's dense layer (M = 65,536 samples, K = 3,136,
// N = 512).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/spike_bf16x3.hip -o /tmp/spike && /tmp/spike
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f4;

#define HIPCHECK(x)                                                   \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

__device__ __forceinline__ uint16_t bf16_rn(float x) {  // round to nearest even, finite inputs
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// dst[term][r][c] (bf16) from src[r][c] (transpose = 0) or src[c][r] (transpose = 1); rows x cols = dst shape
__global__ void split3_kernel(const float* __restrict__ src, int rows, int cols, int transpose, uint16_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  const float x = transpose ? src[(int64_t)c * rows + r] : src[i];
  const uint16_t hi = bf16_rn(x);
  const float r1 = x - bf16_to_f(hi);
  const uint16_t mid = bf16_rn(r1);
  const float r2 = r1 - bf16_to_f(mid);
  const uint16_t lo = bf16_rn(r2);
  const int64_t plane = (int64_t)rows * cols;
  dst[i] = hi;
  dst[plane + i] = mid;
  dst[2 * plane + i] = lo;
}

// ---- the GEMM: 128 x 128 tile per workgroup, 4 waves of 64 x 64, k-block 32 -----------------------
constexpr int TILE = 128, KB = 32;
constexpr int ROWB = 80;                       // bytes per LDS row (32 bf16 = 64 B, padded: conflict-free b128 reads)
constexpr int TERM_BYTES = TILE * ROWB;        // one term of one operand
constexpr int OPND_BYTES = 3 * TERM_BYTES;     // A or B
constexpr int STAGE_BYTES = 2 * OPND_BYTES;    // 61,440

template <int NPROD>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(const uint16_t* __restrict__ A3, const uint16_t* __restrict__ B3,
                                                          int M, int N, int K, float* __restrict__ C) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5, wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * TILE, n0 = blockIdx.x * TILE;
  const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
  // staging: per operand and term 128 rows x 64 B = 512 f4; thread -> (row = idx / 4, quarter = idx % 4), 2 per thread
  f4 ra[3][2], rb[3][2];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j, row = idx >> 2, q = idx & 3;
        ra[t][j] = *(const f4*)(A3 + t * planeA + (int64_t)(m0 + row) * K + kb * KB + q * 8);
        rb[t][j] = *(const f4*)(B3 + t * planeB + (int64_t)(n0 + row) * K + kb * KB + q * 8);
      }
  };
  auto commit = [&](char* buf) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 256 * j, row = idx >> 2, q = idx & 3;
        *(f4*)(buf + t * TERM_BYTES + row * ROWB + q * 16) = ra[t][j];
        *(f4*)(buf + OPND_BYTES + t * TERM_BYTES + row * ROWB + q * 16) = rb[t][j];
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int nkb = K / KB;
  fetch(0);
  commit(lds);
  if (nkb > 1) fetch(1);
  __syncthreads();
  int buf = 0;
  for (int kb = 0; kb < nkb; ++kb) {
    const char* cur = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i][t] = *(const bf16x8*)(cur + t * TERM_BYTES + (wr * 64 + i * 32 + l31) * ROWB + kh * 32 + hi * 16);
          b[i][t] = *(const bf16x8*)(cur + OPND_BYTES + t * TERM_BYTES + (wc * 64 + i * 32 + l31) * ROWB + kh * 32 + hi * 16);
        }
      // small products first, the leading one last
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int p = 6 - NPROD; p < 6; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[p]], b[j][TB[p]], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nkb) {
      commit(lds + (buf ^ 1) * STAGE_BYTES);
      if (kb + 2 < nkb) fetch(kb + 2);
    }
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        C[(int64_t)m * N + n] = acc[i][j][r];
      }
  }
}

// references on a sample of outputs: float64 and a plain fp32 fma chain
__global__ void ref_kernel(const float* __restrict__ A, const float* __restrict__ B, int N, int K, const int* __restrict__ ms,
                           const int* __restrict__ ns, int count, double* __restrict__ ref64, float* __restrict__ ref32) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int m = ms[i], n = ns[i];
  double s = 0.0;
  float f = 0.0f;
  for (int k = 0; k < K; ++k) {
    const float a = A[(int64_t)m * K + k], b = B[(int64_t)k * N + n];
    s += (double)a * (double)b;
    f = __builtin_fmaf(a, b, f);
  }
  ref64[i] = s;
  ref32[i] = f;
}

template <int NPROD>
static float run(const uint16_t* A3, const uint16_t* B3, int M, int N, int K, float* C, int reps) {
  auto kern = gemm_bf16x3_kernel<NPROD>;
  HIPCHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
  dim3 grid(N / TILE, M / TILE);
  hipLaunchKernelGGL(kern, grid, dim3(256), 2 * STAGE_BYTES, 0, A3, B3, M, N, K, C);
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  HIPCHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), 2 * STAGE_BYTES, 0, A3, B3, M, N, K, C);
  HIPCHECK(hipEventRecord(e1, 0));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int M = 65536, K = 3136, N = 512, S = 4096;
  float *A, *B, *C, *ref32;
  double* ref64;
  uint16_t *A3, *B3;
  int *ms, *ns;
  HIPCHECK(hipMalloc((void**)&A, (size_t)M * K * 4));
  HIPCHECK(hipMalloc((void**)&B, (size_t)K * N * 4));
  HIPCHECK(hipMalloc((void**)&C, (size_t)M * N * 4));
  HIPCHECK(hipMalloc((void**)&A3, (size_t)3 * M * K * 2));
  HIPCHECK(hipMalloc((void**)&B3, (size_t)3 * N * K * 2));
  HIPCHECK(hipMalloc((void**)&ms, S * 4));
  HIPCHECK(hipMalloc((void**)&ns, S * 4));
  HIPCHECK(hipMalloc((void**)&ref64, S * 8));
  HIPCHECK(hipMalloc((void**)&ref32, S * 4));
  // activations-like A (leaky-ReLU outputs: mostly positive, a few small negatives), weights-like B
  std::vector<float> hB((size_t)K * N), hrow(K);
  uint32_t s = 777u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) / 16777216.0f;
  };
  for (auto& x : hB) x = (rnd() * 2.f - 1.f) * 0.0179f;
  HIPCHECK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hA((size_t)1024 * K);
  for (auto& x : hA) {
    const float u = rnd() * 2.f - 1.f;
    x = u > 0.f ? u : 0.01f * u;
  }
  for (int blk = 0; blk < M / 1024; ++blk)  // the same 1,024 rows repeated: values, not their variety, matter for timing
    HIPCHECK(hipMemcpy(A + (size_t)blk * 1024 * K, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  std::vector<int> hm(S), hn(S);
  for (int i = 0; i < S; ++i) {
    hm[i] = (int)(rnd() * 1024) % 1024 + 1024 * ((int)(rnd() * 64) % 64);
    hn[i] = (int)(rnd() * N) % N;
  }
  HIPCHECK(hipMemcpy(ms, hm.data(), S * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(ns, hn.data(), S * 4, hipMemcpyHostToDevice));

  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(((int64_t)M * K + 255) / 256)), dim3(256), 0, 0, A, M, K, 0, A3);
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)(((int64_t)N * K + 255) / 256)), dim3(256), 0, 0, B, N, K, 1, B3);
  hipLaunchKernelGGL(ref_kernel, dim3((S + 255) / 256), dim3(256), 0, 0, A, B, N, K, ms, ns, S, ref64, ref32);
  HIPCHECK(hipDeviceSynchronize());
  std::vector<double> r64(S);
  std::vector<float> r32(S), hc(S);
  HIPCHECK(hipMemcpy(r64.data(), ref64, S * 8, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(r32.data(), ref32, S * 4, hipMemcpyDeviceToHost));
  double scale = 0.0, e32 = 0.0;
  for (int i = 0; i < S; ++i) {
    scale = fmax(scale, fabs(r64[i]));
    e32 = fmax(e32, fabs((double)r32[i] - r64[i]));
  }
  const double flop = 2.0 * M * N * K;
  printf("shape M=%d K=%d N=%d   max|ref| %.4g   plain fp32 fma chain: max err %.3g (%.2g of max|ref|)\n", M, K, N, scale, e32,
         e32 / scale);
  float t[3];
  t[0] = run<6>(A3, B3, M, N, K, C, 5);
  for (int variant = 0; variant < 3; ++variant) {
    const int np = variant == 0 ? 6 : (variant == 1 ? 3 : 1);
    const float msr = variant == 0 ? t[0] : (variant == 1 ? run<3>(A3, B3, M, N, K, C, 5) : run<1>(A3, B3, M, N, K, C, 5));
    if (variant == 0) run<6>(A3, B3, M, N, K, C, 1);  // leave the 6-product result in C for the error check below
    if (variant == 0 || true) {
      if (variant == 1) run<3>(A3, B3, M, N, K, C, 1);
      if (variant == 2) run<1>(A3, B3, M, N, K, C, 1);
      HIPCHECK(hipDeviceSynchronize());
      double err = 0.0;
      for (int i = 0; i < S; ++i) {
        float c;
        HIPCHECK(hipMemcpy(&c, C + (size_t)hm[i] * N + hn[i], 4, hipMemcpyDeviceToHost));
        err = fmax(err, fabs((double)c - r64[i]));
      }
      printf("bf16 split, %d product(s): %.3f ms  = %.1f TFLOP/s fp32-equivalent   max err %.3g (%.2g of max|ref|)\n", np, msr,
             flop / (msr * 1e-3) / 1e12, err, err / scale);
    }
  }
  return 0;
}
