// Generic dense layer (nn.Linear) on the pipelined f32-MFMA engine: forward (+bias, +ReLU),
// data gradient (+ReLU mask of the producing layer) and weight/bias gradient, with runtime
// sizes and leading dimensions.  These serve the encoders outside the Atari fast path
// (reference USTC_lab/nn/nav_encoder.py:21-24,103-106, USTC_lab/nn/mlp_encoder.py:18 and the
// `mlp` helper USTC_lab/nn/utils.py:10-20); the 3136->512 Atari layer keeps its own kernels (fc2.hip).
// Since round 4 layers of K >= 128, N >= 64 in launches of >= 128 rows run as fp16 plane products instead (plin.hip; the launchers at
// the bottom dispatch): these kernels remain for small layers, small launches (acting with a few environments: split-K fills the chip
// from one row tile).
//
// Layout contract (checked by the C ABI): every leading dimension is a multiple of 4 floats and
// every base pointer 16-byte aligned, so that all staging loads are aligned f4 loads;
//   wt  [Kp32][N]  = W^T, rows K..Kp32-1 zero   (forward B operand; Kp32 = K rounded up to 32)
//   wn  [N][Kp4]   = W,   cols K..Kp4-1 zero    (data-gradient B operand; Kp4 = K rounded up to 4)
// Tiles are 128 x 128 with a 32-deep k-block, as in fc2.hip.
#include "engine2.h"
#include "ops.h"

namespace ddrl {

namespace glin {

// X[128 rows][32 k] of a K-contiguous matrix -> LDS [row][33]; row and column clamped (reads past
// the logical K only ever meet zero weights)
struct RowTile {
  static constexpr int LD = 33, FLOATS = 128 * LD;
  __device__ __forceinline__ static void fetch(const float* __restrict__ src, int64_t ld, int row0, int nrows, int k0, int kmax4,
                                               int tid, f4 (&r)[4]) {
    const int k4 = tid & 7, rr = tid >> 3;
    const int kc = min(k0 + k4 * 4, kmax4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(row0 + rr + 32 * j, nrows - 1);
      r[j] = ld4(src + (int64_t)row * ld + kc);
    }
  }
  __device__ __forceinline__ static void commit(float* __restrict__ dst, int tid, const f4 (&r)[4]) {
    const int k4 = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = dst + (rr + 32 * j) * LD + k4 * 4;
      p[0] = r[j].x; p[1] = r[j].y; p[2] = r[j].z; p[3] = r[j].w;
    }
  }
};

// X[32 k][128 cols] of a col-contiguous matrix -> LDS as is.  k (the reduction index) beyond nk is
// masked to zero at commit; columns beyond ncols4 are clamped (they only feed discarded outputs).
struct KTile {
  static constexpr int LD = 128, FLOATS = 32 * LD;
  __device__ __forceinline__ static unsigned fetch(const float* __restrict__ src, int64_t ld, int k0, int nk, int col0, int ncols4,
                                                   int tid, f4 (&r)[4]) {
    const int c4 = tid & 31, kk = tid >> 5;
    const int col = (col0 + c4 * 4) < ncols4 ? col0 + c4 * 4 : 0;
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + kk + 8 * j;
      ok |= (k < nk ? 1u : 0u) << j;
      r[j] = ld4(src + (int64_t)min(k, nk - 1) * ld + col);
    }
    return ok;
  }
  __device__ __forceinline__ static void commit(float* __restrict__ dst, int tid, const f4 (&r)[4], unsigned ok) {
    const int c4 = tid & 31, kk = tid >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) st4(dst + (kk + 8 * j) * LD + c4 * 4, ((ok >> j) & 1u) ? r[j] : zero4());
  }
};

struct Common {
  static constexpr int IGLP = 1;  // as fc2.hip
  static constexpr int THREADS = 256, TM = 2, TN = 2, KSTEPS = 16;
  int abase[2], bbase[2];
  int kb_begin, kb_end;
  int wr, wc, l31, hi;
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void lanes(int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
    wr = wave >> 1;
    wc = wave & 1;
  }
};

// out[b][n] = act( sum_k in[b][k] W[n][k] + bias[n] )      rows = b, cols = n, reduction = k
struct Fwd : Common {
  static constexpr int A_OFF = 0, B_OFF = RowTile::FLOATS, STAGE = RowTile::FLOATS + KTile::FLOATS;
  struct Params {
    const float* in;
    int64_t ld_in;
    const float* wt;  // [Kp32][N]
    const float* bias;
    float* out;
    int64_t ld_out;
    int n, K, N, act;  // act: 0 none, 1 relu
    int nsplit;        // > 1: split-K, bias-free partial sums part[split][b][N] (small n x N: fills the GPU)
    float* part;
  };
  struct Regs {
    f4 a[4], b[4];
    unsigned ok;
  };
  int b0, n0, split;
  static constexpr int aoff(int s) { return 2 * s; }
  static constexpr int boff(int s) { return 2 * s * KTile::LD; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    lanes(tid);
    n0 = blockIdx.x * 128;
    b0 = blockIdx.y * 128;
    split = blockIdx.z;
    const int nkb = (p.K + 31) / 32;
    const int per = (nkb + p.nsplit - 1) / p.nsplit;
    kb_begin = min(nkb, split * per);
    kb_end = min(nkb, kb_begin + per);
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = A_OFF + (wr * 64 + i * 32 + l31) * RowTile::LD + hi;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = B_OFF + hi * KTile::LD + wc * 64 + j * 32 + l31;
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    RowTile::fetch(p.in, p.ld_in, b0, p.n, kb * 32, (int)p.ld_in - 4, threadIdx.x, r.a);
    r.ok = KTile::fetch(p.wt, p.N, kb * 32, (p.K + 31) / 32 * 32, n0, p.N, threadIdx.x, r.b);
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    RowTile::commit(buf + A_OFF, threadIdx.x, r.a);
    KTile::commit(buf + B_OFF, threadIdx.x, r.b, r.ok);
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float*) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + l31;
      if (n >= p.N) continue;
      if (p.nsplit > 1) {
        float* dst = p.part + (int64_t)split * p.n * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
            if (b < p.n) dst[(int64_t)b * p.N + n] = acc[i][j][r];
          }
        continue;
      }
      const float bias = p.bias[n];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
          float v = acc[i][j][r] + bias;
          if (p.act == 1) v = fmaxf(v, 0.0f);
          if (b < p.n) p.out[(int64_t)b * p.ld_out + n] = v;
        }
    }
  }
};

// din[b][k] = mask(b,k) * sum_n dout[b][n] W[n][k]          rows = b, cols = k, reduction = n
// mask: relu'(mask_src[b][k]) of the layer that produced this input (mask_src = its output), or 1
struct Dgrad : Common {
  static constexpr int A_OFF = 0, B_OFF = RowTile::FLOATS, STAGE = RowTile::FLOATS + KTile::FLOATS;
  struct Params {
    const float* dout;
    int64_t ld_dout;
    const float* wn;  // [N][Kp4]
    const float* mask_src;  // may be null
    int64_t ld_mask;
    float* din;
    int64_t ld_din;
    int n, K, N, Kp4;
  };
  struct Regs {
    f4 a[4], b[4];
    unsigned ok;
  };
  int b0, k0;
  static constexpr int aoff(int s) { return 2 * s; }
  static constexpr int boff(int s) { return 2 * s * KTile::LD; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    lanes(tid);
    k0 = blockIdx.x * 128;
    b0 = blockIdx.y * 128;
    kb_begin = 0;
    kb_end = (p.N + 31) / 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = A_OFF + (wr * 64 + i * 32 + l31) * RowTile::LD + hi;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = B_OFF + hi * KTile::LD + wc * 64 + j * 32 + l31;
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    RowTile::fetch(p.dout, p.ld_dout, b0, p.n, kb * 32, (int)p.ld_dout - 4, threadIdx.x, r.a);
    r.ok = KTile::fetch(p.wn, p.Kp4, kb * 32, p.N, k0, p.Kp4, threadIdx.x, r.b);
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    RowTile::commit(buf + A_OFF, threadIdx.x, r.a);
    KTile::commit(buf + B_OFF, threadIdx.x, r.b, r.ok);
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float*) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wc * 64 + j * 32 + l31;
      if (k >= p.K) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
          if (b < p.n) {
            float v = acc[i][j][r];
            if (p.mask_src != nullptr && !(p.mask_src[(int64_t)b * p.ld_mask + k] > 0.0f)) v = 0.0f;
            p.din[(int64_t)b * p.ld_din + k] = v;
          }
        }
    }
  }
};

// part[s][n][k] = sum_{b in split s} dout[b][n] in[b][k] ; bias partial [N] appended per slab
// rows = n, cols = k, reduction = b
struct Wgrad : Common {
  static constexpr int A_OFF = 0, B_OFF = KTile::FLOATS, STAGE = 2 * KTile::FLOATS;
  struct Params {
    const float* dout;
    int64_t ld_dout;
    const float* in;
    int64_t ld_in;
    float* part;  // [nsplit][N*K + N]
    int n, K, N, nsplit;
  };
  struct Regs {
    f4 a[4], b[4];
    unsigned oka, okb;
  };
  int split, n0, k0;
  f4 bsum;
  static constexpr int aoff(int s) { return 2 * s * KTile::LD; }
  static constexpr int boff(int s) { return 2 * s * KTile::LD; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    lanes(tid);
    split = blockIdx.z;
    k0 = blockIdx.x * 128;
    n0 = blockIdx.y * 128;
    const int nkb = (p.n + 31) / 32;
    const int per = (nkb + p.nsplit - 1) / p.nsplit;
    kb_begin = min(nkb, split * per);
    kb_end = min(nkb, kb_begin + per);
    bsum = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = A_OFF + hi * KTile::LD + wr * 64 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = B_OFF + hi * KTile::LD + wc * 64 + j * 32 + l31;
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    r.oka = KTile::fetch(p.dout, p.ld_dout, kb * 32, p.n, n0, p.N, threadIdx.x, r.a);
    r.okb = KTile::fetch(p.in, p.ld_in, kb * 32, p.n, k0, (int)p.ld_in, threadIdx.x, r.b);
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    KTile::commit(buf + A_OFF, threadIdx.x, r.a, r.oka);  // samples >= n contribute zero
    KTile::commit(buf + B_OFF, threadIdx.x, r.b, r.okb);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((r.oka >> j) & 1u) bsum += r.a[j];  // this thread always holds the same 4 columns of dout
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float* lds) {
    float* slab = p.part + (int64_t)split * ((int64_t)p.N * p.K + p.N);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wc * 64 + j * 32 + l31;
      if (k >= p.K) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wr * 64 + i * 32 + acc_row(r, hi);
          if (n < p.N) slab[(int64_t)n * p.K + k] = acc[i][j][r];
        }
    }
    if (blockIdx.x == 0) {  // one column tile per (row tile, split) owns the bias partial
      const int c4 = threadIdx.x & 31, kk = threadIdx.x >> 5;
      st4(lds + kk * 128 + c4 * 4, bsum);
      __syncthreads();
      if (threadIdx.x < 128 && n0 + threadIdx.x < p.N) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += lds[q * 128 + threadIdx.x];
        slab[(int64_t)p.N * p.K + n0 + threadIdx.x] = s;
      }
    }
  }
};

}  // namespace glin

// wt[k][n] = W[n][k] (zero rows up to Kp32), wn[n][k] = W[n][k] (zero columns up to Kp4)
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* __restrict__ w, int K, int N, int Kp32, int Kp4,
                                                          float* __restrict__ wt, float* __restrict__ wn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)Kp32 * N) {
    const int k = (int)(i / N), n = (int)(i % N);
    wt[i] = k < K ? w[(int64_t)n * K + k] : 0.0f;
  }
  if (i < (int64_t)N * Kp4) {
    const int n = (int)(i / Kp4), k = (int)(i % Kp4);
    wn[i] = k < K ? w[(int64_t)n * K + k] : 0.0f;
  }
}

// dst[i] = sum_s part[s * slab_stride + i]   (fixed order)
// (two destinations: a slab is [weight gradient | bias gradient]; element i < c0 goes to dst0[i], the rest to dst1[i - c0] -- one launch
// instead of two, the same per-element sums)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, int nsplit, int64_t slab_stride,
                                                           int64_t count, float* __restrict__ dst0, int64_t c0, float* __restrict__ dst1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;  // summed in double, rounded once (optim.hip reduce_partials_kernel)
  // slab order, as ever -- but sixteen loads are requested before the first of their adds (a run-time trip count kept the compiler from
  // hoisting them: one round trip to memory per slab, 74 us for the 64-slab weight gradients of the nav encoder, 13 of its 18 launches)
  int sp = 0;
  for (; sp + 16 <= nsplit; sp += 16) {
    float x[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) x[t] = part[(int64_t)(sp + t) * slab_stride + i];
#pragma unroll
    for (int t = 0; t < 16; ++t) s += (double)x[t];
  }
  for (; sp + 4 <= nsplit; sp += 4) {
    float x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = part[(int64_t)(sp + t) * slab_stride + i];
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (double)x[t];
  }
  for (; sp < nsplit; ++sp) s += (double)part[(int64_t)sp * slab_stride + i];
  if (i < c0) dst0[i] = (float)s;
  else dst1[i - c0] = (float)s;
}

// Many slabs, few elements (thin weight gradients: hundreds of splits of a few hundred values): 16
// lane groups walk the slabs in parallel (group g takes slabs g, g+16, ...), then one lane adds the
// 16 partial sums in group order -- still a fixed order, so still deterministic.
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ part, int nsplit, int64_t slab_stride,
                                                                int64_t count, float* __restrict__ dst0, int64_t c0, float* __restrict__ dst1) {
  __shared__ double red[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + e;
  double s = 0.0;
  if (i < count)
    for (int sp = g; sp < nsplit; sp += 16) s += (double)part[(int64_t)sp * slab_stride + i];
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < count) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e];
    if (i < c0) dst0[i] = (float)t;
    else dst1[i - c0] = (float)t;
  }
}

// dst0[0 .. c0) and dst1[0 .. c1) = the slabs' first c0 + c1 elements summed over the slabs
void launch_reduce_slabs2(const float* part, int nsplit, int64_t slab_stride, int64_t c0, float* dst0, int64_t c1, float* dst1, hipStream_t st) {
  const int64_t count = c0 + c1;
  if (count <= 0) return;
  if (nsplit >= 64 && count < 65536)
    hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, st, part, nsplit,
                       slab_stride, count, dst0, c0, dst1);
  else
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, part, nsplit,
                       slab_stride, count, dst0, c0, dst1);
}

void launch_reduce_slabs(const float* part, int nsplit, int64_t slab_stride, int64_t count, float* dst, hipStream_t st) {
  launch_reduce_slabs2(part, nsplit, slab_stride, count, dst, 0, nullptr, st);
}

__global__ __launch_bounds__(256) void accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] += src[i];
}
void launch_accumulate(float* dst, const float* src, int64_t count, hipStream_t st) {
  int64_t wgs = (count + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)wgs), dim3(256), 0, st, dst, src, count);
}

// d[b][k] = act[b][k] > 0 ? d[b][k] : 0   (ReLU backward where the encoder's last layer ends in a ReLU)
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ d, int64_t ld_d, const float* __restrict__ act,
                                                        int64_t ld_act, int64_t n, int width) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * width) return;
  const int64_t b = i / width;
  const int k = (int)(i % width);
  if (!(act[b * ld_act + k] > 0.0f)) d[b * ld_d + k] = 0.0f;
}
void launch_relu_mask(float* d, int64_t ld_d, const float* act, int64_t ld_act, int64_t n, int width, hipStream_t st) {
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n * width + 255) / 256)), dim3(256), 0, st, d, ld_d, act, ld_act, n,
                     width);
}

static int64_t wt_f32_floats(int K, int N) { return (int64_t)((K + 31) / 32 * 32) * N; }
static int64_t wn_f32_floats(int K, int N) { return (int64_t)N * ((K + 3) / 4 * 4); }

void launch_linear_pack(const float* w, int K, int N, float* wt, float* wn, hipStream_t st) {
  if (linear_has_planes(K, N)) launch_linear_planes_pack(w, K, N, wt + wt_f32_floats(K, N), wn + wn_f32_floats(K, N), st);
  const int Kp32 = (K + 31) / 32 * 32, Kp4 = (K + 3) / 4 * 4;
  const int64_t total = (int64_t)N * (Kp32 > Kp4 ? Kp32 : Kp4);
  hipLaunchKernelGGL(linear_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, K, N, Kp32, Kp4, wt, wn);
}

// out[b][n] = act(bias[n] + sum_s part[s][b][n])   (fixed order)
__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ part, int nsplit, int64_t n, int N,
                                                            const float* __restrict__ bias, int act, float* __restrict__ out,
                                                            int64_t ld_out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * N) return;
  const int64_t b = i / N;
  const int c = (int)(i % N);
  float s = 0.0f;
  for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * n * N + i];
  s += bias[c];
  if (act == 1) s = fmaxf(s, 0.0f);
  out[b * ld_out + c] = s;
}

int linear_fwd_splits(int n, int K, int N) {
  const int tiles = ((N + 127) / 128) * ((n + 127) / 128);
  int s = (512 + tiles - 1) / tiles;
  const int cap = ((K + 31) / 32) / 8;  // at least 8 k-blocks per split
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

void launch_linear_finish(const float* part, int nsplit, int n, int N, const float* bias, int act, float* out, int64_t ld_out, hipStream_t st) {
  hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)(((int64_t)n * N + 255) / 256)), dim3(256), 0, st, part, nsplit, (int64_t)n, N,
                     bias, act, out, ld_out);
}

void launch_linear_fwd(const float* in, int64_t ld_in, const float* wt, const float* bias, float* out, int64_t ld_out, int n,
                       int K, int N, int act, float* part, const float* in_scales, hipStream_t st) {
  if (part && linear_uses_planes(n, K, N)) {
    launch_linear_planes_fwd(in, ld_in, wt + wt_f32_floats(K, N), bias, out, ld_out, n, K, N, act, part, in_scales, st);
    return;
  }
  const int S = part ? linear_fwd_splits(n, K, N) : 1;
  glin::Fwd::Params p{in, ld_in, wt, bias, out, ld_out, n, K, N, act, S, part};
  launch_engine2<glin::Fwd>(dim3((N + 127) / 128, (n + 127) / 128, S), p, st);
  if (S > 1)
    hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)(((int64_t)n * N + 255) / 256)), dim3(256), 0, st, part, S,
                       (int64_t)n, N, bias, act, out, ld_out);
}

void launch_linear_dgrad(const float* dout, int64_t ld_dout, const float* wn, const float* mask_src, int64_t ld_mask,
                         float* din, int64_t ld_din, int n, int K, int N, float* ws, const float* dout_scales, float* din_amax, int amax_lo,
                         int amax_hi, hipStream_t st) {
  if (ws && linear_uses_planes(n, K, N)) {
    launch_linear_planes_dgrad(dout, ld_dout, wn + wn_f32_floats(K, N), mask_src, ld_mask, din, ld_din, n, K, N, ws, dout_scales, din_amax,
                               amax_lo, amax_hi, st);
    return;
  }
  glin::Dgrad::Params p{dout, ld_dout, wn, mask_src, ld_mask, din, ld_din, n, K, N, (K + 3) / 4 * 4};
  launch_engine2<glin::Dgrad>(dim3((K + 127) / 128, (n + 127) / 128, 1), p, st);
  // small launches (fewer than 128 rows) and small layers: the magnitudes come from a pass over the rows just written (api_ops.hip
  // checked that the column range allows 16-byte loads)
  if (din_amax != nullptr) launch_row_amax(din + amax_lo, ld_din, amax_hi - amax_lo, n, din_amax, 1, st);
}

int linear_wgrad_splits(int n, int K, int N) {
  // ~512 workgroups per launch, at least 2 k-blocks (64 samples) per split
  const int tiles = ((K + 127) / 128) * ((N + 127) / 128);
  int s = (512 + tiles - 1) / tiles;
  const int cap = (n + 63) / 64;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

void launch_linear_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* part, int n, int K, int N,
                         float* dw, float* db, const float* in_scales, const float* dout_scales, hipStream_t st) {
  if (linear_uses_planes(n, K, N)) {
    launch_linear_planes_wgrad(in, ld_in, dout, ld_dout, part, n, K, N, dw, db, in_scales, dout_scales, st);
    return;
  }
  const int S = linear_wgrad_splits(n, K, N);
  glin::Wgrad::Params p{dout, ld_dout, in, ld_in, part, n, K, N, S};
  launch_engine2<glin::Wgrad>(dim3((K + 127) / 128, (N + 127) / 128, S), p, st);
  const int64_t slab = (int64_t)N * K + N;
  launch_reduce_slabs2(part, S, slab, (int64_t)N * K, dw, N, db, st);
}

}  // namespace ddrl
