// Dense layers (nn.Linear) with run-time sizes on the 16-bit matrix pipe, fp32-accurate (round 4): the plane scheme of pconv.hip /
// fc2.hip (engine2.h "f16x3": every fp32 operand as two scaled fp16 planes, three products per k-group on v_mfma_f32_32x32x16_f16,
// fp32 accumulation) for the layers that glinear.hip runs on the f32-input matrix instructions:
//   NavPreNet1D  7616 -> 256, 6400 -> 512, 773 -> 512, 512 -> 512   (reference USTC_lab/nn/nav_encoder.py:99-106)
//   NavPreNet / NavPedPreNet  9216 -> 512 ...                        (nav_encoder.py:21-24, 59-63), the `mlp` helper (nn/utils.py:10-20)
//
//   forward        out[b][n] = act(sum_k in[b][k] W[n][k] + bias[n])          C = A B^T, A = in   (rows b), B = W   (rows n), reduction k
//   data gradient  din[b][k] = mask(b,k) sum_n dout[b][n] W[n][k]             C = A B^T, A = dout (rows b), B = W^T (rows k), reduction n
//   weight grad.   dW[n][k]  = sum_b dout[b][n] in[b][k]                      C = A^T B, A = dout, B = in, reduction b (split over workgroups)
//
// The first two are ONE kernel (nt_planes_kernel): the weights arrive pre-split from the pack kernel in fragment order with one
// power-of-two scale per layer ([column tile 128][k-group][plane][column][lane half][8 k] -- a k-block is 16 KB contiguous, copied as
// it is), the activations are split while they are staged with one power-of-two scale PER ROW (per sample), taken from the row's
// largest magnitude (engine2.h scale_of_amax) -- left by the tensor's producer where it has one (the pooling epilogue of pconv.hip for
// the flattened conv output, the data gradient below for its own output: `out_amax`), by the pre-pass row_amax_kernel otherwise --,
// so a sample whose activations or gradients are orders of magnitude below the batch's largest keeps its 22
// bits; the epilogue multiplies every output row by 1 / (S_row S_w).  The weight gradient stages both operands row-major as they
// lie in memory ([sample][column], 256-byte rows, 32-byte blocks XOR-swizzled by the row) and forms its fragments with the
// transposing LDS read (ds_read_b64_tr_b16), both operands under ONE scale for the batch (the largest of the row maxima): a sum over
// samples is accurate in the absolute sense, relative to its largest contribution.
//
// Tiles: 128 x 128 outputs per workgroup of four waves (64 x 64 each = 2 x 2 fragment tiles, 64 accumulators), k-blocks of 32, two
// 32 KB stages: two workgroups per CU (pconv.hip's A/B: at one workgroup per CU nothing covers the barriers and commits).
#include <cstdlib>

#include "engine2.h"
#include "ops.h"

namespace ddrl {

namespace plin {

using u2v = __attribute__((ext_vector_type(2))) unsigned;
using s4w = __attribute__((ext_vector_type(4))) short;

constexpr int STAGE = 32768, A_BYTES = 16384;          // bytes per stage: activation planes, then weight planes
constexpr int LDS_NT = 2 * STAGE + 2 * 128 * 4;        // + the tile's row scales, + the largest |output| of every row of the tile
constexpr int LDS_TN = 2 * STAGE;

// Largest magnitude of every row (the pre-pass for tensors whose producer does not leave it).  One wave per row.  accumulate: raise
// the slot instead of overwriting it (a row assembled from several producers: torch.cat).  Loads are 16 bytes wide (rows are padded to
// a multiple of four floats); columns >= width of the last load are the row's padding or a neighbour's columns and are masked out.
__global__ __launch_bounds__(256) void row_amax_kernel(const float* __restrict__ x, int64_t ld, int width, int n, float* __restrict__ amax, int accumulate) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* src = x + (int64_t)row * ld;
  float m = 0.0f;
  for (int i = lane * 4; i < width; i += 256) {
    const f4 v = ld4(src + i);
    const float y = i + 1 < width ? fabsf(v.y) : 0.0f, z = i + 2 < width ? fabsf(v.z) : 0.0f, w = i + 3 < width ? fabsf(v.w) : 0.0f;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), y)), fmaxf(z, w));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) {
    if (accumulate) amax_raise(m, amax + row);
    else amax[row] = m;
  }
}

// (both packed regions of a layer take the same scale: one pass over the weights leaves it in both headers)
__global__ __launch_bounds__(256) void weight_amax_kernel(const float* __restrict__ w, int64_t count, float* __restrict__ slot, float* __restrict__ slot2) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  amax_update(m, slot);
  amax_update(m, slot2);
}

// dst[column tile][k-group][plane][column 128][lane half 2][8 k] (16-bit), zero beyond the matrix; hdr[0] = largest |w| (in), hdr[1] = scale.
// transposed = 0 (forward): columns = n, reduction = k.  transposed = 1 (data gradient): columns = k, reduction = n.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int K, int N, int transposed, int kgs,
                                                   unsigned short* __restrict__ dst, float* __restrict__ hdr) {
  const int cols = transposed ? K : N, red = transposed ? N : K;
  const int64_t total = (int64_t)((cols + 127) / 128) * kgs * 2048;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float scale = plane_scale(hdr[0]);
  if (i == 0) hdr[1] = scale;
  if (i >= total) return;
  const int e = (int)(i & 7), hf = (int)((i >> 3) & 1), c = (int)((i >> 4) & 127);
  const int64_t q = i >> 11;  // (column tile, k-group)
  const int kg = (int)(q % kgs), ct = (int)(q / kgs);
  const int col = ct * 128 + c, r = kg * 16 + hf * 8 + e;
  float v = 0.0f;
  if (col < cols && r < red) v = transposed ? w[(int64_t)r * K + col] : w[(int64_t)col * K + r];
  unsigned short pl[NPL];
  planes_of(v, scale, pl);
  unsigned short* d = dst + q * (NPL * 2048) + c * 16 + hf * 8 + e;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 2048] = pl[p];
}

struct NtParams {
  const float* a;               // activations [n][lda]
  int64_t lda;
  const float* scales;          // per row: its largest magnitude
  const unsigned short* bp;     // packed weight planes
  const float* whdr;            // [1] = weight scale
  int n, cols, kgs;             // rows, output columns, k-groups of the packed planes (even)
  // forward
  const float* bias;
  int act, nsplit;
  float* part;                  // nsplit > 1: bias-free partial sums part[split][b][cols]
  // data gradient
  const float* mask;            // may be null
  int64_t ldm;
  float* out;
  int64_t ldo;
  // data gradient: the largest |out| of every row over the columns [amax_lo, amax_hi) is raised into out_amax[row] (may be null)
  float* out_amax;
  int amax_lo, amax_hi;
};

template <int DGRAD>
__global__ __launch_bounds__(256, 2) void nt_planes_kernel(NtParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int c0 = blockIdx.x * 128, b0 = blockIdx.y * 128, split = blockIdx.z;
  const int nkb = P.kgs / 2;
  const int per = (nkb + P.nsplit - 1) / P.nsplit;
  const int kb_begin = min(nkb, split * per), kb_end = min(nkb, kb_begin + per);
  float* lsc = (float*)(lds + 2 * STAGE);
  float* omax = lsc + 128;
  if (tid < 128) {
    lsc[tid] = scale_of_amax(P.scales[min(b0 + tid, P.n - 1)]);
    omax[tid] = 0.0f;
  }
  // ---- staging maps: quad (row = tid / 8 + 32 j, k = 4 (tid % 8) ..) of the 128 x 32 activation block
  const int k4 = tid & 7, rr = tid >> 3;
  const float* asrc[4];
  float asc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = min(b0 + rr + 32 * j, P.n - 1);
    asrc[j] = P.a + (int64_t)row * P.lda;
    asc[j] = scale_of_amax(P.scales[row]);
  }
  const int awr = ((k4 >> 2) * (NPL * 128) + rr) * 32 + ((k4 >> 1) & 1) * 16 + (k4 & 1) * 8;  // + plane * 4096, + 32 j rows = 1024 j
  const unsigned short* bsrc = P.bp + (int64_t)blockIdx.x * P.kgs * (NPL * 2048) + tid * 8;     // + kb * 8192 + j * 2048 (shorts)
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (wr * 64 + i * 32 + l31) * 32 + hi * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) bB[j] = A_BYTES + (wc * 64 + j * 32 + l31) * 32 + hi * 16;
  f4 ar[4], br[4];
  const int kmax = (int)P.lda - 4;
  auto fetch = [&](int kb) {
    const int kc = min(kb * 32 + k4 * 4, kmax);  // columns past the reduction length meet zero weights
#pragma unroll
    for (int j = 0; j < 4; ++j) ar[j] = ld4(asrc[j] + kc);
#pragma unroll
    for (int j = 0; j < 4; ++j) br[j] = *(const f4*)(bsrc + (int64_t)kb * 8192 + j * 2048);
  };
  auto commit = [&](char* st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned p0[NPL], p1[NPL];
      split_planes(ar[j].x, ar[j].y, asc[j], p0);
      split_planes(ar[j].z, ar[j].w, asc[j], p1);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u2v*)(st + awr + p * 4096 + j * 1024) = (u2v){p0[p], p1[p]};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f4*)(st + A_BYTES + (tid + 256 * j) * 16) = br[j];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  if (kb_begin < kb_end) {
    fetch(kb_begin);
    commit(lds);
    if (kb_begin + 1 < kb_end) fetch(kb_begin + 1);
    __syncthreads();
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      const char* cur = lds + ((kb - kb_begin) & 1) * STAGE;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        frag8 af[NPL][2], bf[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[p][i] = *(const frag8*)(cur + aA[i] + (g * NPL + p) * 4096);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[p][j] = *(const frag8*)(cur + bB[j] + (g * NPL + p) * 4096);
        }
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int m = 0; m < NPROD; ++m)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(af[PA[m]][i], bf[PB[m]][j], acc[i][j]);
      }
      if (kb + 1 < kb_end) {  // the other stage was released by the barrier that ended the previous block
        commit(lds + ((kb + 1 - kb_begin) & 1) * STAGE);
        if (kb + 2 < kb_end) fetch(kb + 2);
      }
      __syncthreads();
    }
  } else {
    __syncthreads();  // the row scales
  }
  // ---- epilogue: un-scale per row, then bias + ReLU (forward) / the producing layer's ReLU mask (data gradient)
  const float winv = 1.0f / P.whdr[1];
  float rmax[32];   // data gradient with out_amax: this lane's largest |output| of each of its 32 rows (index 16 i + r)
#pragma unroll
  for (int k = 0; k < 32; ++k) rmax[k] = 0.0f;
  float rinv[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) rinv[i][r] = winv / lsc[wr * 64 + i * 32 + acc_row(r, hi)];  // powers of two: exact
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = c0 + wc * 64 + j * 32 + l31;
    if (col >= P.cols) continue;
    const float bias = (!DGRAD && P.nsplit == 1) ? P.bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
        if (b >= P.n) continue;
        float v = acc[i][j][r] * rinv[i][r];
        if (DGRAD) {
          if (P.mask != nullptr && !(P.mask[(int64_t)b * P.ldm + col] > 0.0f)) v = 0.0f;
          P.out[(int64_t)b * P.ldo + col] = v;
          if (col >= P.amax_lo && col < P.amax_hi) rmax[16 * i + r] = fmaxf(rmax[16 * i + r], fabsf(v));
        } else if (P.nsplit > 1) {
          P.part[((int64_t)split * P.n + b) * P.cols + col] = v;
        } else {
          v += bias;
          if (P.act == 1) v = fmaxf(v, 0.0f);
          P.out[(int64_t)b * P.ldo + col] = v;
        }
      }
  }
  if (DGRAD && P.out_amax != nullptr) {   // block-uniform
    // row maxima over the 32 lanes of a half in 31 exchanges: at stage `mask` a lane keeps the half of its values whose index bit equals
    // its own lane bit and takes the partner's candidates for them, so after five stages lane l31 holds the maximum of row index l31
    // (rows of lanes past the batch hold zeros: they raise nothing)
    int cnt = 32;
#pragma unroll
    for (int mask = 16; mask >= 1; mask >>= 1) {
      cnt >>= 1;
      const bool upper = (l31 & mask) != 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < cnt) {
          const float send = upper ? rmax[k] : rmax[k + cnt], keep = upper ? rmax[k + cnt] : rmax[k];
          rmax[k] = fmaxf(keep, __shfl_xor(send, mask, 64));
        }
      }
    }
    lds_amax_raise(rmax[0], omax + wr * 64 + (l31 >> 4) * 32 + acc_row(l31 & 15, hi));   // the two column halves (wc) meet here
    __syncthreads();
    if (tid < 128 && b0 + tid < P.n) amax_raise(omax[tid], P.out_amax + b0 + tid);
  }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ frag8 tr_frag(const char* lds, int off_lo, int off_hi) {
  typedef s4w __attribute__((address_space(3))) * lds_s4;
  const s4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_lo));
  const s4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s8w;
  return __builtin_bit_cast(frag8, (s8w)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct TnParams {
  const float* dout;
  int64_t ldd;
  const float* in;
  int64_t ldi;
  const float* sc_d;
  const float* sc_i;
  float* part;  // [nsplit][N * K + N]
  int n, K, N, nsplit;
};

// stage: [plane][32 samples][128 columns] 16-bit per operand, 256-byte rows; 32-byte block index XOR (row & 3): the four rows a
// 16-lane group of the transposing read touches lie in four different bank groups
__global__ __launch_bounds__(256, 2) void tn_planes_kernel(TnParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  __shared__ float s_min[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wi = wave >> 1, wx = wave & 1;
  const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128, split = blockIdx.z;
  {   // one scale per operand for the whole batch, from the largest of the row magnitudes (sc_d / sc_i hold magnitudes)
    float ma = 0.0f, mb = 0.0f;
    for (int i = tid; i < P.n; i += 256) {
      ma = fmaxf(ma, P.sc_d[i]);
      mb = fmaxf(mb, P.sc_i[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ma = fmaxf(ma, __shfl_xor(ma, off, 64));
      mb = fmaxf(mb, __shfl_xor(mb, off, 64));
    }
    if (lane == 0) {
      s_min[0][wave] = ma;
      s_min[1][wave] = mb;
    }
  }
  __syncthreads();
  const float sd = scale_of_amax(fmaxf(fmaxf(s_min[0][0], s_min[0][1]), fmaxf(s_min[0][2], s_min[0][3])));
  const float sa = scale_of_amax(fmaxf(fmaxf(s_min[1][0], s_min[1][1]), fmaxf(s_min[1][2], s_min[1][3])));
  const float inv = 1.0f / (sd * sa);
  const int nst = (P.n + 31) / 32;
  const int per = (nst + P.nsplit - 1) / P.nsplit;
  const int st_begin = min(nst, split * per), st_end = min(nst, st_begin + per);
  // ---- staging maps: quad (sample = tid / 32 + 8 j, columns 4 (tid % 32) ..) of both 32 x 128 blocks
  const int c4 = tid & 31, sr = tid >> 5;
  const int acol = n0 + c4 * 4, bcol = k0 + c4 * 4;
  const bool aok = acol < P.N, bok = bcol < (int)P.ldi;  // (N is a multiple of 4; columns [K, ldi) hold finite padding, their outputs are dropped)
  const float* asrc = P.dout + (aok ? acol : 0);
  const float* bsrc = P.in + (bok ? bcol : 0);
  const int wr0 = sr * 256 + ((c4 * 8) ^ ((sr & 3) * 32));  // + 8 j rows = 2048 j (row & 3 unchanged), + plane * 8192
  f4 ar[4], br[4], bsum = zero4();
  auto fetch = [&](int st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = min(st * 32 + sr + 8 * j, P.n - 1);
      ar[j] = ld4(asrc + (int64_t)b * P.ldd);
      br[j] = ld4(bsrc + (int64_t)b * P.ldi);
    }
  };
  auto commit = [&](int st, char* dst) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = st * 32 + sr + 8 * j < P.n;  // samples past the batch contribute zero
      const f4 a = (live && aok) ? ar[j] : zero4(), b = (live && bok) ? br[j] : zero4();
      bsum += a;
      unsigned p0[NPL], p1[NPL];
      split_planes(a.x, a.y, sd, p0);
      split_planes(a.z, a.w, sd, p1);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u2v*)(dst + wr0 + j * 2048 + p * 8192) = (u2v){p0[p], p1[p]};
      split_planes(b.x, b.y, sa, p0);
      split_planes(b.z, b.w, sa, p1);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u2v*)(dst + A_BYTES + wr0 + j * 2048 + p * 8192) = (u2v){p0[p], p1[p]};
    }
  };
  // ---- fragment addresses: 16-lane group g16 -> columns 16 (g16 & 1) .. +15 of the 32-column fragment, k-values 8 (g16 >> 1) .. +7;
  // inside the group lane 4 q + pp supplies row q (first read) / q + 4 (second), 4-column chunk pp
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (8 * (g16 >> 1) + q) * 256 + ((((wi * 64 + i * 32) * 2) + (g16 & 1) * 32 + pp * 8) ^ (q * 32));
#pragma unroll
  for (int j = 0; j < 2; ++j) bB[j] = A_BYTES + (8 * (g16 >> 1) + q) * 256 + ((((wx * 64 + j * 32) * 2) + (g16 & 1) * 32 + pp * 8) ^ (q * 32));
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  if (st_begin < st_end) {
    fetch(st_begin);
    commit(st_begin, lds);
    if (st_begin + 1 < st_end) fetch(st_begin + 1);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
      const char* cur = lds + ((st - st_begin) & 1) * STAGE;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        frag8 af[NPL][2], bf[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[p][i] = tr_frag(cur, aA[i] + p * 8192 + g * 4096, aA[i] + p * 8192 + g * 4096 + 1024);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[p][j] = tr_frag(cur, bB[j] + p * 8192 + g * 4096, bB[j] + p * 8192 + g * 4096 + 1024);
        }
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int m = 0; m < NPROD; ++m)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(af[PA[m]][i], bf[PB[m]][j], acc[i][j]);
      }
      if (st + 1 < st_end) {
        commit(st + 1, lds + ((st + 1 - st_begin) & 1) * STAGE);
        if (st + 2 < st_end) fetch(st + 2);
      }
      __syncthreads();
    }
  }
  // ---- epilogue: slab[n][k] (torch layout), then the bias partial (k tile 0)
  float* slab = P.part + (int64_t)split * ((int64_t)P.N * P.K + P.N);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wx * 64 + j * 32 + l31;
    if (k >= P.K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = n0 + wi * 64 + i * 32 + acc_row(r, hi);
        if (nn < P.N) slab[(int64_t)nn * P.K + k] = acc[i][j][r] * inv;
      }
  }
  if (blockIdx.x != 0) return;  // block-uniform
  float* red = (float*)lds;     // [8 sample groups][128 columns]
  st4(red + sr * 128 + c4 * 4, bsum);
  __syncthreads();
  if (tid < 128 && n0 + tid < P.N) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g * 128 + tid];
    slab[(int64_t)P.N * P.K + n0 + tid] = s;
  }
}

}  // namespace plin

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static bool planes_off() {
#ifdef DDRL_PLANES_BF16
  return true;
#else
  return false;
#endif
}

// Layers deep and wide enough for 128 x 128 x 32 tiles; launches of fewer than 128 rows (acting with a few environments) stay on the
// f32-input kernels, whose split-K fills the chip from one row tile.
bool linear_has_planes(int K, int N) { return !planes_off() && K >= 128 && N >= 64; }
bool linear_uses_planes(int n, int K, int N) { return linear_has_planes(K, N) && n >= 128; }

static int kgs_of(int red) { return 2 * ((red + 31) / 32); }
// floats of the packed plane regions behind the f32 layouts of wt / wn: planes (2 bytes x 2 planes) + a 64-float header
int64_t linear_planes_fwd_floats(int K, int N) { return (int64_t)((N + 127) / 128) * kgs_of(K) * (NPL * 2048) / 2 + 64; }
int64_t linear_planes_dgrad_floats(int K, int N) { return (int64_t)((K + 127) / 128) * kgs_of(N) * (NPL * 2048) / 2 + 64; }

void launch_linear_planes_pack(const float* w, int K, int N, float* pf, float* pd, hipStream_t st) {
  float* hdrs[2];
  for (int t = 0; t < 2; ++t) {
    const int cols = t ? K : N, kgs = kgs_of(t ? N : K);
    hdrs[t] = (t ? pd : pf) + (int64_t)((cols + 127) / 128) * kgs * (NPL * 2048) / 2;
    (void)hipMemsetAsync(hdrs[t], 0, 64 * sizeof(float), st);
  }
  const int64_t count = (int64_t)K * N;
  const unsigned ablocks = (unsigned)((count + 4095) / 4096 < 512 ? (count + 4095) / 4096 : 512);  // ~16 elements per thread, up to two workgroups per CU
  hipLaunchKernelGGL(plin::weight_amax_kernel, dim3(ablocks), dim3(256), 0, st, w, count, hdrs[0], hdrs[1]);
  for (int t = 0; t < 2; ++t) {
    float* region = t ? pd : pf;
    const int cols = t ? K : N, kgs = kgs_of(t ? N : K);
    float* hdr = hdrs[t];
    const int64_t total = (int64_t)((cols + 127) / 128) * kgs * 2048;
    hipLaunchKernelGGL(plin::pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, K, N, t, kgs, (unsigned short*)region, hdr);
  }
}

template <int DGRAD>
static void run_nt(const plin::NtParams& p, dim3 grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)plin::nt_planes_kernel<DGRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, plin::LDS_NT);
    configured = true;
  }
  hipLaunchKernelGGL(plin::nt_planes_kernel<DGRAD>, grid, dim3(256), plin::LDS_NT, st, p);
}

static void row_scales(const float* x, int64_t ld, int width, int n, float* amax, hipStream_t st, int accumulate = 0) {
  hipLaunchKernelGGL(plin::row_amax_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, ld, width, n, amax, accumulate);
}

int linear_planes_fwd_splits(int n, int K, int N) {
  const int tiles = ((N + 127) / 128) * ((n + 127) / 128);
  int s = (512 + tiles - 1) / tiles;
  const int cap = ((K + 31) / 32) / 8;  // at least 8 k-blocks per split (the bound ddrl_op_linear_ws_floats sizes the partials by)
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

// ws: n floats (row scales, rounded up to 64), then the split-K partials
void launch_row_amax(const float* x, int64_t ld, int width, int n, float* amax, int accumulate, hipStream_t st) { row_scales(x, ld, width, n, amax, st, accumulate); }

// given*: per-row scales the caller already holds (launch_row_scales on the same tensor), or nullptr for a pre-pass into the scratch
void launch_linear_planes_fwd(const float* in, int64_t ld_in, const float* pf, const float* bias, float* out, int64_t ld_out, int n, int K,
                              int N, int act, float* ws, const float* given, hipStream_t st) {
  const int kgs = kgs_of(K);
  const int64_t planes = (int64_t)((N + 127) / 128) * kgs * (NPL * 2048) / 2;
  const int S = linear_planes_fwd_splits(n, K, N);
  float* part = ws + (n + 63) / 64 * 64;
  if (!given) row_scales(in, ld_in, K, n, ws, st);
  plin::NtParams p{in, ld_in, given ? given : ws, (const unsigned short*)pf, pf + planes, n, N, kgs, bias, act, S, part, nullptr, 0, out, ld_out, nullptr, 0, 0};
  run_nt<0>(p, dim3((N + 127) / 128, (n + 127) / 128, S), st);
  if (S > 1) launch_linear_finish(part, S, n, N, bias, act, out, ld_out, st);
}

void launch_linear_planes_dgrad(const float* dout, int64_t ld_dout, const float* pd, const float* mask_src, int64_t ld_mask, float* din,
                                int64_t ld_din, int n, int K, int N, float* ws, const float* given, float* din_amax, int amax_lo, int amax_hi,
                                hipStream_t st) {
  const int kgs = kgs_of(N);
  const int64_t planes = (int64_t)((K + 127) / 128) * kgs * (NPL * 2048) / 2;
  if (!given) row_scales(dout, ld_dout, N, n, ws, st);
  plin::NtParams p{dout, ld_dout, given ? given : ws, (const unsigned short*)pd, pd + planes, n, K, kgs, nullptr, 0, 1, nullptr, mask_src, ld_mask, din, ld_din,
                   din_amax, amax_lo, amax_hi};
  run_nt<1>(p, dim3((K + 127) / 128, (n + 127) / 128, 1), st);
}

int linear_planes_wgrad_splits(int n, int K, int N) {
  const int tiles = ((K + 127) / 128) * ((N + 127) / 128);
  int s = (1024 + tiles - 1) / tiles;  // two workgroups per CU, about two rounds
  const int cap = (n + 127) / 128;     // at least four stages of 32 samples per split
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

// part: S slabs of N * K + N floats, then 2 n floats for the row scales
void launch_linear_planes_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* part, int n, int K, int N,
                                float* dw, float* db, const float* given_in, const float* given_dout, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)plin::tn_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, plin::LDS_TN);
    configured = true;
  }
  const int S = linear_planes_wgrad_splits(n, K, N);
  const int64_t slab = (int64_t)N * K + N;
  float* sc_i = part + (int64_t)S * slab;
  float* sc_d = sc_i + n;
  if (!given_in) row_scales(in, ld_in, K, n, sc_i, st);
  if (!given_dout) row_scales(dout, ld_dout, N, n, sc_d, st);
  plin::TnParams p{dout, ld_dout, in, ld_in, given_dout ? given_dout : sc_d, given_in ? given_in : sc_i, part, n, K, N, S};
  hipLaunchKernelGGL(plin::tn_planes_kernel, dim3((K + 127) / 128, (N + 127) / 128, S), dim3(256), plin::LDS_TN, st, p);
  launch_reduce_slabs2(part, S, slab, (int64_t)N * K, dw, N, db, st);
}

}  // namespace ddrl
