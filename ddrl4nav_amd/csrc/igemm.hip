// Implicit-GEMM kernels for the AtariPreNet encoder (forward, weight-gradient, data-gradient)
// on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain at the fp32
// vector rate), one LDS-staged tile pipeline shared by all eleven GEMM-shaped ops.
//
// Every op computes  C[row][col] = sum_k A(row,k) * B(k,col)  with
//   rows -> MFMA A operand -> accumulator registers, cols -> MFMA B operand -> lanes,
// so that stores are coalesced along `col`.  Tiles are staged k-major in LDS:
//   As[kk][row] (LDA = BM+1), Bs[kk][col] (LDB = BN+1); operand reads are conflict-free
// ds_read_b32 (32 consecutive dwords per lane group).
//
// Reference arithmetic being replaced: F.conv2d / F.leaky_relu / nn.Linear forward and their
// autograd backward in USTC_lab/nn/atari_encoder.py:25-32 (called from ppo.py:82,122-123).
#include "kernels.h"

namespace ddrl {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BK = 32;

template <int BM_, int BN_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int WM = (BM >= 128) ? 2 : 1;  // wave grid over rows
  static constexpr int WN = 4 / WM;               // wave grid over cols
  static constexpr int TM = BM / WM / 32;         // 32x32 tiles per wave along rows
  static constexpr int TN = BN / WN / 32;
  static constexpr int LDA = BM + 1;
  static constexpr int LDB = BN + 1;
  static constexpr int LDS_FLOATS = BK * (LDA + LDB);
};

// float32(u8 / 255.0) without a divide: q = x*r, one Newton correction with two fmas.  Equal
// to the reference's float64 divide + float32 cast for all 256 inputs (tests/test_lut.py).
__device__ __forceinline__ float u8_to_unit(uint8_t b) {
  const float x = (float)b;
  const float r = 1.0f / 255.0f;
  const float q = x * r;
  const float e = __builtin_fmaf(-255.0f, q, x);
  return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ float cvt_in(float v) { return v; }
__device__ __forceinline__ float cvt_in(uint8_t v) { return u8_to_unit(v); }

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : v * LEAKY; }
__device__ __forceinline__ float leaky_grad(float act, float g) { return act > 0.0f ? g : g * LEAKY; }

template <int L> struct Geo;
template <> struct Geo<1> {
  using in_t = uint8_t;
  static constexpr int CIN = 4, HIN = 84, OC = 32, KS = 8, S = 4, OW = 20, P = 400, K = 256;
};
template <> struct Geo<2> {
  using in_t = float;
  static constexpr int CIN = 32, HIN = 20, OC = 64, KS = 4, S = 2, OW = 9, P = 81, K = 512;
};
template <> struct Geo<3> {
  using in_t = float;
  static constexpr int CIN = 64, HIN = 9, OC = 64, KS = 3, S = 1, OW = 7, P = 49, K = 576;
};

// ============================================================================================
//                                        the tile engine
// ============================================================================================
template <class Op>
__global__ __launch_bounds__(256) void igemm_kernel(typename Op::Params P) {
  using T = typename Op::T;
  __shared__ float smem[T::LDS_FLOATS];
  float* As = smem;
  float* Bs = smem + BK * T::LDA;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave / T::WN, wc = wave % T::WN;

  Op op;
  op.init(P, tid);

  f32x16 acc[T::TM][T::TN];
#pragma unroll
  for (int i = 0; i < T::TM; ++i)
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int kb1 = op.kb_end;
  for (int kb = op.kb_begin; kb < kb1; ++kb) {
    op.load(P, kb, As, Bs, tid);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float a[T::TM], b[T::TN];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) a[i] = As[(2 * s + hi) * T::LDA + (wr * T::TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < T::TN; ++j) b[j] = Bs[(2 * s + hi) * T::LDB + (wc * T::TN + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < T::TN; ++j) {
    const int col = (wc * T::TN + j) * 32 + l31;
    auto cctx = op.col_ctx(P, col);
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wr * T::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        op.store(P, cctx, row, acc[i][j][r]);
      }
    }
  }
}

// ============================================================================================
//  conv forward:  out[b][oc][pix] = leaky( sum_k Wt[k][oc] * patch[b,pix][k] + bias[oc] )
//  rows = oc, cols = b*P + pix, reduction k = (ic,ky,kx)
// ============================================================================================
template <int L>
struct ConvFwdOp {
  using G = Geo<L>;
  using T = Tile<G::OC, 256>;
  struct Params {
    const typename G::in_t* in;  // [e][n][CIN][HIN][HIN]
    int64_t in_es;               // encoder stride of `in` (0: both encoders read the frames)
    const float* wt;             // [e][K][OC]
    const float* params;         // flat arena
    int64_t bias_off[2];
    float* out;                  // [e][n][OC][P]
    int64_t out_es;
    int n;                       // samples
  };
  int kb_begin, kb_end;
  int e, colbase;
  bool valid;
  const typename G::in_t* in;
  const float* wt;

  __device__ void init(const Params& p, int tid) {
    e = blockIdx.z;
    kb_begin = 0;
    kb_end = G::K / BK;
    const int c = blockIdx.x * T::BN + tid;
    valid = c < p.n * G::P;
    const int cc = valid ? c : 0;
    const int b = cc / G::P, pix = cc % G::P;
    const int oy = pix / G::OW, ox = pix % G::OW;
    colbase = b * (G::CIN * G::HIN * G::HIN) + oy * G::S * G::HIN + ox * G::S;
    in = p.in + e * p.in_es;
    wt = p.wt + (int64_t)e * G::K * G::OC;
  }
  __device__ void load(const Params&, int kb, float* As, float* Bs, int tid) {
    // A: 32 x OC weights, coalesced
    constexpr int RPT = 256 / G::OC;  // k-rows covered per pass
    {
      const int r = tid % G::OC, k0 = tid / G::OC;
#pragma unroll
      for (int j = 0; j < BK / RPT; ++j) {
        const int kk = k0 + j * RPT;
        As[kk * T::LDA + r] = wt[(kb * BK + kk) * G::OC + r];
      }
    }
    // B: this thread's column, 32 reduction indices (wave-uniform decomposition)
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const int k = kb * BK + kk;
      const int ic = k / (G::KS * G::KS), rem = k % (G::KS * G::KS);
      const int ky = rem / G::KS, kx = rem % G::KS;
      const int koff = ic * G::HIN * G::HIN + ky * G::HIN + kx;
      float v = 0.0f;
      if (valid) v = cvt_in(in[colbase + koff]);
      Bs[kk * T::LDB + tid] = v;
    }
  }
  struct ColCtx {
    int64_t base;
    bool ok;
  };
  __device__ ColCtx col_ctx(const Params& p, int col) {
    const int c = blockIdx.x * T::BN + col;
    ColCtx x;
    x.ok = c < p.n * G::P;
    const int b = c / G::P, pix = c % G::P;
    x.base = e * p.out_es + (int64_t)b * (G::OC * G::P) + pix;
    return x;
  }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    if (!x.ok) return;
    const float bias = p.params[p.bias_off[e] + row];
    p.out[x.base + row * G::P] = leaky(v + bias);
  }
};

// ============================================================================================
//  conv weight gradient (split over the sample dimension):
//    part[s][e][oc][k] = sum_{m in split s} dz[m][oc] * patch[m][k]
//  rows = oc, cols = k = (ic,ky,kx), reduction m = b*P + pix
// ============================================================================================
template <int L>
struct ConvWgradOp {
  using G = Geo<L>;
  using T = Tile<G::OC, 128>;
  struct Params {
    const typename G::in_t* in;
    int64_t in_es;
    const float* dz;  // [e][n][OC][P]
    int64_t dz_es;
    float* part;      // [nsplit][e][OC][K]
    int n;
    int nsplit;
  };
  int kb_begin, kb_end, e, split;
  int koff;
  bool kvalid;
  int mtotal;
  const typename G::in_t* in;
  const float* dz;

  __device__ void init(const Params& p, int tid) {
    e = blockIdx.z;
    split = blockIdx.y;
    mtotal = p.n * G::P;
    const int nkb = (mtotal + BK - 1) / BK;
    const int per = (nkb + p.nsplit - 1) / p.nsplit;
    kb_begin = split * per;
    kb_end = min(nkb, kb_begin + per);
    const int k = blockIdx.x * T::BN + (tid & 127);
    kvalid = k < G::K;
    const int kk = kvalid ? k : 0;
    const int ic = kk / (G::KS * G::KS), rem = kk % (G::KS * G::KS);
    koff = ic * G::HIN * G::HIN + (rem / G::KS) * G::HIN + rem % G::KS;
    in = p.in + e * p.in_es;
    dz = p.dz + e * p.dz_es;
  }
  __device__ void load(const Params&, int kb, float* As, float* Bs, int tid) {
    const int m0 = kb * BK;
    // A: As[mm][oc] = dz[b][oc][pix]; lanes along mm (contiguous pix), LDA odd -> no conflicts
    {
      const int mm = tid & 31, r0 = tid >> 5;
      const int m = m0 + mm;
      const bool ok = m < mtotal;
      const int mc = ok ? m : 0;
      const int b = mc / G::P, pix = mc % G::P;
      const float* src = dz + (int64_t)b * (G::OC * G::P) + pix;
#pragma unroll
      for (int j = 0; j < G::OC / 8; ++j) {
        const int r = r0 + j * 8;
        As[mm * T::LDA + r] = ok ? src[r * G::P] : 0.0f;
      }
    }
    // B: Bs[mm][k] = patch; lanes along k (fixed per thread), mm wave-uniform
    {
      const int cc = tid & 127;
      const int mh = __builtin_amdgcn_readfirstlane(tid >> 7);
#pragma unroll
      for (int j = 0; j < BK / 2; ++j) {
        const int mm = mh + 2 * j;
        const int m = m0 + mm;
        const bool ok = (m < mtotal) && kvalid;
        const int mc = (m < mtotal) ? m : 0;
        const int b = mc / G::P, pix = mc % G::P;
        const int oy = pix / G::OW, ox = pix % G::OW;
        const int64_t base = (int64_t)b * (G::CIN * G::HIN * G::HIN) + oy * G::S * G::HIN + ox * G::S;
        float v = 0.0f;
        if (ok) v = cvt_in(in[base + koff]);
        Bs[mm * T::LDB + cc] = v;
      }
    }
  }
  struct ColCtx {
    int k;
  };
  __device__ ColCtx col_ctx(const Params&, int col) { return ColCtx{(int)(blockIdx.x * T::BN + col)}; }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    if (x.k < G::K) p.part[(((int64_t)split * 2 + e) * G::OC + row) * G::K + x.k] = v;
  }
};

// ============================================================================================
//  conv3 data gradient:  dz2[b][ic][iy][ix] = leaky'(a2) * sum_{oc,ky,kx} dz3[b][oc][iy-ky][ix-kx] W3[oc][ic][ky][kx]
//  rows = ic, cols = b*81 + iy*9 + ix, reduction k = (oc,ky,kx)
// ============================================================================================
struct ConvDgrad3Op {
  using T = Tile<64, 256>;
  struct Params {
    const float* dz;   // dz3 [e][n][64][49]
    int64_t dz_es;
    const float* wd;   // [e][576][64]
    const float* act;  // a2 [e][n][64][81]
    float* out;        // dz2 [e][n][64][81]
    int64_t out_es;
    int n;
  };
  int kb_begin, kb_end, e, colbase;
  unsigned mask;
  const float* dz;
  const float* wd;
  __device__ void init(const Params& p, int tid) {
    e = blockIdx.z;
    kb_begin = 0;
    kb_end = C3_K / BK;
    const int c = blockIdx.x * T::BN + tid;
    const bool valid = c < p.n * C2_P;
    const int cc = valid ? c : 0;
    const int b = cc / C2_P, pix = cc % C2_P;
    const int iy = pix / 9, ix = pix % 9;
    colbase = b * FLAT + iy * 7 + ix;
    mask = 0;
    if (valid) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int oy = iy - ky, ox = ix - kx;
          if (oy >= 0 && oy < 7 && ox >= 0 && ox < 7) mask |= 1u << (ky * 3 + kx);
        }
    }
    dz = p.dz + e * p.dz_es;
    wd = p.wd + (int64_t)e * C3_K * 64;
  }
  __device__ void load(const Params&, int kb, float* As, float* Bs, int tid) {
    {
      const int r = tid & 63, k0 = tid >> 6;
#pragma unroll
      for (int j = 0; j < BK / 4; ++j) {
        const int kk = k0 + j * 4;
        As[kk * T::LDA + r] = wd[(kb * BK + kk) * 64 + r];
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const int k = kb * BK + kk;
      const int oc = k / 9, t = k % 9;
      const int koff = oc * 49 - (t / 3) * 7 - (t % 3);
      float v = 0.0f;
      if ((mask >> t) & 1u) v = dz[colbase + koff];
      Bs[kk * T::LDB + tid] = v;
    }
  }
  struct ColCtx {
    int64_t base;
    bool ok;
  };
  __device__ ColCtx col_ctx(const Params& p, int col) {
    const int c = blockIdx.x * T::BN + col;
    ColCtx x;
    x.ok = c < p.n * C2_P;
    const int b = c / C2_P, pix = c % C2_P;
    x.base = e * p.out_es + (int64_t)b * (64 * C2_P) + pix;
    return x;
  }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    if (!x.ok) return;
    const int64_t idx = x.base + row * C2_P;
    p.out[idx] = leaky_grad(p.act[idx], v);
  }
};

// ============================================================================================
//  conv2 data gradient, one launch slice per input-parity class (a,c):
//    dz1[b][ic][2p+a][2q+c] = leaky'(a1) * sum_{oc,u,v} dz2[b][oc][p-u][q-v] W2[oc][ic][2u+a][2v+c]
//  rows = ic (32), cols = b*100 + p*10 + q, reduction k = (oc,u,v) = 256
// ============================================================================================
struct ConvDgrad2Op {
  using T = Tile<32, 256>;
  struct Params {
    const float* dz;   // dz2 [e][n][64][81]
    int64_t dz_es;
    const float* wd;   // [e][4][256][32]
    const float* act;  // a1 [e][n][32][400]
    float* out;        // dz1 [e][n][32][400]
    int64_t out_es;
    int n;
  };
  int kb_begin, kb_end, e, cls, colbase;
  unsigned mask;
  const float* dz;
  const float* wd;
  __device__ void init(const Params& p, int tid) {
    e = blockIdx.z;
    cls = blockIdx.y;
    kb_begin = 0;
    kb_end = 256 / BK;
    const int c = blockIdx.x * T::BN + tid;
    const bool valid = c < p.n * 100;
    const int cc = valid ? c : 0;
    const int b = cc / 100, pq = cc % 100;
    const int pp = pq / 10, qq = pq % 10;
    colbase = b * (64 * C2_P) + pp * 9 + qq;
    mask = 0;
    if (valid) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int oy = pp - u, ox = qq - v;
          if (oy >= 0 && oy < 9 && ox >= 0 && ox < 9) mask |= 1u << (u * 2 + v);
        }
    }
    dz = p.dz + e * p.dz_es;
    wd = p.wd + ((int64_t)e * 4 + cls) * 256 * 32;
  }
  __device__ void load(const Params&, int kb, float* As, float* Bs, int tid) {
    {
      const int r = tid & 31, k0 = tid >> 5;
#pragma unroll
      for (int j = 0; j < BK / 8; ++j) {
        const int kk = k0 + j * 8;
        As[kk * T::LDA + r] = wd[(kb * BK + kk) * 32 + r];
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const int k = kb * BK + kk;
      const int oc = k >> 2, t = k & 3;
      const int koff = oc * C2_P - (t >> 1) * 9 - (t & 1);
      float v = 0.0f;
      if ((mask >> t) & 1u) v = dz[colbase + koff];
      Bs[kk * T::LDB + tid] = v;
    }
  }
  struct ColCtx {
    int64_t base;
    bool ok;
  };
  __device__ ColCtx col_ctx(const Params& p, int col) {
    const int c = blockIdx.x * T::BN + col;
    ColCtx x;
    x.ok = c < p.n * 100;
    const int b = c / 100, pq = c % 100;
    const int iy = 2 * (pq / 10) + (cls >> 1), ix = 2 * (pq % 10) + (cls & 1);
    x.base = e * p.out_es + (int64_t)b * (32 * C1_P) + iy * 20 + ix;
    return x;
  }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    if (!x.ok) return;
    const int64_t idx = x.base + row * C1_P;
    p.out[idx] = leaky_grad(p.act[idx], v);
  }
};

// ============================================================================================
//  FC forward: h[b][n] = sum_k a3[b][k] Wl[n][k] + bl[n]     rows = b, cols = n, red = k
// ============================================================================================
struct FcFwdOp {
  using T = Tile<128, 128>;
  struct Params {
    const float* a3;  // [e][n][3136]
    int64_t a3_es;
    const float* wlt;  // [e][3136][512]
    const float* params;
    int64_t bias_off[2];
    float* h;  // [e][n][512]
    int64_t h_es;
    int n;
  };
  int kb_begin, kb_end, e, b0, n0;
  const float* a3;
  const float* wlt;
  __device__ void init(const Params& p, int) {
    e = blockIdx.z;
    kb_begin = 0;
    kb_end = FLAT / BK;
    n0 = blockIdx.x * 128;
    b0 = blockIdx.y * 128;
    a3 = p.a3 + e * p.a3_es;
    wlt = p.wlt + (int64_t)e * FLAT * FEAT;
  }
  __device__ void load(const Params& p, int kb, float* As, float* Bs, int tid) {
    {
      const int kk = tid & 31, r0 = tid >> 5;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = r0 + j * 8;
        const int b = b0 + r;
        As[kk * T::LDA + r] = (b < p.n) ? a3[(int64_t)b * FLAT + kb * BK + kk] : 0.0f;
      }
    }
    {
      const int cc = tid & 127, k0 = tid >> 7;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int kk = k0 + 2 * j;
        Bs[kk * T::LDB + cc] = wlt[(int64_t)(kb * BK + kk) * FEAT + n0 + cc];
      }
    }
  }
  struct ColCtx {
    int n;
    float bias;
  };
  __device__ ColCtx col_ctx(const Params& p, int col) {
    ColCtx x;
    x.n = n0 + col;
    x.bias = p.params[p.bias_off[e] + x.n];
    return x;
  }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    const int b = b0 + row;
    if (b < p.n) p.h[e * p.h_es + (int64_t)b * FEAT + x.n] = v + x.bias;
  }
};

// ============================================================================================
//  FC weight gradient: part[s][e][n][k] = sum_{b in split} dh[b][n] a3[b][k]
//  rows = n (512), cols = k (3136), reduction = b
// ============================================================================================
struct FcWgradOp {
  using T = Tile<128, 128>;
  struct Params {
    const float* dh;  // [e][n][512]
    int64_t dh_es;
    const float* a3;
    int64_t a3_es;
    float* part;  // [nsplit][e][512][3136]
    int n, nsplit;
  };
  int kb_begin, kb_end, e, split, n0, k0;
  const float* dh;
  const float* a3;
  __device__ void init(const Params& p, int) {
    e = blockIdx.z % 2;
    split = blockIdx.z / 2;
    k0 = blockIdx.x * 128;
    n0 = blockIdx.y * 128;
    const int nkb = (p.n + BK - 1) / BK;
    const int per = (nkb + p.nsplit - 1) / p.nsplit;
    kb_begin = split * per;
    kb_end = min(nkb, kb_begin + per);
    dh = p.dh + e * p.dh_es;
    a3 = p.a3 + e * p.a3_es;
  }
  __device__ void load(const Params& p, int kb, float* As, float* Bs, int tid) {
    const int cc = tid & 127, h0 = tid >> 7;
    const bool kok = (k0 + cc) < FLAT;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int bb = h0 + 2 * j;
      const int b = kb * BK + bb;
      const bool ok = b < p.n;
      As[bb * T::LDA + cc] = ok ? dh[(int64_t)b * FEAT + n0 + cc] : 0.0f;
      Bs[bb * T::LDB + cc] = (ok && kok) ? a3[(int64_t)b * FLAT + k0 + cc] : 0.0f;
    }
  }
  struct ColCtx {
    int k;
  };
  __device__ ColCtx col_ctx(const Params&, int col) { return ColCtx{k0 + col}; }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    if (x.k < FLAT) p.part[(((int64_t)split * 2 + e) * FEAT + n0 + row) * FLAT + x.k] = v;
  }
};

// ============================================================================================
//  FC data gradient: dz3[b][k] = leaky'(a3[b][k]) * sum_n dh[b][n] Wl[n][k]
//  rows = b, cols = k (3136), reduction = n (512)
// ============================================================================================
struct FcDgradOp {
  using T = Tile<128, 128>;
  struct Params {
    const float* dh;
    int64_t dh_es;
    const float* params;  // Wl at lw_off[e], natural [512][3136]
    int64_t lw_off[2];
    const float* a3;
    float* dz3;
    int64_t a3_es;
    int n;
  };
  int kb_begin, kb_end, e, b0, k0;
  const float* dh;
  const float* wl;
  __device__ void init(const Params& p, int) {
    e = blockIdx.z;
    kb_begin = 0;
    kb_end = FEAT / BK;
    k0 = blockIdx.x * 128;
    b0 = blockIdx.y * 128;
    dh = p.dh + e * p.dh_es;
    wl = p.params + p.lw_off[e];
  }
  __device__ void load(const Params& p, int kb, float* As, float* Bs, int tid) {
    {
      const int nn = tid & 31, r0 = tid >> 5;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = r0 + j * 8;
        const int b = b0 + r;
        As[nn * T::LDA + r] = (b < p.n) ? dh[(int64_t)b * FEAT + kb * BK + nn] : 0.0f;
      }
    }
    {
      const int cc = tid & 127, h0 = tid >> 7;
      const bool kok = (k0 + cc) < FLAT;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int nn = h0 + 2 * j;
        Bs[nn * T::LDB + cc] = kok ? wl[(int64_t)(kb * BK + nn) * FLAT + k0 + cc] : 0.0f;
      }
    }
  }
  struct ColCtx {
    int k;
  };
  __device__ ColCtx col_ctx(const Params&, int col) { return ColCtx{k0 + col}; }
  __device__ void store(const Params& p, const ColCtx& x, int row, float v) {
    const int b = b0 + row;
    if (b < p.n && x.k < FLAT) {
      const int64_t idx = e * p.a3_es + (int64_t)b * FLAT + x.k;
      p.dz3[idx] = leaky_grad(p.a3[idx], v);
    }
  }
};

// ============================================================================================
//                                       host launchers
// ============================================================================================
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void launch_encoder_forward(const EncCall& c, bool acting, hipStream_t st) {
  launch_conv_forward2(c, st);
  launch_fc_forward2(c, acting, st);
}

// Backward of both encoders given dh[e][n][512]; leaves split-K partial slabs reduced into the
// grad arena by reduce_partials (optim.hip).
#define PROF(name, stmt)              \
  do {                                 \
    ProfRange _pr(c.prof, name, st);   \
    stmt;                              \
  } while (0)

void launch_encoder_backward(const EncCall& c, float* grads, hipStream_t st) {
  launch_fc_backward2(c, grads, st);
  launch_conv_wgrad3_2(c, grads, st);
  launch_conv_dgrad3_2(c, st);
  launch_conv_wgrad2_2(c, grads, st);
  launch_conv_dgrad2_2(c, st);
  launch_conv_wgrad1_2(c, grads, st);  // no data gradient for conv1: the frames are leaves
}

}  // namespace ddrl
