// Generic 2-D convolution (runtime geometry) as gather-type implicit GEMMs on the pipelined
// f32-MFMA engine: forward (+bias, +ReLU), data gradient and weight/bias gradient, plus the
// 2x2 max-pool pair.  These serve the encoders outside the Atari fast path -- NavPreNet /
// NavPedPreNet / NavPreNet1D (reference USTC_lab/nn/nav_encoder.py:18-20,27-32,96-101,108-122; a
// Conv1d is a Conv2d with H = KH = 1).  The Atari encoder keeps its direct-convolution kernels
// (conv2.hip, wgrad2.hip), which are 2-3x faster than a gather formulation.
//
//   forward   rows = cout (64/tile), cols = (b, oy, ox) (256/tile), k = (ci, ky, kx)
//   dgrad     rows = cin,            cols = (b, y, x),               k = (co, ky, kx)
//   wgrad     rows = cout,           cols = taps (ci, ky, kx),       k = (b, oy, ox), split over k
//
// A thread of the forward / dgrad kernels owns ONE column for the whole k loop, so the
// (b, y, x) decode happens once; the k -> (channel, ky, kx) decode is a table lookup (uniform).
// All gathers are unconditional loads from clamped addresses, masked when committed to LDS.
#include "engine2.h"
#include "ops.h"

namespace ddrl {

namespace gconv {

// k-block depth: 16 keeps two stages at 42 KB of LDS, i.e. 2-3 workgroups per CU (32 would be
// 83 KB -> one workgroup, one wave per SIMD, nothing to hide the gather latency behind)
constexpr int KBLK = 16;
constexpr int LDA = 68, LDB = 260;  // k-major tiles A[KBLK][64+4], B[KBLK][256+4]
constexpr int A_FLOATS = KBLK * LDA, B_FLOATS = KBLK * LDB;

struct Common {
  static constexpr int THREADS = 256, TM = 2, TN = 2, KSTEPS = KBLK / 2;
  static constexpr int A_OFF = 0, B_OFF = A_FLOATS, STAGE = A_FLOATS + B_FLOATS;
  int abase[2], bbase[2], kb_begin, kb_end;
  int l31, hi, wc;
  static constexpr int aoff(int s) { return 2 * s * LDA; }
  static constexpr int boff(int s) { return 2 * s * LDB; }
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void lanes(int tid) {
    const int lane = tid & 63;
    wc = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = A_OFF + hi * LDA + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = B_OFF + hi * LDB + wc * 64 + j * 32 + l31;
  }
};

// ktab[k] = {offset of (channel, ky, kx) inside a sample of the gathered tensor, (ky << 16) | kx};
// entries k >= K hold ky = 0x7fff (never in range -> zero)
//
// DGRAD = false: out[b][co][oy][ox] = act(bias[co] + sum W[co][ci][ky][kx] in[b][ci][oy*S+ky-ph][ox*S+kx-pw])
// DGRAD = true : din[b][ci][y][x]   = sum W[co][ci][ky][kx] dz[b][co][(y+ph-ky)/S][(x+pw-kx)/S]
template <bool DGRAD>
struct Gather : Common {
  struct Params {
    ConvGeom g;
    const float* src;   // gathered tensor (input for forward, dz for dgrad)
    const float* wp;    // [row tile][kb][32][64]
    const int2* ktab;   // [Kp32]
    const float* bias;  // forward only
    float* dst;
    int K;    // reduction length (cin*kh*kw or cout*kh*kw)
    int act;  // forward: 0 none, 1 relu
  };
  struct Regs {
    f4 a;
    float b[KBLK];
    unsigned ok;
  };
  int c0, r0;
  int y0, x0;        // forward: oy*S-ph, ox*S-pw ; dgrad: y+ph, x+pw
  int64_t colbase;   // sample base of this thread's column in src (+ spatial part for forward)
  bool colok;
  const float* wp;
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    lanes(tid);
    const ConvGeom& g = p.g;
    c0 = blockIdx.x * 256;
    r0 = blockIdx.y * 64;
    kb_begin = 0;
    kb_end = (p.K + KBLK - 1) / KBLK;
    wp = p.wp + (int64_t)blockIdx.y * kb_end * (KBLK * 64);
    const int P = DGRAD ? g.h * g.w : g.oh * g.ow;
    const int cw = DGRAD ? g.w : g.ow;
    const int64_t col = (int64_t)c0 + tid;
    colok = col < (int64_t)g.n * P;
    const int64_t cc = colok ? col : 0;
    const int b = (int)(cc / P), pix = (int)(cc % P);
    const int y = pix / cw, x = pix % cw;
    if (DGRAD) {
      y0 = y + g.pad_h;
      x0 = x + g.pad_w;
      colbase = (int64_t)b * g.out_sn;  // dz has the forward OUTPUT's layout
    } else {
      y0 = y * g.stride - g.pad_h;
      x0 = x * g.stride - g.pad_w;
      colbase = (int64_t)b * g.in_sn + (int64_t)y0 * g.w + x0;
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int tid = threadIdx.x;
    const ConvGeom& g = p.g;
    r.a = ld4(wp + (int64_t)kb * (KBLK * 64) + tid * 4);
    r.ok = 0;
#pragma unroll
    for (int j = 0; j < KBLK; ++j) {
      const int2 e = p.ktab[kb * KBLK + j];  // uniform
      const int ky = e.y >> 16, kx = e.y & 0xffff;
      bool ok;
      int64_t off;
      if (DGRAD) {
        const int ty = y0 - ky, tx = x0 - kx;
        const int oy = ty >> g.lgs, ox = tx >> g.lgs;
        ok = colok && ty >= 0 && tx >= 0 && ((ty | tx) & (g.stride - 1)) == 0 && oy < g.oh && ox < g.ow;
        off = colbase + e.x + oy * g.ow + ox;
      } else {
        ok = colok && (unsigned)(y0 + ky) < (unsigned)g.h && (unsigned)(x0 + kx) < (unsigned)g.w;
        off = colbase + e.x;
      }
      r.b[j] = p.src[ok ? off : 0];
      r.ok |= (ok ? 1u : 0u) << j;
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
    st4(buf + A_OFF + (tid >> 4) * LDA + (tid & 15) * 4, r.a);  // f4 index tid inside [KBLK][64]
#pragma unroll
    for (int j = 0; j < KBLK; ++j) buf[B_OFF + j * LDB + tid] = ((r.ok >> j) & 1u) ? r.b[j] : 0.0f;
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float*) {
    const ConvGeom& g = p.g;
    const int P = DGRAD ? g.h * g.w : g.oh * g.ow;
    const int nrows = DGRAD ? g.cin : g.cout;
    const int64_t dsn = DGRAD ? g.in_sn : g.out_sn;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = (int64_t)c0 + wc * 64 + j * 32 + l31;
      if (col >= (int64_t)g.n * P) continue;
      const int b = (int)(col / P), pix = (int)(col % P);
      float* dst = p.dst + (int64_t)b * dsn + pix;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + i * 32 + acc_row(r, hi);
          if (row < nrows) {
            float v = acc[i][j][r];
            if (!DGRAD) {
              v += p.bias[row];
              if (p.act == 1) v = fmaxf(v, 0.0f);
            }
            dst[(int64_t)row * P] = v;
          }
        }
    }
  }
};

// part[s][co][tap] = sum over the split's (b, oy, ox) of dz[b][co][oy][ox] * in[b][ci][oy*S+ky-ph][ox*S+kx-pw]
// followed by the bias partial [cout].  Requires oh*ow >= KBLK.
struct Wgrad : Common {
  static constexpr int LDAW = 65;  // A tile written along k by consecutive lanes: odd stride
  static constexpr int B_OFFW = KBLK * LDAW, STAGE = B_OFFW + B_FLOATS;
  struct Params {
    ConvGeom g;
    const float* in;
    const float* dz;
    const int* ptab;  // [oh*ow] = (oy << 16) | ox
    float* part;      // [nsplit][cout*KT + cout]
    int KT;           // taps = cin*kh*kw
    int nsplit;
  };
  struct Regs {
    float a[4], b[KBLK];
    unsigned oka, okb;
  };
  int r0, t0, split;
  // Staging roles: this thread owns pixel k = tid & 15 of every k-block (so the 16 lanes of a
  // group read 16 CONSECUTIVE pixels: coalesced) and the 16 taps (tid >> 4) + 16 j of the tile.
  int tapoff[KBLK];   // ci*H*W + (ky-ph)*W + (kx-pw) of tap j
  int tapyx[KBLK];    // ((ky-ph) << 16) | ((kx-pw) & 0xffff); ky = 0x4000 marks a tap >= KT
  float bacc[4];
  static constexpr int aoff(int s) { return 2 * s * LDAW; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    const ConvGeom& g = p.g;
    const int lane = tid & 63;
    wc = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = hi * LDAW + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) bbase[j] = B_OFFW + hi * LDB + wc * 64 + j * 32 + l31;
    t0 = blockIdx.x * 256;
    r0 = blockIdx.y * 64;
    split = blockIdx.z;
    const int64_t ktot = (int64_t)g.n * g.oh * g.ow;
    const int nkb = (int)((ktot + KBLK - 1) / KBLK);
    const int per = (nkb + p.nsplit - 1) / p.nsplit;
    kb_begin = min(nkb, split * per);
    kb_end = min(nkb, kb_begin + per);
    const int khw = g.kh * g.kw;
#pragma unroll
    for (int j = 0; j < KBLK; ++j) {
      const int tap = t0 + (tid >> 4) + 16 * j;
      const bool ok = tap < p.KT;
      const int tt = ok ? tap : 0;
      const int ci = tt / khw, rr = tt % khw;
      const int ty = rr / g.kw - g.pad_h, tx = rr % g.kw - g.pad_w;
      tapoff[j] = ci * g.h * g.w + ty * g.w + tx;
      tapyx[j] = ok ? ((ty << 16) | (tx & 0xffff)) : (0x4000 << 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bacc[j] = 0.0f;
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int tid = threadIdx.x;
    const ConvGeom& g = p.g;
    const int P = g.oh * g.ow;
    const int64_t k0 = (int64_t)kb * KBLK;
    const int bu = (int)(k0 / P), remu = (int)(k0 % P);  // uniform
    const int kk = tid & (KBLK - 1), rr = tid >> 4;
    int pix = remu + kk, b = bu;  // this thread's k = (sample, output pixel)
    if (pix >= P) {
      pix -= P;
      b += 1;
    }
    const bool kok = b < g.n;
    const int bc = kok ? b : 0;
    // ---- A: dz[b][row][pix], rows (tid >> 4) + 16 j ----
    {
      const int64_t base = (int64_t)bc * g.out_sn + pix;
      r.oka = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = r0 + rr + 16 * j;
        const bool ok = kok && row < g.cout;
        r.a[j] = p.dz[ok ? base + (int64_t)row * P : 0];
        r.oka |= (ok ? 1u : 0u) << j;
      }
    }
    // ---- B: in[b][ci][oy*S+ky-ph][ox*S+kx-pw] for this pixel and the thread's 16 taps ----
    {
      const int e = p.ptab[pix];
      const int iy = (e >> 16) * g.stride, ix = (e & 0xffff) * g.stride;
      const int64_t base = (int64_t)bc * g.in_sn + iy * g.w + ix;
      r.okb = 0;
#pragma unroll
      for (int j = 0; j < KBLK; ++j) {
        const int ty = tapyx[j] >> 16, tx = (int)(short)(tapyx[j] & 0xffff);
        const bool ok = kok && (unsigned)(iy + ty) < (unsigned)g.h && (unsigned)(ix + tx) < (unsigned)g.w;
        r.b[j] = p.in[ok ? base + tapoff[j] : 0];
        r.okb |= (ok ? 1u : 0u) << j;
      }
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
    const int kk = tid & (KBLK - 1), rr = tid >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = ((r.oka >> j) & 1u) ? r.a[j] : 0.0f;
      buf[kk * LDAW + rr + 16 * j] = v;
      bacc[j] += v;
    }
    // B[k][tap]: lanes differ in k (stride LDB = 260 -> bank 4k) and in tap (+1 per 16-lane group): no conflicts
#pragma unroll
    for (int j = 0; j < KBLK; ++j) buf[B_OFFW + kk * LDB + rr + 16 * j] = ((r.okb >> j) & 1u) ? r.b[j] : 0.0f;
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float*) {
    const ConvGeom& g = p.g;
    float* slab = p.part + (int64_t)split * ((int64_t)g.cout * p.KT + g.cout);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tap = t0 + wc * 64 + j * 32 + l31;
      if (tap >= p.KT) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + i * 32 + acc_row(r, hi);
          if (row < g.cout) slab[(int64_t)row * p.KT + tap] = acc[i][j][r];
        }
    }
    if (blockIdx.x == 0) {  // bias partial: sum the 16 k lanes that share this thread's 4 rows
      const int rr = threadIdx.x >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = bacc[j];
#pragma unroll
        for (int off = KBLK / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        const int row = r0 + rr + 16 * j;
        if ((threadIdx.x & (KBLK - 1)) == 0 && row < g.cout) slab[(int64_t)g.cout * p.KT + row] = s;
      }
    }
  }
};

}  // namespace gconv

// wp[row tile][kb][kk < KBLK][64]: forward  A(row = co, k = (ci,ky,kx)) = W[co][ci][ky][kx]
//                           dgrad    A(row = ci, k = (co,ky,kx)) = W[co][ci][ky][kx]
// ktab: forward  {ci*H*W + ky*W + kx, ky<<16|kx};  dgrad {co*OH*OW, ky<<16|kx}
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, ConvGeom g, float* __restrict__ wpf,
                                                        int2* __restrict__ ktf, float* __restrict__ wpd, int2* __restrict__ ktd) {
  const int khw = g.kh * g.kw;
  const int Kf = g.cin * khw, Kd = g.cout * khw;
  constexpr int KB = gconv::KBLK;
  const int kbf = (Kf + KB - 1) / KB, kbd = (Kd + KB - 1) / KB;
  const int rtf = (g.cout + 63) / 64, rtd = (g.cin + 63) / 64;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)rtf * kbf * (KB * 64)) {
    const int row = (int)(i & 63), k = (int)((i >> 6) % (kbf * KB)), rt = (int)((i >> 6) / (kbf * KB));
    const int co = rt * 64 + row;
    wpf[i] = (co < g.cout && k < Kf) ? w[(int64_t)co * Kf + k] : 0.0f;
  }
  if (i < (int64_t)rtd * kbd * (KB * 64)) {
    const int row = (int)(i & 63), k = (int)((i >> 6) % (kbd * KB)), rt = (int)((i >> 6) / (kbd * KB));
    const int ci = rt * 64 + row;
    const int co = k / khw, r = k % khw;
    wpd[i] = (ci < g.cin && k < Kd) ? w[((int64_t)co * g.cin + ci) * khw + r] : 0.0f;
  }
  if (i < kbf * KB) {
    const int k = (int)i, ci = k / khw, r = k % khw, ky = r / g.kw, kx = r % g.kw;
    ktf[i] = k < Kf ? make_int2(ci * g.h * g.w + ky * g.w + kx, (ky << 16) | kx) : make_int2(0, 0x7fff << 16);
  }
  if (i < kbd * KB) {
    const int k = (int)i, co = k / khw, r = k % khw, ky = r / g.kw, kx = r % g.kw;
    ktd[i] = k < Kd ? make_int2(co * g.oh * g.ow, (ky << 16) | kx) : make_int2(0, 0x7fff << 16);
  }
}

__global__ __launch_bounds__(256) void conv_ptab_kernel(int ow, int P, int* __restrict__ ptab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < P) ptab[i] = ((i / ow) << 16) | (i % ow);
}

// 2x2 / stride 2 max pool over [planes][H][W] (H, W even) -- F.max_pool2d(x, 2, stride=2).
// Rows 2y and 2y+1 of a plane are adjacent in memory and H is even, so the tensor is a flat list of ROW PAIRS of 2W floats, each
// producing one output row of W/2 windows: window i sits in pair i / ow, and that one division (32-bit whenever the tensor allows) is
// all the index arithmetic there is.  Every access is 8 bytes per lane (W even keeps them aligned), contiguous across the wave.
template <typename IT>
struct PoolAt {
  IT i, in_off;
  __device__ PoolAt(IT idx, int W) : i(idx) {
    const IT ow = (IT)(W / 2), rp = idx / ow, x = idx - rp * ow;
    in_off = rp * (IT)(2 * W) + 2 * x;
  }
};

__device__ __forceinline__ int first_max(float v0, float v1, float v2, float v3, float& m) {  // PyTorch's scan order; ties keep the first
  int am = 0;
  m = v0;
  if (v1 > m) { m = v1; am = 1; }
  if (v2 > m) { m = v2; am = 2; }
  if (v3 > m) { m = v3; am = 3; }
  return am;
}

template <typename IT>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ in, IT total, int W, float* __restrict__ out) {
  const IT i = (IT)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const PoolAt<IT> at(i, W);
  const float2 r0 = *(const float2*)(in + at.in_off), r1 = *(const float2*)(in + at.in_off + W);
  out[i] = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
}

// The same pool, also leaving the DECISIONS its backward needs in one byte per window: bits 0-1 = position of the FIRST maximum
// (row-major scan, as PyTorch routes the gradient), bit 2 = that maximum is positive (the ReLU in front of the pool lets the gradient
// through).  The backward then reads dpool and this byte instead of the full-resolution activations: 1.31 instead of 2.25 tensor sizes.
template <typename IT>
__global__ __launch_bounds__(256) void maxpool2_fwd_idx_kernel(const float* __restrict__ in, IT total, int W, float* __restrict__ out,
                                                               uint8_t* __restrict__ code) {
  const IT i = (IT)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const PoolAt<IT> at(i, W);
  const float2 r0 = *(const float2*)(in + at.in_off), r1 = *(const float2*)(in + at.in_off + W);
  float m;
  const int am = first_max(r0.x, r0.y, r1.x, r1.y, m);
  out[i] = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));  // the value exactly as maxpool2_fwd_kernel forms it
  code[i] = (uint8_t)(am | (m > 0.0f ? 4 : 0));
}

// One thread per 16 bytes of dz (H*W is a multiple of 4, so the flat tensor is whole float4s): the kernel is bound by its 4-byte-per-
// element WRITE, and 1 KiB contiguous per wave is what the write path wants (one thread per window stores two half-used 128-byte lines per
// instruction and measured 2.7 TB/s).  W even makes elements (0,1) and (2,3) of a quad two whole window halves; the byte and the pooled
// gradient of a window are read by the two threads that own its rows, the second time from cache.
#ifndef DDRL_POOL_QUADS
#define DDRL_POOL_QUADS 4
#endif
template <typename IT>
__global__ __launch_bounds__(256) void maxpool2_bwd_idx_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ code, IT total4,
                                                               int W, float* __restrict__ dz) {
  const IT ow = (IT)(W / 2);
  IT e[DDRL_POOL_QUADS];
  int ca[DDRL_POOL_QUADS], cb[DDRL_POOL_QUADS], sa[DDRL_POOL_QUADS], sb[DDRL_POOL_QUADS];
  float ga[DDRL_POOL_QUADS], gb[DDRL_POOL_QUADS];
  bool live[DDRL_POOL_QUADS];
  // every load of the thread's quads is issued before the first store: a wave's stores wait on its loads, and one quad per wave
  // lifetime leaves too few bytes in flight to fill the write path
#pragma unroll
  for (int k = 0; k < DDRL_POOL_QUADS; ++k) {
    IT q = ((IT)blockIdx.x * DDRL_POOL_QUADS + k) * 256 + threadIdx.x;
    live[k] = q < total4;
    q = live[k] ? q : total4 - 1;
    e[k] = q * 4;
    const IT row = e[k] / (IT)W, col = e[k] - row * (IT)W;
    IT row2 = row, col2 = col + 2;
    if (col2 >= (IT)W) { col2 = 0; row2 = row + 1; }
    const IT wa = (row >> 1) * ow + (col >> 1), wb = (row2 >> 1) * ow + (col2 >> 1);
    ca[k] = code[wa];
    cb[k] = code[wb];
    ga[k] = dpool[wa];
    gb[k] = dpool[wb];
    sa[k] = (int)(row & 1) * 2;  // which half of the window this row is
    sb[k] = (int)(row2 & 1) * 2;
  }
#pragma unroll
  for (int k = 0; k < DDRL_POOL_QUADS; ++k) {
    const float xa = (ca[k] & 4) ? ga[k] : 0.0f, xb = (cb[k] & 4) ? gb[k] : 0.0f;
    if (live[k])
      *(float4*)(dz + e[k]) = make_float4((ca[k] & 3) == sa[k] ? xa : 0.0f, (ca[k] & 3) == sa[k] + 1 ? xa : 0.0f,
                                          (cb[k] & 3) == sb[k] ? xb : 0.0f, (cb[k] & 3) == sb[k] + 1 ? xb : 0.0f);
  }
}

// W a multiple of 4 (the usual case): one thread per PAIR of windows of a row pair -- one 2-byte and one 8-byte load (both aligned, since
// an even number of windows per row keeps pair p at pooled offset 2p) for two 16-byte stores, a quarter of the loads per byte of the
// kernel above.  The two stores of a wave land in the two rows of the same pairs, so together they cover whole lines.
template <typename IT>
__global__ __launch_bounds__(256) void maxpool2_bwd_idx_pairs_kernel(const float* __restrict__ dpool, const uint8_t* __restrict__ code,
                                                                     IT pairs, int W, float* __restrict__ dz) {
  const IT pw = (IT)(W / 4);
  IT off[DDRL_POOL_QUADS];
  unsigned c[DDRL_POOL_QUADS];
  float2 g[DDRL_POOL_QUADS];
  bool live[DDRL_POOL_QUADS];
#pragma unroll
  for (int k = 0; k < DDRL_POOL_QUADS; ++k) {
    IT p = ((IT)blockIdx.x * DDRL_POOL_QUADS + k) * 256 + threadIdx.x;
    live[k] = p < pairs;
    p = live[k] ? p : pairs - 1;
    const IT rp = p / pw, xp = p - rp * pw;
    off[k] = rp * (IT)(2 * W) + 4 * xp;
    c[k] = *(const uint16_t*)(code + 2 * p);
    g[k] = *(const float2*)(dpool + 2 * p);
  }
#pragma unroll
  for (int k = 0; k < DDRL_POOL_QUADS; ++k) {
    const unsigned c0 = c[k] & 255u, c1 = c[k] >> 8;
    const float x0 = (c0 & 4) ? g[k].x : 0.0f, x1 = (c1 & 4) ? g[k].y : 0.0f;
    if (live[k]) {
      *(float4*)(dz + off[k]) = make_float4((c0 & 3) == 0 ? x0 : 0.0f, (c0 & 3) == 1 ? x0 : 0.0f, (c1 & 3) == 0 ? x1 : 0.0f,
                                            (c1 & 3) == 1 ? x1 : 0.0f);
      *(float4*)(dz + off[k] + W) = make_float4((c0 & 3) == 2 ? x0 : 0.0f, (c0 & 3) == 3 ? x0 : 0.0f, (c1 & 3) == 2 ? x1 : 0.0f,
                                                (c1 & 3) == 3 ? x1 : 0.0f);
    }
  }
}

// d(pre-activation of the conv) from d(pooled): gradient goes to the FIRST maximum of each 2x2 window (PyTorch's scan order) and only
// where the ReLU in front of the pool was active (max > 0).  `a` = relu output (pre-pool), dz gets every element written.
template <typename IT>
__global__ __launch_bounds__(256) void maxpool2_relu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dpool, IT total,
                                                                int W, float* __restrict__ dz) {
  const IT i = (IT)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const PoolAt<IT> at(i, W);
  const float2 r0 = *(const float2*)(a + at.in_off), r1 = *(const float2*)(a + at.in_off + W);
  float m;
  const int am = first_max(r0.x, r0.y, r1.x, r1.y, m);
  const float g = (m > 0.0f) ? dpool[i] : 0.0f;
  *(float2*)(dz + at.in_off) = make_float2(am == 0 ? g : 0.0f, am == 1 ? g : 0.0f);
  *(float2*)(dz + at.in_off + W) = make_float2(am == 2 ? g : 0.0f, am == 3 ? g : 0.0f);
}

// ---- host side --------------------------------------------------------------------------------

static int ilog2(int s) { return s == 1 ? 0 : (s == 2 ? 1 : (s == 4 ? 2 : -1)); }

bool conv_geom_fill(ConvGeom& g) {
  g.lgs = ilog2(g.stride);
  if (g.lgs < 0 || g.n < 1 || g.cin < 1 || g.cout < 1 || g.kh < 1 || g.kw < 1 || g.kh > 255 || g.kw > 255) return false;
  g.oh = (g.h + 2 * g.pad_h - g.kh) / g.stride + 1;
  g.ow = (g.w + 2 * g.pad_w - g.kw) / g.stride + 1;
  if (g.oh < 1 || g.ow < 1 || g.oh > 32767 || g.ow > 32767) return false;
  if ((int64_t)g.cin * g.h * g.w >= (int64_t)1 << 30 || (int64_t)g.cout * g.oh * g.ow >= (int64_t)1 << 30) return false;
  return true;
}

// float counts of the packed buffers: wpf, ktf (int2 = 2 floats each), wpd, ktd, ptab
void conv_pack_sizes(const ConvGeom& g, int64_t out[5]) {
  const int khw = g.kh * g.kw;
  constexpr int KB = gconv::KBLK;
  const int kbf = (g.cin * khw + KB - 1) / KB, kbd = (g.cout * khw + KB - 1) / KB;
  out[0] = (int64_t)((g.cout + 63) / 64) * kbf * (KB * 64);
  out[1] = (int64_t)kbf * KB * 2;
  out[2] = (int64_t)((g.cin + 63) / 64) * kbd * (KB * 64);
  out[3] = (int64_t)kbd * KB * 2;
  out[4] = (int64_t)g.oh * g.ow;
}

void launch_conv_pack(const ConvGeom& g, const float* w, float* wpf, int2* ktf, float* wpd, int2* ktd, int* ptab, hipStream_t st) {
  int64_t sz[5];
  conv_pack_sizes(g, sz);
  const int64_t total = sz[0] > sz[2] ? sz[0] : sz[2];
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, g, wpf, ktf, wpd, ktd);
  hipLaunchKernelGGL(conv_ptab_kernel, dim3((unsigned)((sz[4] + 255) / 256)), dim3(256), 0, st, g.ow, (int)sz[4], ptab);
}

void launch_conv_fwd(const ConvGeom& g, const float* in, const float* wpf, const int2* ktf, const float* bias, int act,
                     float* out, hipStream_t st) {
  gconv::Gather<false>::Params p{g, in, wpf, ktf, bias, out, g.cin * g.kh * g.kw, act};
  const int64_t cols = (int64_t)g.n * g.oh * g.ow;
  launch_engine2<gconv::Gather<false>>(dim3((unsigned)((cols + 255) / 256), (g.cout + 63) / 64, 1), p, st);
}

void launch_conv_dgrad(const ConvGeom& g, const float* dz, const float* wpd, const int2* ktd, float* din, hipStream_t st) {
  gconv::Gather<true>::Params p{g, dz, wpd, ktd, nullptr, din, g.cout * g.kh * g.kw, 0};
  const int64_t cols = (int64_t)g.n * g.h * g.w;
  launch_engine2<gconv::Gather<true>>(dim3((unsigned)((cols + 255) / 256), (g.cin + 63) / 64, 1), p, st);
}

bool conv_is_thin(const ConvGeom& g);
int conv_thin_wgs(const ConvGeom& g);

int conv_wgrad_splits(const ConvGeom& g) {
  if (conv_is_thin(g)) return conv_thin_wgs(g);  // one slab per workgroup of thin_wgrad_kernel
  const int KT = g.cin * g.kh * g.kw;
  const int tiles = ((KT + 255) / 256) * ((g.cout + 63) / 64);
  int s = (768 + tiles - 1) / tiles;
  const int64_t nkb = ((int64_t)g.n * g.oh * g.ow + gconv::KBLK - 1) / gconv::KBLK;
  const int64_t cap = (nkb + 7) / 8;  // at least 8 k-blocks per split
  if (s > cap) s = (int)cap;
  return s < 1 ? 1 : s;
}

// "Thin" weight gradient: a Conv1d with a handful of taps and few output channels (NavPreNet1D.conv1d1:
// 1 -> 32 channels, 5 taps) is a reduction over (b, x), not a GEMM -- 160 sums fed by 60 KB of dz per
// sample.  Lanes run along x, so the dz rows and the input taps are read coalesced; every thread keeps
// all cout x KT partial sums in registers over the samples its workgroup owns; one shuffle + LDS
// reduction at the end, then the usual slab reduction.
constexpr int THIN_MAX_TAPS = 6, THIN_COUT = 32, THIN_WGS = 512;
template <int KT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ConvGeom g, const float* __restrict__ in, const float* __restrict__ dz,
                                                         float* __restrict__ part) {
  __shared__ float red[4][THIN_COUT * (KT + 1)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[THIN_COUT][KT], bias[THIN_COUT];
#pragma unroll
  for (int c = 0; c < THIN_COUT; ++c) {
    bias[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[c][k] = 0.0f;
  }
  for (int b = blockIdx.x; b < g.n; b += gridDim.x) {
    const float* dzb = dz + (int64_t)b * g.out_sn;
    const float* inb = in + (int64_t)b * g.in_sn;
    for (int x = threadIdx.x; x < g.ow; x += 256) {
      float v[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int ci = k / g.kw, ix = x * g.stride + (k % g.kw) - g.pad_w;
        v[k] = inb[ci * g.w + min(max(ix, 0), g.w - 1)];
        if (ix < 0 || ix >= g.w) v[k] = 0.0f;
      }
#pragma unroll
      for (int c = 0; c < THIN_COUT; ++c) {
        const float d = (c < g.cout) ? dzb[min(c, g.cout - 1) * g.ow + x] : 0.0f;
        bias[c] += d;
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[c][k] = __builtin_fmaf(d, v[k], acc[c][k]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < THIN_COUT; ++c) {
#pragma unroll
    for (int k = 0; k <= KT; ++k) {
      float s = (k < KT) ? acc[c][k < KT ? k : 0] : bias[c];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) red[wave][c * (KT + 1) + k] = s;
    }
  }
  __syncthreads();
  float* slab = part + (int64_t)blockIdx.x * ((int64_t)g.cout * KT + g.cout);
  for (int i = threadIdx.x; i < g.cout * (KT + 1); i += 256) {
    const int c = i / (KT + 1), k = i % (KT + 1);
    const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (k == KT) slab[(int64_t)g.cout * KT + c] = s;
    else slab[c * KT + k] = s;
  }
}

bool conv_is_thin(const ConvGeom& g) {
  return g.kh == 1 && g.h == 1 && g.oh == 1 && g.cin * g.kw <= THIN_MAX_TAPS && g.cout <= THIN_COUT && g.pad_h == 0;
}
int conv_thin_wgs(const ConvGeom& g) { return g.n < THIN_WGS ? g.n : THIN_WGS; }

void launch_conv_wgrad(const ConvGeom& g, const float* in, const float* dz, const int* ptab, float* part, float* dw, float* db,
                       hipStream_t st) {
  if (conv_is_thin(g)) {
    const int KT = g.cin * g.kw, W = conv_thin_wgs(g);
    switch (KT) {
      case 1: hipLaunchKernelGGL(thin_wgrad_kernel<1>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
      case 2: hipLaunchKernelGGL(thin_wgrad_kernel<2>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
      case 3: hipLaunchKernelGGL(thin_wgrad_kernel<3>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
      case 4: hipLaunchKernelGGL(thin_wgrad_kernel<4>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
      case 5: hipLaunchKernelGGL(thin_wgrad_kernel<5>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
      default: hipLaunchKernelGGL(thin_wgrad_kernel<6>, dim3(W), dim3(256), 0, st, g, in, dz, part); break;
    }
    const int64_t slab = (int64_t)g.cout * KT + g.cout;
    launch_reduce_slabs2(part, W, slab, (int64_t)g.cout * KT, dw, g.cout, db, st);
    return;
  }
  const int KT = g.cin * g.kh * g.kw;
  const int S = conv_wgrad_splits(g);
  gconv::Wgrad::Params p{g, in, dz, ptab, part, KT, S};
  launch_engine2<gconv::Wgrad>(dim3((KT + 255) / 256, (g.cout + 63) / 64, S), p, st);
  const int64_t slab = (int64_t)g.cout * KT + g.cout;
  launch_reduce_slabs2(part, S, slab, (int64_t)g.cout * KT, dw, g.cout, db, st);
}

// 32-bit index arithmetic whenever every float offset of the full-resolution tensor fits (the usual case: < 16 GiB)
#define DDRL_POOL_LAUNCH(KERNEL, ...)                                                                               \
  do {                                                                                                              \
    const int64_t total = planes * (H / 2) * (W / 2);                                                               \
    const dim3 grid((unsigned)((total + 255) / 256));                                                               \
    if (total * 4 < ((int64_t)1 << 32) - 1024)                                                                      \
      hipLaunchKernelGGL(KERNEL<uint32_t>, grid, dim3(256), 0, st, __VA_ARGS__);                                    \
    else                                                                                                            \
      hipLaunchKernelGGL(KERNEL<int64_t>, grid, dim3(256), 0, st, __VA_ARGS__);                                     \
  } while (0)

void launch_maxpool2_fwd(const float* in, int64_t planes, int H, int W, float* out, hipStream_t st) {
  DDRL_POOL_LAUNCH(maxpool2_fwd_kernel, in, total, W, out);
}

void launch_maxpool2_fwd_idx(const float* in, int64_t planes, int H, int W, float* out, uint8_t* code, hipStream_t st) {
  DDRL_POOL_LAUNCH(maxpool2_fwd_idx_kernel, in, total, W, out, code);
}

void launch_maxpool2_bwd_idx(const float* dpool, const uint8_t* code, int64_t planes, int H, int W, float* dz, hipStream_t st) {
  const int64_t total4 = planes * H * W / 4;  // H, W even
  if (W % 4 == 0 && !(((uintptr_t)dpool & 7) | ((uintptr_t)code & 1))) {
    const int64_t pairs = total4 / 2;
    const dim3 pgrid((unsigned)((pairs + 256 * DDRL_POOL_QUADS - 1) / (256 * DDRL_POOL_QUADS)));
    if (total4 * 4 < ((int64_t)1 << 32) - 1024)
      hipLaunchKernelGGL(maxpool2_bwd_idx_pairs_kernel<uint32_t>, pgrid, dim3(256), 0, st, dpool, code, (uint32_t)pairs, W, dz);
    else
      hipLaunchKernelGGL(maxpool2_bwd_idx_pairs_kernel<int64_t>, pgrid, dim3(256), 0, st, dpool, code, pairs, W, dz);
    return;
  }
  const dim3 grid((unsigned)((total4 + 256 * DDRL_POOL_QUADS - 1) / (256 * DDRL_POOL_QUADS)));
  if (total4 * 4 < ((int64_t)1 << 32) - 1024)
    hipLaunchKernelGGL(maxpool2_bwd_idx_kernel<uint32_t>, grid, dim3(256), 0, st, dpool, code, (uint32_t)total4, W, dz);
  else
    hipLaunchKernelGGL(maxpool2_bwd_idx_kernel<int64_t>, grid, dim3(256), 0, st, dpool, code, total4, W, dz);
}

void launch_maxpool2_relu_bwd(const float* a, const float* dpool, int64_t planes, int H, int W, float* dz, hipStream_t st) {
  DDRL_POOL_LAUNCH(maxpool2_relu_bwd_kernel, a, dpool, total, W, dz);
}

}  // namespace ddrl
