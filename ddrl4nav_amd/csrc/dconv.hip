// Direct stride-1 convolution from zero-padded LDS images with COMPILE-TIME geometry, on the
// pipelined f32-MFMA engine -- the design of the Atari conv3 / conv3-dgrad kernels (conv2.hip)
// as a template.  ddrl_op_conv_forward / ddrl_op_conv_dgrad dispatch here when a layer's geometry
// matches one of the instantiations (the heavy nav-encoder layers, reference
// USTC_lab/nn/nav_encoder.py:97-98); everything else goes through the gather kernels (gconv.hip).
//
//   out[b][r][oy][ox] = sum_{c, ky, kx} Wk[r][c][ky][kx] * in[b][c][oy + ky - PAD][ox + kx - PAD]
//
// rows = r (64 per workgroup), cols = (b, oy, ox) (256 per workgroup), k-block = CPB input
// channels x KS x KS taps.  The raw planes of the samples a column tile touches are copied once per
// k-block (16-byte coalesced loads) into padded LDS planes whose border stays zero, so that every
// MFMA B operand is one ds_read_b32 at lane_base + immediate; the two k indices of an MFMA (lane
// halves) are two adjacent input channels.  The data gradient of a stride-1 convolution is the
// same computation on dz with flipped, transposed kernels and PAD' = KS - 1 - PAD.
#include "engine2.h"
#include "ops.h"

namespace ddrl {

namespace dconv {

template <int CIN, int COUT, int KS, int HIN, int PAD, int CPB>
struct Direct {
  static_assert(CIN % CPB == 0 && CPB % 2 == 0, "channels per k-block: even divisor of CIN");
  static constexpr int THREADS = 256, TM = 2, TN = 2;
  static constexpr int KK = KS * KS, KSTEPS = (CPB / 2) * KK;
  static constexpr int OH = HIN + 2 * PAD - KS + 1, P = OH * OH;
  static constexpr int LP = HIN + 2 * PAD, PLANE = LP * LP, RAW = HIN * HIN;
  static constexpr int NS = (256 + P - 2) / P + 1;  // samples a 256-column tile can touch
  static constexpr int W_FLOATS = KSTEPS * 2 * 64, IMG_OFF = W_FLOATS;
  static constexpr int IMG_FLOATS = (NS * CPB * PLANE + 3) / 4 * 4;
  static constexpr int STAGE = W_FLOATS + IMG_FLOATS;
  static constexpr int NW4 = W_FLOATS / 4, NWJ = (NW4 + 255) / 256;                // weight f4 per k-block / per thread
  static_assert((CPB * RAW) % 4 == 0, "a sample's k-block of planes must be whole f4s");
  static constexpr int SB4 = CPB * RAW / 4, NI4 = NS * SB4, NIJ = (NI4 + 255) / 256;  // image f4 per sample / tile / thread
  static constexpr int NKB = CIN / CPB;
  struct Params {
    const float* in;
    int64_t in_sn;
    const float* wp;    // [row tile][NKB][KSTEPS][2][64]
    const float* bias;  // may be null (data gradient)
    float* out;
    int64_t out_sn;
    int n, act;
  };
  struct Regs {
    f4 w[NWJ], im[NIJ];
  };
  int abase[2], bbase[2], kb_begin, kb_end;
  int c0, r0, b_first, l31, hi, wc;
  int64_t imoff[NIJ];
  const float* wp;
  static constexpr int aoff(int s) { return 2 * s * 64; }
  static constexpr int boff(int s) { return (s / KK) * 2 * PLANE + ((s % KK) / KS) * LP + (s % KK) % KS; }
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void init(const Params& p, int tid, float* lds) {
    const int lane = tid & 63;
    wc = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
    c0 = blockIdx.x * 256;
    r0 = blockIdx.y * 64;
    b_first = c0 / P;
    kb_begin = 0;
    kb_end = NKB;
    wp = p.wp + (int64_t)blockIdx.y * NKB * W_FLOATS;
    if (PAD > 0) {  // the borders of both stages' images stay zero for the whole kernel
      for (int i = tid; i < IMG_FLOATS; i += 256) {
        lds[IMG_OFF + i] = 0.0f;
        lds[STAGE + IMG_OFF + i] = 0.0f;
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NIJ; ++j) {  // unconditional, clamped loads (a guarded load costs a branch + wait each)
      const int idx = tid + 256 * j;
      const bool has = idx < NI4;
      const int b = min(b_first + (has ? idx / SB4 : 0), p.n - 1);
      imoff[j] = (int64_t)b * p.in_sn + (has ? (idx % SB4) * 4 : 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = hi * 64 + i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t c = (int64_t)c0 + wc * 64 + j * 32 + l31;
      if (c >= (int64_t)p.n * P) c = c0;
      const int b = (int)(c / P), pix = (int)(c % P);
      bbase[j] = IMG_OFF + (b - b_first) * CPB * PLANE + (pix / OH) * LP + (pix % OH) + hi * PLANE;
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NWJ; ++j) r.w[j] = ld4(wp + kb * W_FLOATS + min(tid + 256 * j, NW4 - 1) * 4);
#pragma unroll
    for (int j = 0; j < NIJ; ++j) r.im[j] = ld4(p.in + imoff[j] + kb * (CPB * RAW));
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NW4) st4(buf + idx * 4, r.w[j]);
    }
#pragma unroll
    for (int j = 0; j < NIJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NI4) {
        const int bl = idx / SB4, q = idx % SB4;
        if (PAD == 0) {
          st4(buf + IMG_OFF + bl * CPB * PLANE + q * 4, r.im[j]);
        } else {
          const float v[4] = {r.im[j].x, r.im[j].y, r.im[j].z, r.im[j].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int el = q * 4 + i;
            const int cc = el / RAW, rr = el % RAW;
            buf[IMG_OFF + (bl * CPB + cc) * PLANE + (rr / HIN + PAD) * LP + (rr % HIN) + PAD] = v[i];
          }
        }
      }
    }
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float*) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t c = (int64_t)c0 + wc * 64 + j * 32 + l31;
      if (c >= (int64_t)p.n * P) continue;
      const int b = (int)(c / P), pix = (int)(c % P);
      float* dst = p.out + (int64_t)b * p.out_sn + pix;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + i * 32 + acc_row(r, hi);
          float v = acc[i][j][r];
          if (p.bias != nullptr) v += p.bias[row];
          if (p.act == 1) v = fmaxf(v, 0.0f);
          dst[(int64_t)row * P] = v;
        }
    }
  }
};

// wp[rt][kb][s][hi][row] for the forward (flip = 0):  W[co = rt*64+row][ci = kb*CPB + 2*(s/KK) + hi][tap = s%KK]
//                        for the data gradient (flip = 1): rows = ci, k-channels = co, tap flipped:
//                        W[co = kb*CPB + 2*(s/KK) + hi][ci = rt*64+row][KK-1 - s%KK]
__global__ __launch_bounds__(256) void direct_pack_kernel(const float* __restrict__ w, int cin, int cout, int kk, int cpb,
                                                          int flip, float* __restrict__ wp) {
  const int rows_total = flip ? cin : cout, kch = flip ? cout : cin;
  const int ksteps = (cpb / 2) * kk, nkb = kch / cpb;
  const int64_t total = (int64_t)(rows_total / 64) * nkb * ksteps * 128;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int row = (int)(i & 63), hi = (int)((i >> 6) & 1);
  int64_t r = i >> 7;
  const int s = (int)(r % ksteps);
  r /= ksteps;
  const int kb = (int)(r % nkb), rt = (int)(r / nkb);
  const int ch = kb * cpb + 2 * (s / kk) + hi, tap = s % kk;
  const int rr = rt * 64 + row;
  wp[i] = flip ? w[((int64_t)ch * cin + rr) * kk + (kk - 1 - tap)] : w[((int64_t)rr * cin + ch) * kk + tap];
}

// (Since round 4 this layer runs on fconv.hip's fp16-plane kernels; what follows is the f32-input form kept for DDRL_NAV_F32=1 / DDRL_FIRST_F32=1.)
// Forward for layers with few input channels (NavPreNet1D.conv1: 3 -> 64, 7x7, 48 -> 44), where the
// channel-pair MFMA k of `Direct` does not apply: the two k indices of an MFMA are the taps (kx, kx+1)
// of one kernel row (KS is padded to an even width with a zero weight), one k-block per input channel.
// Column tile = a band of R output rows of ONE sample; only the R + KS - 1 input rows the band
// touches are staged (8-byte loads), rows outside the image as zeros.
template <int CIN, int COUT, int KS, int HIN, int PAD, int R, int TNV>
struct DirectBandFwd {
  static constexpr int THREADS = 256, TM = 1, TN = TNV;
  static constexpr int KXP = (KS + 1) / 2, KSTEPS = KS * KXP;  // k-steps per input channel
  static constexpr int OH = HIN + 2 * PAD - KS + 1, P = OH * OH, RAW = HIN * HIN, LP = HIN + 2 * PAD + 1;
  static_assert(OH % R == 0 && HIN % 2 == 0 && COUT == 64, "band geometry");
  static constexpr int NB = OH / R, COLS = R * OH, BR = R + KS - 1;
  static_assert(COLS <= 2 * TN * 32, "column tile");
  static constexpr int W_FLOATS = KSTEPS * 2 * 64, IMG_OFF = W_FLOATS, IMG_FLOATS = (BR * LP + 3) / 4 * 4;
  static constexpr int STAGE = W_FLOATS + IMG_FLOATS;
  static constexpr int NW4 = W_FLOATS / 4, NWJ = (NW4 + 255) / 256;
  static constexpr int H2 = HIN / 2, NI2 = BR * H2, NIJ = (NI2 + 255) / 256;
  struct Params {
    const float* in;
    int64_t in_sn;
    const float* wp;  // [CIN][KSTEPS][2][64]
    const float* bias;
    float* out;
    int64_t out_sn;
    int n, act;
  };
  struct Regs {
    f4 w[NWJ];
    float2 im[NIJ];
  };
  int abase[1], bbase[TN], kb_begin, kb_end;
  int b, band, l31, hi, wr, wc;
  static constexpr int aoff(int s) { return 2 * s * 64; }
  static constexpr int boff(int s) { return (s / KXP) * LP + 2 * (s % KXP); }
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void init(const Params& p, int tid, float* lds) {
    const int lane = tid & 63, wave = tid >> 6;
    wr = wave >> 1;
    wc = wave & 1;
    l31 = lane & 31;
    hi = lane >> 5;
    band = blockIdx.x;
    b = blockIdx.y;
    kb_begin = 0;
    kb_end = CIN;
    for (int i = tid; i < IMG_FLOATS; i += 256) {  // PAD columns stay zero
      lds[IMG_OFF + i] = 0.0f;
      lds[STAGE + IMG_OFF + i] = 0.0f;
    }
    __syncthreads();
    abase[0] = hi * 64 + wr * 32 + l31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = min(wc * (TN * 32) + j * 32 + l31, COLS - 1);
      bbase[j] = IMG_OFF + (c / OH) * LP + (c % OH) + hi;
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NWJ; ++j) r.w[j] = ld4(p.wp + kb * W_FLOATS + min(tid + 256 * j, NW4 - 1) * 4);
#pragma unroll
    for (int j = 0; j < NIJ; ++j) {
      const int idx = min(tid + 256 * j, NI2 - 1);
      const int iy = band * R - PAD + idx / H2;
      r.im[j] = *(const float2*)(p.in + (int64_t)b * p.in_sn + kb * RAW + min(max(iy, 0), HIN - 1) * HIN + (idx % H2) * 2);
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NW4) st4(buf + idx * 4, r.w[j]);
    }
#pragma unroll
    for (int j = 0; j < NIJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NI2) {
        const int rr = idx / H2, x2 = idx % H2;
        const int iy = band * R - PAD + rr;
        const bool ok = iy >= 0 && iy < HIN;
        float* d = buf + IMG_OFF + rr * LP + PAD + x2 * 2;
        d[0] = ok ? r.im[j].x : 0.0f;
        d[1] = ok ? r.im[j].y : 0.0f;
      }
    }
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[1][TN], float*) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = wc * (TN * 32) + j * 32 + l31;
      if (c >= COLS) continue;
      float* dst = p.out + (int64_t)b * p.out_sn + band * COLS + c;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 + acc_row(r, hi);
        float v = acc[0][j][r] + p.bias[row];
        if (p.act == 1) v = fmaxf(v, 0.0f);
        dst[(int64_t)row * P] = v;
      }
    }
  }
};

// wp[ci][s = ky*KXP + kxp][hi][co] = W[co][ci][ky][2*kxp + hi]  (zero where 2*kxp + hi == KS)
__global__ __launch_bounds__(256) void band_pack_kernel(const float* __restrict__ w, int cin, int ks, float* __restrict__ wp) {
  const int kxp = (ks + 1) / 2, ksteps = ks * kxp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cin * ksteps * 128) return;
  const int co = i & 63, hi = (i >> 6) & 1, s = (i >> 7) % ksteps, ci = (i >> 7) / ksteps;
  const int ky = s / kxp, kx = 2 * (s % kxp) + hi;
  wp[i] = kx < ks ? w[((co * cin + ci) * ks + ky) * ks + kx] : 0.0f;
}

// Weight gradient of the same layers:
//   part[split][co][ci][ky][kx] = sum over the split's sample pairs and all output pixels of
//                                 dz[b][co][oy][ox] * in[b][ci][oy + ky - PAD][ox + kx - PAD]
// rows = co (64 per workgroup: 2 wave rows x 32), cols = taps of CT input channels (4 x 32 per wave
// column), k-block = (sample pair, band of R output rows); the two k indices of an MFMA are the SAME
// pixel of the pair's two samples.  dz of the band and the R + KS - 1 input rows it touches are
// staged raw (16- / 8-byte coalesced loads); image rows outside the input are written as zeros and
// the PAD columns of the LDS planes stay zero, so every operand is again lane_base + immediate.
template <int CIN, int COUT, int KS, int HIN, int PAD, int R, int CT, int TNV = 4>
struct DirectWgrad {
  static constexpr int THREADS = 256, TM = 1, TN = TNV;  // column tile = 2 wave columns x TN x 32 taps
  static constexpr int KK = KS * KS, KT = CIN * KK;
  static constexpr int OH = HIN + 2 * PAD - KS + 1, P = OH * OH, RAW = HIN * HIN, LP = HIN + 2 * PAD;
  static_assert(OH % R == 0 && (R * OH) % 4 == 0 && HIN % 2 == 0, "band geometry");
  static constexpr int NB = OH / R, KSTEPS = R * OH, BR = R + KS - 1;
  static constexpr int ASTR = KSTEPS + 1;  // odd row stride: lanes (= rows) hit distinct banks
  static constexpr int A_FLOATS = 2 * 64 * ASTR, B_OFF = (A_FLOATS + 3) / 4 * 4;
  static constexpr int BPL = BR * LP, B_FLOATS = 2 * CT * BPL, STAGE = B_OFF + (B_FLOATS + 3) / 4 * 4;
  static constexpr int CTILES = (CIN + CT - 1) / CT, COLS = CT * KK;
  static_assert(COLS <= 2 * TN * 32, "column tile");
  static constexpr int K4 = KSTEPS / 4, NA4 = 2 * 64 * K4, NAJ = (NA4 + 255) / 256;
  static constexpr int H2 = HIN / 2, NB2 = 2 * CT * BR * H2, NBJ = (NB2 + 255) / 256;
  struct Params {
    const float* in;
    int64_t in_sn;
    const float* dz;
    int64_t out_sn;
    float* part;  // [nsplit][COUT*KT + COUT]
    int n, nsplit;
  };
  struct Regs {
    f4 a[NAJ];
    float2 b[NBJ];
    unsigned oka, okb;
  };
  int abase[1], bbase[TN], kb_begin, kb_end;
  int ct, split, r0, ch0, l31, hi, wr, wc;
  float bacc;
  static constexpr int aoff(int s) { return s; }
  static constexpr int boff(int s) { return (s / OH) * LP + (s % OH); }
  __device__ __forceinline__ void init(const Params& p, int tid, float* lds) {
    const int lane = tid & 63, wave = tid >> 6;
    wr = wave >> 1;
    wc = wave & 1;
    l31 = lane & 31;
    hi = lane >> 5;
    ct = blockIdx.x;
    split = blockIdx.y;
    r0 = blockIdx.z * 64;
    ch0 = ct * CT;
    const int npairs = (p.n + 1) >> 1;
    const int per = (npairs + p.nsplit - 1) / p.nsplit;
    const int pb = min(npairs, split * per), pe = min(npairs, pb + per);
    kb_begin = pb * NB;
    kb_end = pe * NB;
    bacc = 0.0f;
    for (int i = tid; i < B_FLOATS; i += 256) {  // PAD columns (and nothing else) rely on this
      lds[B_OFF + i] = 0.0f;
      lds[STAGE + B_OFF + i] = 0.0f;
    }
    __syncthreads();
    abase[0] = hi * (64 * ASTR) + (wr * 32 + l31) * ASTR;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = min(wc * (TN * 32) + j * 32 + l31, COLS - 1);
      const int ch = col / KK, t = col % KK;
      bbase[j] = B_OFF + hi * (CT * BPL) + ch * BPL + (t / KS) * LP + (t % KS);
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int tid = threadIdx.x;
    const int pair = kb / NB, band = kb % NB;
    r.oka = 0;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      const int idx = min(tid + 256 * j, NA4 - 1);
      const int smp = idx / (64 * K4), row = (idx / K4) % 64, q = idx % K4;
      const int b = 2 * pair + smp;
      r.oka |= (b < p.n ? 1u : 0u) << j;
      r.a[j] = ld4(p.dz + (int64_t)min(b, p.n - 1) * p.out_sn + (int64_t)(r0 + row) * P + band * KSTEPS + q * 4);
    }
    r.okb = 0;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int idx = min(tid + 256 * j, NB2 - 1);
      const int x2 = idx % H2, rr = (idx / H2) % BR, ch = (idx / (H2 * BR)) % CT, smp = idx / (H2 * BR * CT);
      const int iy = band * R - PAD + rr;
      const bool ok = iy >= 0 && iy < HIN && ch0 + ch < CIN;
      const int b = min(2 * pair + smp, p.n - 1);
      const int64_t off = (int64_t)b * p.in_sn + (int64_t)min(ch0 + ch, CIN - 1) * RAW + min(max(iy, 0), HIN - 1) * HIN + x2 * 2;
      r.b[j] = *(const float2*)(p.in + off);
      r.okb |= (ok ? 1u : 0u) << j;
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NA4) {
        const int smp = idx / (64 * K4), row = (idx / K4) % 64, q = idx % K4;
        const f4 v = ((r.oka >> j) & 1u) ? r.a[j] : zero4();  // a missing sample contributes zero
        float* d = buf + smp * (64 * ASTR) + row * ASTR + q * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NB2) {
        const int x2 = idx % H2, rr = (idx / H2) % BR, ch = (idx / (H2 * BR)) % CT, smp = idx / (H2 * BR * CT);
        const bool ok = (r.okb >> j) & 1u;
        float* d = buf + B_OFF + smp * (CT * BPL) + ch * BPL + rr * LP + PAD + x2 * 2;
        d[0] = ok ? r.b[j].x : 0.0f;
        d[1] = ok ? r.b[j].y : 0.0f;
      }
    }
  }
  __device__ __forceinline__ void extra(const float* cur) {  // bias gradient: row sums of the staged dz
    if (ct == 0 && threadIdx.x < 128) {
      const float* row = cur + threadIdx.x * ASTR;
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) s += row[q];
      bacc += s;
    }
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[1][TN], float* lds) {
    float* slab = p.part + (int64_t)split * ((int64_t)COUT * KT + COUT);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wc * (TN * 32) + j * 32 + l31;
      if (col >= COLS || ch0 + col / KK >= CIN) continue;
      const int tap = ct * COLS + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(int64_t)(r0 + wr * 32 + acc_row(r, hi)) * KT + tap] = acc[0][j][r];
    }
    if (ct == 0) {
      if (threadIdx.x < 128) lds[threadIdx.x] = bacc;
      __syncthreads();
      if (threadIdx.x < 64) slab[(int64_t)COUT * KT + r0 + threadIdx.x] = lds[threadIdx.x] + lds[64 + threadIdx.x];
    }
  }
};

}  // namespace dconv

// ---- dispatch table ------------------------------------------------------------------------------
// NavPreNet1D (nav_encoder.py:96-98): conv1 3->64 7x7 @48 (weight gradient only: 3 input channels do
// not pair up for the forward's channel-pair MFMA), conv2 64->128 5x5 @22, conv3 128->256 3x3 @10.
// NavPreNet / NavPedPreNet (nav_encoder.py:18-20): conv2 64->128 3x3 @24, conv3 128->256 3x3 @12
// (their conv1 planes, 50x50 padded, do not fit the whole-plane staging: gather kernels).
// A data gradient is the forward template with channels swapped, HIN' = OH and PAD' = KS-1-PAD.
using N1dC2F = dconv::Direct<64, 128, 5, 22, 1, 2>;
using N1dC2D = dconv::Direct<128, 64, 5, 20, 3, 2>;
using N1dC3F = dconv::Direct<128, 256, 3, 10, 1, 4>;
using N1dC3D = dconv::Direct<256, 128, 3, 10, 1, 4>;
using NavC2F = dconv::Direct<64, 128, 3, 24, 1, 4>;
using NavC2D = dconv::Direct<128, 64, 3, 24, 1, 4>;
using NavC3F = dconv::Direct<128, 256, 3, 12, 1, 4>;
using NavC3D = dconv::Direct<256, 128, 3, 12, 1, 4>;
using N1dC1W = dconv::DirectWgrad<3, 64, 7, 48, 1, 1, 3, 3>;   // one output row per band, 147 taps in a 192-wide tile
using N1dC1F = dconv::DirectBandFwd<3, 64, 7, 48, 1, 4, 3>;     // bands of 4 output rows (176 pixels) in a 192-wide tile
using N1dC2W = dconv::DirectWgrad<64, 128, 5, 22, 1, 2, 10>;   // bands of 2 rows, 10 channels x 25 taps per tile
using N1dC3W = dconv::DirectWgrad<128, 256, 3, 10, 1, 2, 28>;  // bands of 2 rows, 28 channels x 9 taps per tile
using NavC2W = dconv::DirectWgrad<64, 128, 3, 24, 1, 1, 28>;
using NavC3W = dconv::DirectWgrad<128, 256, 3, 12, 1, 2, 28>;

enum DirectId { kNone = -1, kN1dC2, kN1dC3, kN1dC1, kNavC2, kNavC3 };

static DirectId direct_id(const ConvGeom& g) {
  if (g.stride != 1 || g.h != g.w || g.kh != g.kw || g.pad_h != g.pad_w || g.pad_h != 1) return kNone;
  const auto is = [&](int cin, int cout, int ks, int h) { return g.cin == cin && g.cout == cout && g.kh == ks && g.h == h; };
  if (is(64, 128, 5, 22)) return kN1dC2;
  if (is(128, 256, 3, 10)) return kN1dC3;
  if (is(3, 64, 7, 48)) return kN1dC1;
  if (is(64, 128, 3, 24)) return kNavC2;
  if (is(128, 256, 3, 12)) return kNavC3;
  return kNone;
}

bool conv_has_direct(const ConvGeom& g) { return direct_id(g) != kNone && direct_id(g) != kN1dC1; }
bool conv_has_band_fwd(const ConvGeom& g) { return direct_id(g) == kN1dC1; }
bool conv_has_direct_wgrad(const ConvGeom& g) { return direct_id(g) != kNone; }

// floats of the two extra packed regions (forward, data gradient); 0 when there is no specialisation
void conv_direct_pack_sizes(const ConvGeom& g, int64_t out[2]) {
  out[0] = out[1] = 0;
  if (conv_has_band_fwd(g)) {  // forward only: [CIN][KSTEPS][2][64]
    out[0] = (int64_t)g.cin * N1dC1F::W_FLOATS;
    return;
  }
  if (!conv_has_direct(g)) return;
  out[0] = out[1] = (int64_t)g.cout * g.cin * g.kh * g.kw;  // same element count, different order
}

void launch_conv_direct_pack(const ConvGeom& g, const float* w, float* wpf, float* wpd, hipStream_t st) {
  if (conv_has_band_fwd(g)) {
    const int total = g.cin * N1dC1F::W_FLOATS;
    hipLaunchKernelGGL(dconv::band_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, g.cin, g.kh, wpf);
    return;
  }
  if (!conv_has_direct(g)) return;
  const int kk = g.kh * g.kw, cpb = direct_id(g) == kN1dC2 ? 2 : 4;  // = the CPB of the instantiations above
  const int64_t total = (int64_t)g.cout * g.cin * kk;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(dconv::direct_pack_kernel, dim3(blocks), dim3(256), 0, st, w, g.cin, g.cout, kk, cpb, 0, wpf);
  hipLaunchKernelGGL(dconv::direct_pack_kernel, dim3(blocks), dim3(256), 0, st, w, g.cin, g.cout, kk, cpb, 1, wpd);
}

template <class Op>
static void run_direct(const float* in, int64_t in_sn, const float* wp, const float* bias, int act, float* out,
                       int64_t out_sn, int n, int rows, hipStream_t st) {
  typename Op::Params p{in, in_sn, wp, bias, out, out_sn, n, act};
  launch_engine2<Op>(dim3((unsigned)(((int64_t)n * Op::P + 255) / 256), rows / 64, 1), p, st);
}

void launch_conv_direct_fwd(const ConvGeom& g, const float* in, const float* wpf, const float* bias, int act, float* out,
                            hipStream_t st) {
  if (conv_has_band_fwd(g)) {
    N1dC1F::Params p{in, g.in_sn, wpf, bias, out, g.out_sn, g.n, act};
    launch_engine2<N1dC1F>(dim3(N1dC1F::NB, g.n, 1), p, st);
    return;
  }
  switch (direct_id(g)) {
    case kN1dC2: run_direct<N1dC2F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st); break;
    case kN1dC3: run_direct<N1dC3F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st); break;
    case kNavC2: run_direct<NavC2F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st); break;
    case kNavC3: run_direct<NavC3F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st); break;
    default: break;
  }
}

void launch_conv_direct_dgrad(const ConvGeom& g, const float* dz, const float* wpd, float* din, hipStream_t st) {
  switch (direct_id(g)) {
    case kN1dC2: run_direct<N1dC2D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st); break;
    case kN1dC3: run_direct<N1dC3D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st); break;
    case kNavC2: run_direct<NavC2D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st); break;
    case kNavC3: run_direct<NavC3D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st); break;
    default: break;
  }
}

static int direct_ctiles(DirectId id) {
  switch (id) {
    case kN1dC1: return N1dC1W::CTILES;
    case kN1dC2: return N1dC2W::CTILES;
    case kN1dC3: return N1dC3W::CTILES;
    case kNavC2: return NavC2W::CTILES;
    case kNavC3: return NavC3W::CTILES;
    default: return 0;
  }
}

int conv_direct_wgrad_splits(const ConvGeom& g) {
  const DirectId id = direct_id(g);
  if (id == kNone) return 0;
  const int tiles = direct_ctiles(id) * (g.cout / 64);
  int s = (1024 + tiles - 1) / tiles;
  // a sample pair is 10 (conv2/3) to 44 (conv1) k-blocks: at least 2 pairs per split, 1 for the long ones
  const int pairs = (g.n + 1) / 2;
  const int cap = (id == kN1dC1 || id == kNavC2) ? pairs : (pairs + 1) / 2;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

template <class Op>
static void run_direct_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, int S, hipStream_t st) {
  typename Op::Params p{in, g.in_sn, dz, g.out_sn, part, g.n, S};
  launch_engine2<Op>(dim3(Op::CTILES, S, g.cout / 64), p, st);
}

void launch_conv_direct_wgrad(const ConvGeom& g, const float* in, const float* dz, float* part, float* dw, float* db,
                              hipStream_t st) {
  const int S = conv_direct_wgrad_splits(g);
  switch (direct_id(g)) {
    case kN1dC1: run_direct_wgrad<N1dC1W>(g, in, dz, part, S, st); break;
    case kN1dC2: run_direct_wgrad<N1dC2W>(g, in, dz, part, S, st); break;
    case kN1dC3: run_direct_wgrad<N1dC3W>(g, in, dz, part, S, st); break;
    case kNavC2: run_direct_wgrad<NavC2W>(g, in, dz, part, S, st); break;
    case kNavC3: run_direct_wgrad<NavC3W>(g, in, dz, part, S, st); break;
    default: return;
  }
  const int KT = g.cin * g.kh * g.kw;
  const int64_t slab = (int64_t)g.cout * KT + g.cout;
  launch_reduce_slabs2(part, S, slab, (int64_t)g.cout * KT, dw, g.cout, db, st);
}

}  // namespace ddrl
