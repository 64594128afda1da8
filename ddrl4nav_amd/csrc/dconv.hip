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

}  // namespace dconv

// ---- dispatch table ------------------------------------------------------------------------------
// NavPreNet1D.conv2 (64 -> 128, 5x5, 22 -> 20, pad 1) and conv3 (128 -> 256, 3x3, 10 -> 10, pad 1),
// each with its data gradient (channels swapped, PAD' = KS-1-PAD, input = the forward's output size)
using Nav1dC2F = dconv::Direct<64, 128, 5, 22, 1, 2>;
using Nav1dC2D = dconv::Direct<128, 64, 5, 20, 3, 2>;
using Nav1dC3F = dconv::Direct<128, 256, 3, 10, 1, 4>;
using Nav1dC3D = dconv::Direct<256, 128, 3, 10, 1, 4>;

static int direct_id(const ConvGeom& g) {
  if (g.stride != 1 || g.h != g.w || g.kh != g.kw || g.pad_h != g.pad_w) return -1;
  if (g.cin == 64 && g.cout == 128 && g.kh == 5 && g.h == 22 && g.pad_h == 1) return 0;
  if (g.cin == 128 && g.cout == 256 && g.kh == 3 && g.h == 10 && g.pad_h == 1) return 1;
  return -1;
}

bool conv_has_direct(const ConvGeom& g) { return direct_id(g) >= 0; }

// floats of the two extra packed regions (forward, data gradient); 0 when there is no specialisation
void conv_direct_pack_sizes(const ConvGeom& g, int64_t out[2]) {
  out[0] = out[1] = 0;
  if (direct_id(g) < 0) return;
  out[0] = out[1] = (int64_t)g.cout * g.cin * g.kh * g.kw;  // same element count, different order
}

void launch_conv_direct_pack(const ConvGeom& g, const float* w, float* wpf, float* wpd, hipStream_t st) {
  const int id = direct_id(g);
  if (id < 0) return;
  const int kk = g.kh * g.kw, cpb = id == 0 ? 2 : 4;
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
  if (direct_id(g) == 0) run_direct<Nav1dC2F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st);
  else run_direct<Nav1dC3F>(in, g.in_sn, wpf, bias, act, out, g.out_sn, g.n, g.cout, st);
}

void launch_conv_direct_dgrad(const ConvGeom& g, const float* dz, const float* wpd, float* din, hipStream_t st) {
  if (direct_id(g) == 0) run_direct<Nav1dC2D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st);
  else run_direct<Nav1dC3D>(dz, g.out_sn, wpd, nullptr, 0, din, g.in_sn, g.n, g.cin, st);
}

}  // namespace ddrl
