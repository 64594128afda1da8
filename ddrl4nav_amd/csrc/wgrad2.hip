// Convolution weight-gradient kernels on the 16-bit matrix pipe (fp16 plane products, fp32 accumulation; engine2.h "plane scheme").
//
//   part[split][e][oc][tap] = sum over the split's samples and output pixels of
//                             dz[b][oc][pix] * in[b][ic][oy*S+ky][ox*S+kx]
// The reduction index (sample, pixel) is the SLOW index of both operands in memory: conv2 / conv3 stage both blocks
// channel-innermost and read their MFMA fragments with the transposing LDS read (ds_read_b64_tr_b16), conv1 multiplies the
// exact-fp16 pixels by two planes of dz1.  dz arrives NORMALISED per sample (common.h Workspace::gsc): every kernel multiplies the
// sample's power-of-two scale back in while it stages the sample.  The bias gradient (sum of dz) rides along in fp32.  Slabs are
// laid out like the arena (weights then bias) so that one reduce_partials launch finishes both.
//
// Reference: autograd weight/bias gradients of conv1..conv3 (atari_encoder.py:16-18 through
// actor_loss.backward(); v_loss.backward(), ppo.py:122-123).
#include <type_traits>

#include "engine2.h"

namespace ddrl {


// per-sample scale g_s through the VECTOR memory path: the index is wave-uniform, and a scalar load would make every consumer wait
// for lgkmcnt(0) -- the counter the k loop's LDS reads live on (ConvWgrad1 12.2 instead of 2.7 ms with s_load_dword in its fetch)
__device__ __forceinline__ float ld_gs(const float* gs, int idx) {
  asm volatile("" : "+v"(idx));
  return gs[idx];
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (a `#pragma unroll` over 27-36 tap blocks with lambdas in the body
// is not always honoured, and an index that stays a run-time value puts the fragment / staging arrays into scratch)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct WgradSplit {
  int pair_begin, pair_end;
  __device__ __forceinline__ void set(int n, int nsplit, int split) {
    const int npairs = (n + 1) >> 1;
    const int per = (npairs + nsplit - 1) / nsplit;
    pair_begin = min(npairs, split * per);
    pair_end = min(npairs, pair_begin + per);
  }
};

// ================================================================================================
// conv3 weight gradient as plane products (both operands fp32: dz3 and a2, split into NPL 16-bit planes while staged, NPROD plane
// products, fp32 accumulation):
//   part[s][e][oc][ic][tap] = sum_{b in split s} sum_p dz3[b][oc][p] * a2[b][ic][pos(p, tap)],   pos = (p/7 + ky) 9 + p%7 + kx
// GEMM rows = oc (64), columns = (tap, ic) (9 x 64), reduction index kappa = (sample, output pixel).
// The reduction index is the SLOW index of both operands in memory.  gfx950's transposing LDS read
// (ds_read_b64_tr_b16) removes the need for a k-contiguous layout: every lane of a 16-lane group supplies the address
// of 4 contiguous 16-bit elements (one of 4 "rows" x 4 chunks) and receives one column of the 4 rows, so an MFMA
// fragment's 8 k-values may sit ANYWHERE in LDS as long as 4 neighbouring lanes (channels) are contiguous.  Both blocks
// are therefore staged channel-innermost -- dz3 as [kappa][64 oc], a2 as [sample pixel][64 ic], 128-byte rows -- and
// a lane's k-values are 8 consecutive kappa: rows kappa (dz3) and rows rho(kappa) + tap offset (a2), rho from a table.
// One stage = 2 whole samples = 98 kappa = 7 k-groups (the last one 2/16 full: 12.5 % of the MFMAs meet zero rows).
// A workgroup owns the WHOLE 64 x 576 gradient: wave (i, j) = (oc half, ic half) x 9 taps = 9 fragment tiles, so every
// staged element is used by all 9 taps and 32 channels; 256 workgroups = 128 sample splits x 2 encoders, slabs summed in
// fixed order by reduce_partials.  LDS: 3 x 14 KB (dz3, rows 98..111 zero) + 3 x 20.25 KB (a2) = 103 KB, one stage; the
// next stage waits in registers (80 VGPRs) and is split + committed between two barriers.
// dz3 rows carry a half-swap swizzle (64-byte halves swapped on rows with bit 1 set) so that the four rows of a read
// fall into four different bank quarters; the a2 rows are read 2-way conflicted (their row index varies with the tap).
// ================================================================================================
using s4w = __attribute__((ext_vector_type(4))) short;
using bf8w = __attribute__((ext_vector_type(8))) __bf16;
using u4w = __attribute__((ext_vector_type(4))) unsigned;
#ifndef DDRL_W3_BPITCH
#define DDRL_W3_BPITCH 64
#endif
struct Wgrad3B {
  // whole samples per stage.  One sample per stage (56 KB of LDS, two workgroups per CU, 4 k-groups 49/64 full) measured
  // 3.22 against 3.07 ms: the second workgroup hides the commit, the extra zero rows cost more.
  static constexpr int NB = 2, KAPPA = NB * 49, NKG = (KAPPA + 15) / 16, AROWS = NKG * 16, BROWS = NB * 81;
  // a2 as [ic half][sample pixel][32 ic]: 64-byte rows at a pitch of 64 B, so that the four rows of a transposing read (four
  // consecutive pixels) fall into the four 64-byte bank quarters.  (As [pixel][64 ic] with 128-byte rows they shared two quarters:
  // 2-way conflicts on 62 % of the LDS cycles of a kernel that three plane products per fragment pair made LDS-bound.)
  static constexpr int BP = DDRL_W3_BPITCH, B_HALF = BROWS * BP;
  static constexpr int A_PLANE = AROWS * 128, B_PLANE = 2 * B_HALF, B_OFF = NPL * A_PLANE;
  static constexpr int LDS_BYTES = NPL * (A_PLANE + B_PLANE);      // NB = 2: 14,336 + 20,736 per plane
  static constexpr int A_UNITS = KAPPA * 8, B_UNITS = BROWS * 8;   // (row, 8-channel group) staging units: 784 / 1,296
  static constexpr int NA = (A_UNITS + 255) / 256, NBU = (B_UNITS + 255) / 256;  // per thread: 4 / 6
  static constexpr int64_t SLAB = 64 * 576 + 64;
  static constexpr int WG_PER_CU = LDS_BYTES <= 80 * 1024 ? 2 : 1;
};

__device__ __forceinline__ frag8 tr_frag3(const char* lds, int off_lo, int off_hi) {
  typedef s4w __attribute__((address_space(3))) * lds_s4;
  const s4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_lo));
  const s4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s8w;
  return __builtin_bit_cast(frag8, (s8w)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256) void conv_wgrad3_planes_kernel(const float* __restrict__ a2, int64_t a2_es, const float* __restrict__ dz3,
                                                                 int64_t dz_es, const float* __restrict__ amax, const float* __restrict__ gsc,
                                                                 int64_t gsc_es, float* __restrict__ part, int n, int nsplit, int ne) {
  using K = Wgrad3B;
  extern __shared__ __attribute__((aligned(16))) char ldsw3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  const int e = blockIdx.x % ne, split = blockIdx.x / ne;
  // dz3 is NORMALISED per sample (common.h Workspace::gsc): sample s is staged with the factor sd g_s / g_max (<= sd), the sums are
  // multiplied by g_max / (sd sa), the bias gradient sums g_s dz3[s] in fp32
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ3, e)]) / amax[amax_idx(AMAX_GMAX, e)], sa = plane_scale(amax[amax_idx(AMAX_A2, e)]),
              inv = 1.0f / (sd * sa);
  const float* gs = gsc + e * gsc_es;
  const int nst = (n + K::NB - 1) / K::NB;
  const int per = (nst + nsplit - 1) / nsplit;
  const int st_begin = split * per, st_end = min(nst, st_begin + per);
  // zero rows of the dz3 image (kappa >= 98): written once
  for (int i = tid; i < NPL * (K::AROWS - K::KAPPA) * 8; i += 256) {
    const int pl = i / ((K::AROWS - K::KAPPA) * 8), r = i % ((K::AROWS - K::KAPPA) * 8);
    *(u4w*)(ldsw3 + pl * K::A_PLANE + K::KAPPA * 128 + r * 16) = (u4w){0u, 0u, 0u, 0u};
  }
  // ---- staging maps: unit u = tid + 256 t -> (8-channel group u / rows, row u % rows); a lane's neighbours hold
  // neighbouring pixels of the same channels (coalesced dword loads at stride 49 / 81 floats over the 8 channels)
  const float* asrc[K::NA];
  const float* bsrc[K::NBU];
  int awr[K::NA], bwr[K::NBU], asmp[K::NA], bsmp[K::NBU];
#pragma unroll
  for (int t = 0; t < K::NA; ++t) {
    const int u = min(tid + 256 * t, K::A_UNITS - 1);
    const int c8 = u / K::KAPPA, kap = u % K::KAPPA, bl = kap / 49, px = kap % 49;
    asmp[t] = bl;
    asrc[t] = dz3 + e * dz_es + (c8 * 8) * 49 + px;                                 // + sample * FLAT, + c * 49
    awr[t] = kap * 128 + ((c8 * 16) ^ (((kap >> 1) & 1) * 64));
  }
#pragma unroll
  for (int t = 0; t < K::NBU; ++t) {
    const int u = min(tid + 256 * t, K::B_UNITS - 1);
    const int c8 = u / K::BROWS, rho = u % K::BROWS, bl = rho / 81, pos = rho % 81;
    bsmp[t] = bl;
    bsrc[t] = a2 + e * a2_es + (c8 * 8) * 81 + pos;                                 // + sample * 5184, + c * 81
    bwr[t] = K::B_OFF + (c8 >> 2) * K::B_HALF + rho * K::BP + (c8 & 3) * 16;
  }
  // ---- fragment addresses.  16-lane group g16: columns 16 (g16 & 1) .. +15 of the 32-channel fragment, k-values
  // 8 (g16 >> 1) .. +7; inside the group lane 4 q + pp supplies row q (first read) / q + 4 (second), chunk pp.
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = (q >> 1) & 1;  // the half swap of rows kappa = 16 g + 8 h + q (+4): bit 1 of kappa = bit 1 of q
  const int a_lane = (8 * (g16 >> 1) + q) * 128 + (((wi ^ sw) * 64) + (g16 & 1) * 32 + pp * 8);
  const int b_lane = K::B_OFF + wj * K::B_HALF + (g16 & 1) * 32 + pp * 8;
  int brow[K::NKG][2];  // byte offset of the a2 row that belongs to this lane's kappa (tap 0), first / second read
#pragma unroll
  for (int g = 0; g < K::NKG; ++g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kap = 16 * g + 8 * (g16 >> 1) + q + 4 * r;
      const int bl = kap / 49, px = kap % 49;
      brow[g][r] = kap < K::KAPPA ? (bl * 81 + (px / 7) * 9 + px % 7) * K::BP : 0;  // padded kappa: any row (dz3 is zero there)
    }
  float ar[K::NA][8], br[K::NBU][8];
  float bsum[K::NA][8];
  float gs0 = 0.0f, gs1 = 0.0f;  // g_s of the stage's two samples (wave-uniform)
#pragma unroll
  for (int t = 0; t < K::NA; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) bsum[t][c] = 0.0f;
  auto fetch = [&](int st) {
    const int s0 = st * K::NB;
    gs0 = ld_gs(gs, min(s0, n - 1));
    gs1 = ld_gs(gs, min(s0 + 1, n - 1));
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      const float* src = asrc[t] + (int64_t)min(s0 + asmp[t], n - 1) * FLAT;  // clamped sample, masked at commit
#pragma unroll
      for (int c = 0; c < 8; ++c) ar[t][c] = src[c * 49];
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      const float* src = bsrc[t] + (int64_t)min(s0 + bsmp[t], n - 1) * 5184;
#pragma unroll
      for (int c = 0; c < 8; ++c) br[t][c] = src[c * 81];
    }
  };
  auto commit = [&](int st) {
    const int s0 = st * K::NB;
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      if (t + 1 < K::NA || tid + 256 * t < K::A_UNITS) {
        if (s0 + asmp[t] >= n) {  // second sample of a ragged last stage: contributes zero
#pragma unroll
          for (int c = 0; c < 8; ++c) ar[t][c] = 0.0f;
        }
        unsigned pl[4][NPL];
        const float gt = asmp[t] ? gs1 : gs0, sdt = sd * gt;
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes(ar[t][2 * c], ar[t][2 * c + 1], sdt, pl[c]);
        char* d = ldsw3 + awr[t];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * K::A_PLANE) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
#pragma unroll
        for (int c = 0; c < 8; ++c) bsum[t][c] += ar[t][c] * gt;
      }
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      if (t + 1 < K::NBU || tid + 256 * t < K::B_UNITS) {
        unsigned pl[4][NPL];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes(br[t][2 * c], br[t][2 * c + 1], sa, pl[c]);
        char* d = ldsw3 + bwr[t];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * K::B_PLANE) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
      }
    }
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  if (st_begin < st_end) {
    fetch(st_begin);
    commit(st_begin);
    if (st_begin + 1 < st_end) fetch(st_begin + 1);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
#pragma unroll
      for (int g = 0; g < K::NKG; ++g) {
        frag8 a[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) a[p] = tr_frag3(ldsw3, a_lane + p * K::A_PLANE + g * 2048, a_lane + p * K::A_PLANE + g * 2048 + 512);
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int toff = ((t / 3) * 9 + t % 3) * K::BP;
          frag8 b[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) b[p] = tr_frag3(ldsw3, b_lane + p * K::B_PLANE + brow[g][0] + toff, b_lane + p * K::B_PLANE + brow[g][1] + toff);
#pragma unroll
          for (int m = 0; m < NPROD; ++m) acc[t] = mfma_planes(a[PA[m]], b[PB[m]], acc[t]);
        }
      }
      __syncthreads();  // every wave is done with the stage
      if (st + 1 < st_end) {
        commit(st + 1);
        if (st + 2 < st_end) fetch(st + 2);
      }
      __syncthreads();
    }
  }
  // ---- epilogue: slab[oc][ic][tap] (torch layout of conv3.weight), then the bias partial
  float* slab = part + ((int64_t)split * 2 + e) * K::SLAB;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(wi * 32 + acc_row(r, hi)) * 576 + (wj * 32 + l31) * 9 + t] = acc[t][r] * inv;
  __syncthreads();
  float* red = (float*)ldsw3;  // [unit][8]
#pragma unroll
  for (int t = 0; t < K::NA; ++t)
    if (t + 1 < K::NA || tid + 256 * t < K::A_UNITS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[(tid + 256 * t) * 8 + c] = bsum[t][c];
    }
  __syncthreads();
  if (tid < 64) {  // oc = tid: units (c8 = tid / 8) * 98 .. +97, channel tid % 8
    float sacc = 0.0f;
    for (int k = 0; k < K::KAPPA; ++k) sacc += red[((tid >> 3) * K::KAPPA + k) * 8 + (tid & 7)];
    slab[64 * 576 + tid] = sacc;
  }
}

// ------------------------------------------------------------------------------------------------
// The same weight gradient as a two-buffer software pipeline (round 6, the form of conv_wgrad2_pipe_kernel below; -DDDRL_W3_PIPE=0:
// the one-stage kernel above).  A turn = two samples = 98 reduction indices, cut into two HALF-STAGES with an LDS buffer each:
//   half 0: sample 0, pixels 0..47  (3 k-groups, exact)          its a2 image: sample 0 (81 pixels)
//   half 1: sample 0, pixel 48 + sample 1, pixels 0..48 (50 of 64) its a2 image: rows 6..8 of sample 0 (27 pixels) + sample 1 (81)
// 76 KB together (the one stage: 103 KB).  While the waves multiply one buffer they split + commit the other half-stage and request
// the one after it, one staging unit per few tap blocks; ONE barrier per half-stage.  Rows 6..8 of sample 0 are staged twice.
// ------------------------------------------------------------------------------------------------
#ifndef DDRL_W3_PIPE
#ifdef DDRL_PLANES_BF16
#define DDRL_W3_PIPE 0
#else
#define DDRL_W3_PIPE 1
#endif
#endif
#ifndef DDRL_W3_PIN
#define DDRL_W3_PIN 4
#endif
#ifndef DDRL_W3_WPE
#define DDRL_W3_WPE 1   // waves per SIMD the registers are cut for: 2 = two workgroups per CU (the two buffers leave room for them)
#endif
struct Wgrad3P {
  static constexpr int BP = DDRL_W3_BPITCH;
  static constexpr int KAP0 = 48, KAP1 = 50, NKG0 = 3, NKG1 = 4;   // reduction indices / k-groups of the halves
  static constexpr int PX0 = 81, PX1 = 27 + 81;                     // a2 pixels staged per half
  static constexpr int A_PLANE0 = NKG0 * 16 * 128, A_PLANE1 = NKG1 * 16 * 128;
  static constexpr int B_HALF0 = PX0 * BP, B_HALF1 = PX1 * BP, B_PLANE0 = 2 * B_HALF0, B_PLANE1 = 2 * B_HALF1;
  static constexpr int A0 = 0, B0 = A0 + NPL * A_PLANE0, A1 = B0 + NPL * B_PLANE0, B1 = A1 + NPL * A_PLANE1;
  static constexpr int LDS_BYTES = B1 + NPL * B_PLANE1;
  static constexpr int AU0 = KAP0 * 8, AU1 = KAP1 * 8, BU0 = PX0 * 8, BU1 = PX1 * 8;   // staging units per half
  static constexpr int NA = 2, NB0 = 3, NB1 = 4;                    // per thread
  static_assert(AU0 <= 256 * NA && AU1 <= 256 * NA && BU0 <= 256 * NB0 && BU1 <= 256 * NB1 && NB0 + 1 == NB1, "units per thread");
  static_assert(!DDRL_W3_PIPE || LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(98 * 8 * 8 * 4 <= LDS_BYTES, "the bias reduction reuses the buffers");
  static constexpr int WG_PER_CU = (DDRL_W3_WPE >= 2 && LDS_BYTES <= 80 * 1024) ? 2 : 1;
};

__global__ __launch_bounds__(256, DDRL_W3_WPE) void conv_wgrad3_pipe_kernel(const float* __restrict__ a2, int64_t a2_es, const float* __restrict__ dz3,
                                                                          int64_t dz_es, const float* __restrict__ amax, const float* __restrict__ gsc,
                                                                          int64_t gsc_es, float* __restrict__ part, int n, int nsplit, int ne) {
  using K = Wgrad3P;
  extern __shared__ __attribute__((aligned(16))) char ldsq[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  const int e = blockIdx.x % ne, split = blockIdx.x / ne;
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ3, e)]) / amax[amax_idx(AMAX_GMAX, e)], sa = plane_scale(amax[amax_idx(AMAX_A2, e)]),
              inv = 1.0f / (sd * sa);
  const float* gs = gsc + e * gsc_es;
  const int nst = (n + 1) / 2;                           // turns = sample pairs
  const int per = (nst + nsplit - 1) / nsplit;
  const int st_begin = split * per, st_end = min(nst, st_begin + per);
  // zero rows of half 1's dz3 image (local kappa >= 50): written once, never touched by a commit
  for (int i = tid; i < NPL * (K::NKG1 * 16 - K::KAP1) * 8; i += 256) {
    const int pl = i / ((K::NKG1 * 16 - K::KAP1) * 8), r = i % ((K::NKG1 * 16 - K::KAP1) * 8);
    *(u4w*)(ldsq + K::A1 + pl * K::A_PLANE1 + K::KAP1 * 128 + r * 16) = (u4w){0u, 0u, 0u, 0u};
  }
  // ---- staging maps per half h and unit t (offsets relative to the turn's FIRST sample; a unit past the half's count repeats the last
  // one and contributes nothing to the bias sums)
  const float* dzb = dz3 + e * dz_es;
  const float* a2b = a2 + e * a2_es;
  int aoff[2][K::NA], awr[2][K::NA], ared[2][K::NA], boff[2][K::NB1], bwr[2][K::NB1];
  float alive[2][K::NA];
  unsigned a_s1 = 0u, b_s1 = 0u;                        // bit (h * 4 + t): the unit reads the turn's SECOND sample
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kaph = h ? K::KAP1 : K::KAP0, au = h ? K::AU1 : K::AU0, pxh = h ? K::PX1 : K::PX0, bu = h ? K::BU1 : K::BU0;
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      const int u = tid + 256 * t, uc = min(u, au - 1), c8 = uc / kaph, kl = uc % kaph;
      const int second = h && kl > 0, px = h ? (kl > 0 ? kl - 1 : 48) : kl;   // half 1: local 0 = sample 0's last pixel, then sample 1
      alive[h][t] = u < au ? 1.0f : 0.0f;
      a_s1 |= (unsigned)second << (h * 4 + t);
      aoff[h][t] = second * FLAT + (c8 * 8) * 49 + px;                                       // + sample0 * FLAT, + c * 49
      awr[h][t] = (h ? K::A1 : K::A0) + kl * 128 + ((c8 * 16) ^ (((kl >> 1) & 1) * 64));
      ared[h][t] = u < au ? (c8 * 98 + second * 49 + px) * 8 : -1;                            // slot of the unit's bias sums in the final reduction
    }
#pragma unroll
    for (int t = 0; t < K::NB1; ++t) {
      const int u = tid + 256 * t, uc = min(u, bu - 1), c8 = uc / pxh, lp = uc % pxh;
      const int second = h && lp >= 27, pos = h ? (lp >= 27 ? lp - 27 : 54 + lp) : lp;       // half 1: rows 6..8 of sample 0, then sample 1
      b_s1 |= (unsigned)second << (h * 4 + t);
      boff[h][t] = second * 5184 + (c8 * 8) * 81 + pos;                                       // + sample0 * 5184, + c * 81
      bwr[h][t] = (h ? K::B1 : K::B0) + (c8 >> 2) * (h ? K::B_HALF1 : K::B_HALF0) + lp * K::BP + (c8 & 3) * 16;
    }
  }
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = (q >> 1) & 1;
  const int a_lane = (8 * (g16 >> 1) + q) * 128 + (((wi ^ sw) * 64) + (g16 & 1) * 32 + pp * 8);
  const int b_lane = (g16 & 1) * 32 + pp * 8;
  int brow0[K::NKG0][2], brow1[K::NKG1][2];  // a2 row of this lane's local kappa (tap 0); padded kappa read row 0 (dz3 is zero there)
#pragma unroll
  for (int g = 0; g < K::NKG1; ++g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kl = 16 * g + 8 * (g16 >> 1) + q + 4 * r;
      if (g < K::NKG0) brow0[g][r] = ((kl / 7) * 9 + kl % 7) * K::BP;                      // sample 0, pixel kl
      const int px = kl - 1;
      brow1[g][r] = kl == 0 ? 6 * K::BP : (kl < K::KAP1 ? (27 + (px / 7) * 9 + px % 7) * K::BP : 0);
    }
  float ar[K::NA][8], br[K::NB1][8], bsum[2][K::NA][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < K::NA; ++t)
#pragma unroll
      for (int c = 0; c < 8; ++c) bsum[h][t][c] = 0.0f;
  // register SLOT j of a half: 0 .. 3 = a2 units (br[j]; half 0 has three, its slot 3 is empty), 4 / 5 = the two dz3 units (ar[j - 4]).
  // A slot is requested for the next half-stage right after the one it held has been committed, so the slots -- not the unit counts --
  // pair the two lists.  s0 = the turn; its second sample is clamped to the batch's last one when the batch is odd (valid memory; its
  // dz3 is committed with g = 0)
  auto fetch_unit = [&](int h, int j, int s0) __attribute__((always_inline)) {
    const int nb = K::NB1;
    if (j < nb) {
      const int second = (int)((b_s1 >> (h * 4 + j)) & 1u);
      const int64_t base = (int64_t)(second && 2 * s0 + 1 >= n ? 2 * s0 * 5184 + boff[h][j] - 5184 : 2 * s0 * 5184 + boff[h][j]);
      const float* src = a2b + base;
#pragma unroll
      for (int c = 0; c < 8; ++c) br[j][c] = src[c * 81];
    } else {
      const int t = j - nb;
      const int second = (int)((a_s1 >> (h * 4 + t)) & 1u);
      const int64_t base = (int64_t)(second && 2 * s0 + 1 >= n ? 2 * s0 * FLAT + aoff[h][t] - FLAT : 2 * s0 * FLAT + aoff[h][t]);
      const float* src = dzb + base;
#pragma unroll
      for (int c = 0; c < 8; ++c) ar[t][c] = src[c * 49];
    }
  };
  auto commit_unit = [&](int h, int j, float g0, float g1) __attribute__((always_inline)) {  // g = the samples' g_s (0: absent sample)
    const int nb = K::NB1;
    unsigned pl[4][NPL];
    if (j < nb) {
#pragma unroll
      for (int c = 0; c < 4; ++c) split_planes(br[j][2 * c], br[j][2 * c + 1], sa, pl[c]);
      char* d = ldsq + bwr[h][j];
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * (h ? K::B_PLANE1 : K::B_PLANE0)) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
    } else {
      const int t = j - nb;
      const float g = ((a_s1 >> (h * 4 + t)) & 1u) ? g1 : g0, sdt = sd * g, gb = g * alive[h][t];
#pragma unroll
      for (int c = 0; c < 4; ++c) split_planes(ar[t][2 * c], ar[t][2 * c + 1], sdt, pl[c]);
      char* d = ldsq + awr[h][t];
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * (h ? K::A_PLANE1 : K::A_PLANE0)) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
#pragma unroll
      for (int c = 0; c < 8; ++c) bsum[h][t][c] += ar[t][c] * gb;
    }
  };
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  // one phase: multiply buffer H (NKG x 9 tap blocks of NPROD MFMAs) while half 1 - H (of the turn `sc` scales belong to) is committed
  // into the other buffer and half H of turn `sf` requested into the registers that held it
  auto phase = [&](auto hc, float g0, float g1, int sf) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value;
    constexpr int NKG = H ? K::NKG1 : K::NKG0, NBLK = NKG * 9;
    constexpr int APL = H ? K::A_PLANE1 : K::A_PLANE0, BPL = H ? K::B_PLANE1 : K::B_PLANE0;
    constexpr int NSLOT = K::NB1 + K::NA, NOPS = 2 * NSLOT, STEP = NBLK / NOPS;   // commit slot 0, request slot 0, commit slot 1, ...
    const char* ab = ldsq + (H ? K::A1 : K::A0) + a_lane;
    const char* bb = ldsq + (H ? K::B1 : K::B0) + wj * (H ? K::B_HALF1 : K::B_HALF0) + b_lane;
    DDRL_PLANE_PRODUCTS;
    auto read_a = [&](int g, frag8 (&a)[NPL]) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) a[p] = tr_frag3(ab, p * APL + g * 2048, p * APL + g * 2048 + 512);
    };
    auto read_b = [&](int blk, frag8 (&b)[NPL]) __attribute__((always_inline)) {
      const int g = blk / 9, t = blk % 9, toff = ((t / 3) * 9 + t % 3) * K::BP;
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        b[p] = tr_frag3(bb, p * BPL + (H ? brow1[g][0] : brow0[g < K::NKG0 ? g : 0][0]) + toff, p * BPL + (H ? brow1[g][1] : brow0[g < K::NKG0 ? g : 0][1]) + toff);
    };
    frag8 a[2][NPL], b[2][NPL];
    read_a(0, a[0]);
    read_b(0, b[0]);
    static_for<0, NBLK>([&](auto blk_c) __attribute__((always_inline)) {
      constexpr int blk = decltype(blk_c)::value;
      constexpr int g = blk / 9, t = blk % 9;
      if constexpr (blk + 1 < NBLK) {
        read_b(blk + 1, b[(blk + 1) & 1]);
        if constexpr ((blk + 1) % 9 == 0) read_a(g + 1, a[(g + 1) & 1]);
      }
#pragma unroll
      for (int m = 0; m < NPROD; ++m) acc[t] = mfma_planes(a[g & 1][PA[m]], b[blk & 1][PB[m]], acc[t]);
      if constexpr (blk % STEP == STEP - 1 && blk / STEP < NOPS) {
        constexpr int op = blk / STEP, slot = op >> 1, hh = (op & 1) ? H : 1 - H;   // even: commit the other half's unit, odd: request this half's
        constexpr bool exists = slot != K::NB0 || hh == 1 || K::NB0 == K::NB1;      // half 0 has no a2 unit in slot 3
        if constexpr (exists) {
          if constexpr ((op & 1) == 0) commit_unit(hh, slot, g0, g1);
          else fetch_unit(hh, slot, sf);
        }
      }
#if DDRL_W3_PIN
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int k = 0; k < NPROD; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2 * DDRL_W3_PIN, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 2 * NPL, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
    });
  };
  if (st_begin < st_end) {
    auto gs_of = [&](int st, float& g0, float& g1) {
      g0 = ld_gs(gs, min(2 * st, n - 1));
      g1 = 2 * st + 1 < n ? ld_gs(gs, 2 * st + 1) : 0.0f;
    };
    float g0, g1;
    gs_of(st_begin, g0, g1);
#pragma unroll
    for (int j = 0; j < K::NB1 + K::NA; ++j)
      if (j != K::NB0) fetch_unit(0, j, st_begin);
#pragma unroll
    for (int j = 0; j < K::NB1 + K::NA; ++j)
      if (j != K::NB0) commit_unit(0, j, g0, g1);
#pragma unroll
    for (int j = 0; j < K::NB1 + K::NA; ++j) fetch_unit(1, j, st_begin);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
      const bool more = st + 1 < st_end;
      const int sn = more ? st + 1 : st;               // past the end: re-read the last turn (valid memory), committed with g = 0, never multiplied
      float n0 = 0.0f, n1 = 0.0f;
      if (more) gs_of(sn, n0, n1);
      phase(std::integral_constant<int, 0>{}, g0, g1, sn);   // multiply half 0 of turn st, commit half 1 of turn st, request half 0 of turn sn
      __syncthreads();
      phase(std::integral_constant<int, 1>{}, n0, n1, sn);   // multiply half 1 of turn st, commit half 0 of turn sn, request half 1 of turn sn
      __syncthreads();
      g0 = n0, g1 = n1;
    }
  }
  // ---- epilogue: slab[oc][ic][tap] (torch layout of conv3.weight), then the bias partial
  float* slab = part + ((int64_t)split * 2 + e) * Wgrad3B::SLAB;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(wi * 32 + acc_row(r, hi)) * 576 + (wj * 32 + l31) * 9 + t] = acc[t][r] * inv;
  __syncthreads();
  float* red = (float*)ldsq;  // [c8 * 98 + kappa][8]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < K::NA; ++t)
      if (ared[h][t] >= 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) red[ared[h][t] + c] = bsum[h][t][c];
      }
  __syncthreads();
  if (tid < 64) {
    float sacc = 0.0f;
    for (int k = 0; k < 98; ++k) sacc += red[((tid >> 3) * 98 + k) * 8 + (tid & 7)];
    slab[64 * 576 + tid] = sacc;
  }
}

void launch_conv_wgrad3_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
#if DDRL_W3_PIPE
  const int want = 256 * Wgrad3P::WG_PER_CU / L.NE;  // as many workgroups as fit the chip at once
#else
  const int want = 256 * Wgrad3B::WG_PER_CU / L.NE;
#endif
  const int S = c.splits->c3 < want ? c.splits->c3 : want;
  {
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad3_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wgrad3B::LDS_BYTES);
#if DDRL_W3_PIPE
      (void)hipFuncSetAttribute((const void*)conv_wgrad3_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wgrad3P::LDS_BYTES);
#endif
      configured = true;
    }
    ProfRange pr(c.prof, "ConvWgrad3", st);
#if DDRL_W3_PIPE
    hipLaunchKernelGGL(conv_wgrad3_pipe_kernel, dim3((unsigned)(L.NE * S)), dim3(256), Wgrad3P::LDS_BYTES, st, w.a2, MB * 5184, w.dz3, MB * FLAT,
                       w.amax, w.gsc, MB, w.wpart, c.n, S, L.NE);
#else
    hipLaunchKernelGGL(conv_wgrad3_planes_kernel, dim3((unsigned)(L.NE * S)), dim3(256), Wgrad3B::LDS_BYTES, st, w.a2, MB * 5184, w.dz3, MB * FLAT,
                       w.amax, w.gsc, MB, w.wpart, c.n, S, L.NE);
#endif
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, Wgrad3B::SLAB, L.NE, grads, L.enc_base[0] + L.enc.c3w, L.enc_base[1] + L.enc.c3w, st);
}

// ================================================================================================
// conv2 weight gradient as plane products, the conv3 design above on conv2's geometry:
//   part[s][e][oc][ic][ky][kx] = sum_{b in split s} sum_p dz2[b][oc][p] * a1[b][ic][(2 y + ky) 20 + 2 x + kx],   p = 9 y + x
// rows = oc (64), columns = (tap, ic) (16 x 32), reduction kappa = output pixel of ONE sample per stage (81 -> 6 k-groups,
// the last one 1/16 full: 15.6 % of the MFMAs meet zero rows; two samples do not fit LDS).  dz2 is staged as
// [kappa][64 oc] (128-byte rows, half-swap swizzle), a1 as [input pixel 400][32 ic] (64-byte rows: one 32-channel
// fragment; the four rows of a read are 2 rows apart -> 2-way conflicted).  Wave (i, tg) = oc half x tap rows
// {2 tg, 2 tg + 1} = 8 fragment tiles.  LDS 3 x 12 KB + 3 x 25 KB = 111 KB, one stage, next stage in registers.
// ================================================================================================
#ifndef DDRL_W2_BPITCH
#define DDRL_W2_BPITCH 96
#endif
struct Wgrad2B {
  static constexpr int KAPPA = 81, NKG = 6, AROWS = NKG * 16, BROWS = 400;
  // a1 rows (one input pixel, 32 channels = 64 B) at a pitch of 96 B: the four rows of a transposing read are two pixels apart, so
  // their 32-byte pieces start 192 B apart and the eight pieces of a 32-lane half fall into eight different 32-byte bank groups
  // (pitch 64: two rows per group, SQ_LDS_BANK_CONFLICT 59 % of the LDS cycles -- and with three plane products per fragment pair
  // instead of six the kernel is bound by LDS bandwidth: 768 B per MFMA)
  static constexpr int BP = DDRL_W2_BPITCH;
  static constexpr int A_PLANE = AROWS * 128, B_PLANE = BROWS * BP, B_OFF = NPL * A_PLANE;
  static constexpr int LDS_BYTES = NPL * (A_PLANE + B_PLANE);      // 12,288 + 25,600 per plane
  static constexpr int WG_PER_CU = LDS_BYTES <= 80 * 1024 ? 2 : 1;
  static constexpr int A_UNITS = KAPPA * 8, B_UNITS = BROWS * 4;   // (row, 8-channel group) staging units: 648 / 1,600
  static constexpr int NA = (A_UNITS + 255) / 256, NBU = (B_UNITS + 255) / 256;  // per thread: 3 / 7
  static constexpr int64_t SLAB = 64 * 512 + 64;
};

__global__ __launch_bounds__(256) void conv_wgrad2_planes_kernel(const float* __restrict__ a1, int64_t a1_es, const float* __restrict__ dz2,
                                                                 int64_t dz_es, const float* __restrict__ amax, const float* __restrict__ gsc,
                                                                 int64_t gsc_es, float* __restrict__ part, int n, int nsplit, int ne) {
  using K = Wgrad2B;
  extern __shared__ __attribute__((aligned(16))) char ldsw2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  // wave = tap row ky (4 column tiles: kx = 0..3) x BOTH oc halves: a k-group reads 2 x 2 dz2 fragments and 4 x 2 a1 fragments for its 24
  // MFMAs (one oc half x two tap rows read 1 x 2 + 8 x 2: a third more LDS bytes per MFMA in a kernel that LDS bandwidth bounds)
  const int e = blockIdx.x % ne, split = blockIdx.x / ne;
  // dz2 is NORMALISED per sample (see conv_wgrad3_planes_kernel): the stage's one sample is staged with sd g_s / g_max
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ2, e)]) / amax[amax_idx(AMAX_GMAX, e)], sa = plane_scale(amax[amax_idx(AMAX_A1, e)]),
              inv = 1.0f / (sd * sa);
  const float* gs = gsc + e * gsc_es;
  const int per = (n + nsplit - 1) / nsplit;
  const int st_begin = split * per, st_end = min(n, st_begin + per);
  for (int i = tid; i < NPL * (K::AROWS - K::KAPPA) * 8; i += 256) {  // zero rows of the dz2 image (kappa >= 81): written once
    const int pl = i / ((K::AROWS - K::KAPPA) * 8), r = i % ((K::AROWS - K::KAPPA) * 8);
    *(u4w*)(ldsw2 + pl * K::A_PLANE + K::KAPPA * 128 + r * 16) = (u4w){0u, 0u, 0u, 0u};
  }
  // ---- staging maps (see conv_wgrad3_planes_kernel)
  const float* asrc[K::NA];
  const float* bsrc[K::NBU];
  int awr[K::NA], bwr[K::NBU];
#pragma unroll
  for (int t = 0; t < K::NA; ++t) {
    const int u = min(tid + 256 * t, K::A_UNITS - 1);
    const int c8 = u / K::KAPPA, kap = u % K::KAPPA;
    asrc[t] = dz2 + e * dz_es + (c8 * 8) * 81 + kap;                                // + sample * 5184, + c * 81
    awr[t] = kap * 128 + ((c8 * 16) ^ (((kap >> 1) & 1) * 64));
  }
#pragma unroll
  for (int t = 0; t < K::NBU; ++t) {
    const int u = min(tid + 256 * t, K::B_UNITS - 1);
    const int c8 = u / K::BROWS, pos = u % K::BROWS;
    bsrc[t] = a1 + e * a1_es + (c8 * 8) * 400 + pos;                                // + sample * 12800, + c * 400
    bwr[t] = K::B_OFF + pos * K::BP + c8 * 16;
  }
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = (q >> 1) & 1;
  int a_lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_lane[i] = (8 * (g16 >> 1) + q) * 128 + (((i ^ sw) * 64) + (g16 & 1) * 32 + pp * 8);
  const int b_lane = K::B_OFF + wave * (20 * K::BP) + (g16 & 1) * 32 + pp * 8;
  int brow[K::NKG][2];  // byte offset of the a1 row (2 y) 20 + 2 x of this lane's kappa, first / second read
#pragma unroll
  for (int g = 0; g < K::NKG; ++g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kap = 16 * g + 8 * (g16 >> 1) + q + 4 * r;
      brow[g][r] = kap < K::KAPPA ? ((kap / 9) * 40 + (kap % 9) * 2) * K::BP : 0;
    }
  float ar[K::NA][8], br[K::NBU][8];
  float bsum[K::NA][8];
#pragma unroll
  for (int t = 0; t < K::NA; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) bsum[t][c] = 0.0f;
  float gst = 0.0f;  // g_s of the staged sample (wave-uniform)
  auto fetch = [&](int st) {
    gst = ld_gs(gs, st);
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      const float* src = asrc[t] + (int64_t)st * 5184;
#pragma unroll
      for (int c = 0; c < 8; ++c) ar[t][c] = src[c * 81];
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      const float* src = bsrc[t] + (int64_t)st * 12800;
#pragma unroll
      for (int c = 0; c < 8; ++c) br[t][c] = src[c * 400];
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      if (t + 1 < K::NA || tid + 256 * t < K::A_UNITS) {
        unsigned pl[4][NPL];
        const float sdt = sd * gst;
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes_c(ar[t][2 * c], ar[t][2 * c + 1], sdt, pl[c]);
        char* d = ldsw2 + awr[t];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * K::A_PLANE) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
#pragma unroll
        for (int c = 0; c < 8; ++c) bsum[t][c] += ar[t][c] * gst;
      }
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      if (t + 1 < K::NBU || tid + 256 * t < K::B_UNITS) {
        unsigned pl[4][NPL];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_planes_c(br[t][2 * c], br[t][2 * c + 1], sa, pl[c]);
        char* d = ldsw2 + bwr[t];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * K::B_PLANE) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
      }
    }
  };
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  if (st_begin < st_end) {
    fetch(st_begin);
    commit();
    if (st_begin + 1 < st_end) fetch(st_begin + 1);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
#pragma unroll
      for (int g = 0; g < K::NKG; ++g) {
        frag8 a[2][NPL];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            a[i][p] = tr_frag3(ldsw2, a_lane[i] + p * K::A_PLANE + g * 2048, a_lane[i] + p * K::A_PLANE + g * 2048 + 512);
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // kx
          frag8 b[NPL];
#pragma unroll
          for (int p = 0; p < NPL; ++p) b[p] = tr_frag3(ldsw2, b_lane + p * K::B_PLANE + brow[g][0] + t * K::BP, b_lane + p * K::B_PLANE + brow[g][1] + t * K::BP);
#pragma unroll
          for (int m = 0; m < NPROD; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[4 * i + t] = mfma_planes(a[i][PA[m]], b[PB[m]], acc[4 * i + t]);
        }
      }
      __syncthreads();  // every wave is done with the stage
      if (st + 1 < st_end) {
        commit();
        if (st + 2 < st_end) fetch(st + 2);
      }
      __syncthreads();
    }
  }
  // ---- epilogue: slab[oc][ic][ky][kx] (torch layout of conv2.weight), then the bias partial
  float* slab = part + ((int64_t)split * 2 + e) * K::SLAB;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[((t / 4) * 32 + acc_row(r, hi)) * 512 + l31 * 16 + wave * 4 + t % 4] = acc[t][r] * inv;
  __syncthreads();
  float* red = (float*)ldsw2;  // [unit][8]
#pragma unroll
  for (int t = 0; t < K::NA; ++t)
    if (t + 1 < K::NA || tid + 256 * t < K::A_UNITS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) red[(tid + 256 * t) * 8 + c] = bsum[t][c];
    }
  __syncthreads();
  if (tid < 64) {
    float sacc = 0.0f;
    for (int k = 0; k < K::KAPPA; ++k) sacc += red[((tid >> 3) * K::KAPPA + k) * 8 + (tid & 7)];
    slab[64 * 512 + tid] = sacc;
  }
}

// ------------------------------------------------------------------------------------------------
// The same weight gradient as a two-buffer software pipeline (round 6; -DDDRL_W2_PIPE=0: the one-stage kernel above).
// conv_wgrad2_planes_kernel runs ONE wave per SIMD (128 accumulators, 111 KB of LDS) and alternates "multiply the stage" with "split
// + commit the next stage": nothing covers the commit (its counters: matrix pipe busy 0.41, 3.1 vector instructions per MFMA, all of
// them outside the matrix phase).  Here a sample is cut into two HALF-STAGES of three k-groups -- output rows y 0..4 (45 pixels, a1
// rows 0..11) and y 5..8 (36 pixels, a1 rows 10..19) -- each with its own LDS buffer (12 + 46 KB and 12 + 38 KB: 106.5 KB together,
// LESS than the one stage, which held all 20 rows at once).  While the waves multiply one buffer they split and commit the next
// half-stage into the other and request the one after it, one staging unit per pair of tap blocks, so the vector work sits in the
// shadow of the wave's own MFMAs and there is ONE barrier per half-stage.  Rows 10 and 11 are staged twice (+10 % of the a1 reads);
// 48 staging registers instead of 80.  Results: the same products in another grouping of the reduction index (bias sums bit-identical).
// ------------------------------------------------------------------------------------------------
#ifndef DDRL_W2_PIN
#define DDRL_W2_PIN 4  // vector instructions pinned into each gap between the MFMAs of a tap block (0: the compiler's own order)
#endif
#ifndef DDRL_W2_SPLIT
#define DDRL_W2_SPLIT split_planes  // the four-instruction v_fma_mix form (engine2.h); split_planes_c = the plain one, which the compiler packs
                                    // into v_pk_mul / v_pk_fma beside the MFMAs: 2.65 against 2.58 ms, profiles/r06_w2pipe_ab_bench.txt
#endif
#ifndef DDRL_W2_PIPE
#ifdef DDRL_PLANES_BF16
#define DDRL_W2_PIPE 0  // three planes per operand: the two buffers exceed the LDS
#else
#define DDRL_W2_PIPE 1
#endif
#endif
struct Wgrad2P {
  static constexpr int BP = DDRL_W2_BPITCH, NKG = 3, AROWS = NKG * 16;
  static constexpr int KAP0 = 45, KAP1 = 36;                  // output pixels of the halves: y 0..4 / y 5..8
  static constexpr int ROW1 = 10;                             // first a1 row of half 1
  static constexpr int PX0 = 12 * 20, PX1 = 10 * 20;          // a1 pixels staged per half: rows 0..11 / 10..19
  static constexpr int A_PLANE = AROWS * 128, B_PLANE0 = PX0 * BP, B_PLANE1 = PX1 * BP;
  static constexpr int A0 = 0, B0 = A0 + NPL * A_PLANE, A1 = B0 + NPL * B_PLANE0, B1 = A1 + NPL * A_PLANE;
  static constexpr int LDS_BYTES = B1 + NPL * B_PLANE1;
  static constexpr int AU0 = KAP0 * 8, AU1 = KAP1 * 8, BU0 = PX0 * 4, BU1 = PX1 * 4;  // staging units (row, 8-channel group) per half
  static constexpr int NA = 2, NBU = 4;                       // per thread and half
  static_assert(AU0 <= 256 * NA && AU1 <= 256 * NA && BU0 <= 256 * NBU && BU1 <= 256 * NBU, "units per thread");
  static_assert(!DDRL_W2_PIPE || LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(648 * 8 * 4 <= LDS_BYTES, "the bias reduction reuses the buffers");
};

__global__ __launch_bounds__(256) void conv_wgrad2_pipe_kernel(const float* __restrict__ a1, int64_t a1_es, const float* __restrict__ dz2,
                                                               int64_t dz_es, const float* __restrict__ amax, const float* __restrict__ gsc,
                                                               int64_t gsc_es, float* __restrict__ part, int n, int nsplit, int ne) {
  using K = Wgrad2P;
  extern __shared__ __attribute__((aligned(16))) char ldsp[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int e = blockIdx.x % ne, split = blockIdx.x / ne;
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ2, e)]) / amax[amax_idx(AMAX_GMAX, e)], sa = plane_scale(amax[amax_idx(AMAX_A1, e)]),
              inv = 1.0f / (sd * sa);
  const float* gs = gsc + e * gsc_es;
  const int per = (n + nsplit - 1) / nsplit;
  const int st_begin = split * per, st_end = min(n, st_begin + per);
  // zero rows of the dz2 images (local kappa >= 45 / 36): written once, never touched by a commit
  for (int i = tid; i < NPL * (3 + 12) * 8; i += 256) {
    const int pl = i / (15 * 8), r = i % (15 * 8), row = r >> 3, qd = r & 7;
    const int off = row < 3 ? K::A0 + (K::KAP0 + row) * 128 : K::A1 + (K::KAP1 + row - 3) * 128;
    *(u4w*)(ldsp + off + pl * K::A_PLANE + qd * 16) = (u4w){0u, 0u, 0u, 0u};
  }
  // ---- staging maps, per half h and unit t (see conv_wgrad2_planes_kernel); a unit index past the half's count repeats the last unit
  // (the same bytes to the same address) and contributes nothing to the bias sums
  const float* dzb = dz2 + e * dz_es;
  const float* a1b = a1 + e * a1_es;
  int aoff[2][K::NA], awr[2][K::NA], boff[2][K::NBU], bwr[2][K::NBU], ared[2][K::NA];
  float alive[2][K::NA];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kaph = h ? K::KAP1 : K::KAP0, kap0 = h ? K::KAP0 : 0, au = h ? K::AU1 : K::AU0;
    const int pxh = h ? K::PX1 : K::PX0, bu = h ? K::BU1 : K::BU0, abase = h ? K::A1 : K::A0, bbase = h ? K::B1 : K::B0;
#pragma unroll
    for (int t = 0; t < K::NA; ++t) {
      const int u = tid + 256 * t, uc = min(u, au - 1), c8 = uc / kaph, kl = uc % kaph;
      alive[h][t] = u < au ? 1.0f : 0.0f;
      aoff[h][t] = (c8 * 8) * 81 + kap0 + kl;                                      // + sample * 5184, + c * 81
      awr[h][t] = abase + kl * 128 + ((c8 * 16) ^ (((kl >> 1) & 1) * 64));
      ared[h][t] = u < au ? (c8 * 81 + kap0 + kl) * 8 : -1;                        // slot of the unit's bias sums in the final reduction
    }
#pragma unroll
    for (int t = 0; t < K::NBU; ++t) {
      const int u = tid + 256 * t, uc = min(u, bu - 1), c8 = uc / pxh, pos = uc % pxh;
      boff[h][t] = (c8 * 8) * 400 + (h ? K::ROW1 * 20 : 0) + pos;                  // + sample * 12800, + c * 400
      bwr[h][t] = bbase + pos * K::BP + c8 * 16;
    }
  }
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int sw = (q >> 1) & 1;
  int a_lane[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_lane[i] = (8 * (g16 >> 1) + q) * 128 + (((i ^ sw) * 64) + (g16 & 1) * 32 + pp * 8);
  const int b_lane = wave * (20 * K::BP) + (g16 & 1) * 32 + pp * 8;
  // a1 row (2 y') 20 + 2 x of the lane's local kappa = 9 y' + x: the same formula for both halves (half 1 starts at y = 5 = a1 row 10, its
  // buffer's first row); padded kappa read row 0 (dz2 is zero there, but the operand must be a finite number: a row that was staged)
  int brow[2][K::NKG][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int g = 0; g < K::NKG; ++g)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int kl = 16 * g + 8 * (g16 >> 1) + q + 4 * r;
        brow[h][g][r] = kl < (h ? K::KAP1 : K::KAP0) ? ((kl / 9) * 40 + (kl % 9) * 2) * K::BP : 0;
      }
  float ar[K::NA][8], br[K::NBU][8], bsum[2][K::NA][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < K::NA; ++t)
#pragma unroll
      for (int c = 0; c < 8; ++c) bsum[h][t][c] = 0.0f;
  // staging unit j of a half: 0 .. NBU - 1 = a1 units, NBU .. NBU + NA - 1 = dz2 units
  constexpr int NU = K::NBU + K::NA;
  auto fetch_unit = [&](int h, int j, int s) __attribute__((always_inline)) {
    if (j < K::NBU) {
      const float* src = a1b + (int64_t)s * 12800 + boff[h][j];
#pragma unroll
      for (int c = 0; c < 8; ++c) br[j][c] = src[c * 400];
    } else {
      const float* src = dzb + (int64_t)s * 5184 + aoff[h][j - K::NBU];
#pragma unroll
      for (int c = 0; c < 8; ++c) ar[j - K::NBU][c] = src[c * 81];
    }
  };
  auto commit_unit = [&](int h, int j, float g) __attribute__((always_inline)) {  // g = the sample's g_s (0: a stage past the split's end)
    unsigned pl[4][NPL];
    if (j < K::NBU) {
#pragma unroll
      for (int c = 0; c < 4; ++c) DDRL_W2_SPLIT(br[j][2 * c], br[j][2 * c + 1], sa, pl[c]);
      char* d = ldsp + bwr[h][j];
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * (h ? K::B_PLANE1 : K::B_PLANE0)) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
    } else {
      const int t = j - K::NBU;
      const float sdt = sd * g, gb = g * alive[h][t];
#pragma unroll
      for (int c = 0; c < 4; ++c) DDRL_W2_SPLIT(ar[t][2 * c], ar[t][2 * c + 1], sdt, pl[c]);
      char* d = ldsp + awr[h][t];
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(u4w*)(d + p * K::A_PLANE) = (u4w){pl[0][p], pl[1][p], pl[2][p], pl[3][p]};
#pragma unroll
      for (int c = 0; c < 8; ++c) bsum[h][t][c] += ar[t][c] * gb;
    }
  };
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  // one phase: multiply buffer H (12 tap blocks of 6 MFMAs) while half 1 - H of sample `sc` is committed into the other buffer and half
  // H of sample `sf` requested into the registers that held it: one unit per pair of tap blocks
  auto phase = [&](auto hc, float gc, int sf) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value;
    const char* ab = ldsp + (H ? K::A1 : K::A0);
    const char* bb = ldsp + (H ? K::B1 : K::B0) + b_lane;
    constexpr int BPL = H ? K::B_PLANE1 : K::B_PLANE0;
    DDRL_PLANE_PRODUCTS;
    auto read_a = [&](int g, frag8 (&a)[2][NPL]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p) a[i][p] = tr_frag3(ab, a_lane[i] + p * K::A_PLANE + g * 2048, a_lane[i] + p * K::A_PLANE + g * 2048 + 512);
    };
    auto read_b = [&](int blk, frag8 (&b)[NPL]) __attribute__((always_inline)) {
      const int g = blk >> 2, t = blk & 3;
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[p] = tr_frag3(bb, p * BPL + brow[H][g][0] + t * K::BP, p * BPL + brow[H][g][1] + t * K::BP);
    };
    // fragments one tap block ahead (the scheduling fence at the end of a block keeps the reads of the next one from being hoisted by
    // the compiler, so they are requested here, in front of the block's MFMAs)
    frag8 a[2][2][NPL], b[2][NPL];
    read_a(0, a[0]);
    read_b(0, b[0]);
#pragma unroll
    for (int blk = 0; blk < 4 * K::NKG; ++blk) {
      const int g = blk >> 2, t = blk & 3, j = blk >> 1;
      if (blk + 1 < 4 * K::NKG) {
        read_b(blk + 1, b[(blk + 1) & 1]);
        if (((blk + 1) & 3) == 0) read_a(g + 1, a[(g + 1) & 1]);
      }
#pragma unroll
      for (int m = 0; m < NPROD; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[4 * i + t] = mfma_planes(a[g & 1][i][PA[m]], b[blk & 1][PB[m]], acc[4 * i + t]);
      if (j < NU) {
        if ((blk & 1) == 0) commit_unit(1 - H, j, gc);
        else fetch_unit(H, j, sf);
      }
#if DDRL_W2_PIN
      // the block's order: fragment reads first, then its vector work spread over the gaps between the six MFMAs, LDS stores last
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int k = 0; k < 2 * NPROD; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, DDRL_W2_PIN, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 2 * NPL, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  if (st_begin < st_end) {
    float g0 = ld_gs(gs, st_begin);
#pragma unroll
    for (int j = 0; j < NU; ++j) fetch_unit(0, j, st_begin);
#pragma unroll
    for (int j = 0; j < NU; ++j) commit_unit(0, j, g0);
#pragma unroll
    for (int j = 0; j < NU; ++j) fetch_unit(1, j, st_begin);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
      const bool more = st + 1 < st_end;
      const int sn = more ? st + 1 : st;             // past the end: re-read the last sample (valid memory), committed with g = 0, never multiplied
      const float g1 = more ? ld_gs(gs, sn) : 0.0f;
      phase(std::integral_constant<int, 0>{}, g0, sn);
      __syncthreads();
      phase(std::integral_constant<int, 1>{}, g1, sn);
      __syncthreads();
      g0 = g1;
    }
  }
  // ---- epilogue: slab[oc][ic][ky][kx] (torch layout of conv2.weight), then the bias partial
  float* slab = part + ((int64_t)split * 2 + e) * Wgrad2B::SLAB;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[((t / 4) * 32 + acc_row(r, hi)) * 512 + l31 * 16 + wave * 4 + t % 4] = acc[t][r] * inv;
  __syncthreads();
  float* red = (float*)ldsp;  // [c8 * 81 + kappa][8]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < K::NA; ++t)
      if (ared[h][t] >= 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) red[ared[h][t] + c] = bsum[h][t][c];
      }
  __syncthreads();
  if (tid < 64) {
    float sacc = 0.0f;
    for (int k = 0; k < 81; ++k) sacc += red[((tid >> 3) * 81 + k) * 8 + (tid & 7)];
    slab[64 * 512 + tid] = sacc;
  }
}

void launch_conv_wgrad2_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
  const int want = 256 * Wgrad2B::WG_PER_CU / L.NE;  // as many workgroups as fit the chip at once
  const int S = c.splits->c2 < want ? c.splits->c2 : want;
  {
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad2_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wgrad2B::LDS_BYTES);
      configured = true;
    }
    ProfRange pr(c.prof, "ConvWgrad2", st);
#if DDRL_W2_PIPE
    static bool configured_p = false;
    if (!configured_p) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad2_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wgrad2P::LDS_BYTES);
      configured_p = true;
    }
    hipLaunchKernelGGL(conv_wgrad2_pipe_kernel, dim3((unsigned)(L.NE * S)), dim3(256), Wgrad2P::LDS_BYTES, st, w.a1, MB * 12800, w.dz2, MB * 5184,
                       w.amax, w.gsc, MB, w.wpart, c.n, S, L.NE);
#else
    hipLaunchKernelGGL(conv_wgrad2_planes_kernel, dim3((unsigned)(L.NE * S)), dim3(256), Wgrad2B::LDS_BYTES, st, w.a1, MB * 12800, w.dz2, MB * 5184,
                       w.amax, w.gsc, MB, w.wpart, c.n, S, L.NE);
#endif
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, Wgrad2B::SLAB, L.NE, grads, L.enc_base[0] + L.enc.c2w, L.enc_base[1] + L.enc.c2w, st);
}

// ================================================================================================
// conv1 weight gradient on the 16-bit matrix pipe, fp32-accurate (the counterpart of conv_fwd1_planes_kernel).
//   dW1[(e,oc)][tap] = (1/255) sum_{b,oy,ox} dz1[b][(e,oc)][oy][ox] * pixel[b][ch][4 oy + ky][4 ox + kx]
// The pixels (0..255) are exact in fp16; dz1 = leaky'(a1) * da1 is split into NPL planes while it is staged (two scaled
// fp16 planes, 22 bits; -DDDRL_PLANES_BF16: three bf16 planes), the products are exact in fp32 and accumulated in fp32 by
// the MFMA; 1/255 and the planes' scale are applied to the accumulators.
//   rows = (e, oc), cols = 256 taps (wave w: taps 64 w ..), k-block = (sample pair, output row oy) as in
//   ConvWgrad1v2; its 2 x 20 output pixels form 6 fragments of 8 consecutive ox (ox 0-7, 8-15, 16-19 + 4 zeros
//   per sample) = 3 MFMAs of K = 16 (lane half h takes fragment 2 m + h).
// LDS: dz planes [sample][fragment][plane][row][8 x 16 bit]; the image rows de-interleaved by x mod 4 as 16-bit values
//   [sample][ch][8 rows][q = x mod 4][24], so that the 8 pixels 4 (ox0 + j) + kx of a fragment are contiguous in
//   plane kx mod 4 from index ox0 + kx / 4; the one-element shift of kx >= 4 is done with v_alignbit on 5 dwords.
// ================================================================================================
using bf2w = __attribute__((ext_vector_type(2))) __bf16;
using f2w = __attribute__((ext_vector_type(2))) float;

// Timing-only knock-outs (-DDDRL_W1_KO=bits; results are WRONG, only the kernel time means something):
// 1 no a1 load / leaky mask, 2 no plane split of dz1, 4 no image loads / conversion, 8 no MFMAs, 16 no alignbit shift of the pixel fragments,
// 32 no dz1 / a1 loads at all (values made up in registers)
#ifndef DDRL_W1_KO
#define DDRL_W1_KO 0
#endif
template <int NE>
struct Wgrad1B {
  static constexpr int ROWS = 32 * NE, PLANE = ROWS * 16, A_BYTES = 2 * 3 * NPL * PLANE;
  // image: a ring of 16 rows (4 groups of 4) per (sample, channel), NOT part of the double-buffered stage: consecutive
  // k-blocks (output rows) share 4 of their 8 input rows, so only the 4 new ones are loaded, converted and written per block
  static constexpr int IMG_OFF = 2 * A_BYTES, IMG_BYTES = 2 * 4 * 16 * 192 + 64, STAGE = A_BYTES;
  static constexpr int TPE = 256 / NE, QPE = 2 * 32 * 5, NDZ_J = (QPE + TPE - 1) / TPE;
  static constexpr int64_t SLAB = 32 * 256 + 32;
  static constexpr size_t LDS_BYTES = 2 * A_BYTES + IMG_BYTES;
};

template <int NE>
__global__ __launch_bounds__(256) void conv_wgrad1_planes_kernel(const uint8_t* __restrict__ frames, const float* __restrict__ dz,
                                                                 const unsigned* __restrict__ m1, int64_t m1_es, int64_t dz_es,
                                                                 const float* __restrict__ amax, const float* __restrict__ gsc, int64_t gsc_es,
                                                                 float* __restrict__ part, int n, int nsplit, int C) {
  using K = Wgrad1B<NE>;
  const int fs = C * 7056;  // bytes of one sample's C stacked frames (1..4); wave wc owns the 64 taps of channel wc and idles past C
  extern __shared__ __attribute__((aligned(16))) char ldsw[];
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int split = blockIdx.y;
  WgradSplit sp;
  sp.set(n, nsplit, split);
  const int kb_begin = sp.pair_begin * 20, kb_end = sp.pair_end * 20;
  // zero both stages once: the pad half of fragment 2 and the plane pads are never written again
  for (int i = tid; i < (int)(K::LDS_BYTES / 16); i += 256) *(f4*)(ldsw + i * 16) = zero4();
  // ---- dz staging map (as ConvWgrad1v2: threads split by encoder so that the encoder's base is wave-uniform)
  const int ew = __builtin_amdgcn_readfirstlane(tid / K::TPE), te = tid % K::TPE;
  uint32_t dzoff[K::NDZ_J], mcol[K::NDZ_J];
  int mbit[K::NDZ_J];
  int adst[K::NDZ_J];
  unsigned dz_s1 = 0, dz_ok = 0;
  float bacc[K::NDZ_J];
#pragma unroll
  for (int j = 0; j < K::NDZ_J; ++j) {
    const int idx = te + K::TPE * j, c = min(idx, K::QPE - 1);
    const int row5 = c / 5, q4 = c % 5, smp = row5 >> 5, oc = row5 & 31, f = q4 >> 1, half = q4 & 1;
    dzoff[j] = (uint32_t)((smp * 12800 + oc * 400 + q4 * 4) * 4);
    mcol[j] = (uint32_t)((smp * 400 + q4 * 4) * 4);  // byte offset of the quad's four mask words inside the pair's output row
    mbit[j] = m1_bit(oc);
    adst[j] = ((smp * 3 + f) * NPL * K::ROWS + ew * 32 + oc) * 16 + half * 8;
    dz_s1 |= (unsigned)smp << j;
    dz_ok |= (idx < K::QPE ? 1u : 0u) << j;
    bacc[j] = 0.0f;
  }
  // ---- image staging map: a k-block stages NG new row groups (1; 2 for the first block of a sample pair).
  // unit u = (R = (group gsel, sample, ch, r), g): g < 5 = pixels 16 g .. 16 g + 15, g = 5 = 80 .. 83; 192 units per group,
  // i.e. the second group = the k = 1 units of threads 0 .. 127, a lone group = the k = 0 units of waves 0 .. 2
  uint32_t imoff[2];
  int bdst[2];
  unsigned im_s1 = 0, im_g1 = 0, im_ok1 = 0, im_ok2 = 0, im_tail = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = tid + 256 * k, uc = min(u, 383);
    const int R = uc / 6, g = uc % 6, gsel = R >> 5, smp = (R >> 4) & 1, ch = min((R >> 2) & 3, C - 1), r = R & 3;  // missing channels re-read the last one
    imoff[k] = (uint32_t)(smp * fs + ch * 7056 + (4 * gsel + r) * 84 + (g < 5 ? g * 16 : 68));  // g = 5: bytes 68..83, last dword
    bdst[k] = K::IMG_OFF + ((smp * 4 + ch) * 16 + r) * 192 + (g < 5 ? g * 8 : 40);              // + ring group * 4 * 192
    im_s1 |= (unsigned)smp << k;
    im_g1 |= (unsigned)gsel << k;
    im_ok1 |= (u < 192 ? 1u : 0u) << k;
    im_ok2 |= (u < 384 ? 1u : 0u) << k;
    im_tail |= (g == 5 ? 1u : 0u) << k;
  }
  const bool lone_group_wave = __builtin_amdgcn_readfirstlane(tid) < 192;  // waves 0 .. 2
  // this thread stages encoder ew's dz1, which is NORMALISED per sample (see conv_wgrad3_planes_kernel): staged with sd g_s / g_max
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ1, ew)]) / amax[amax_idx(AMAX_GMAX, ew)];
  const float* gs = gsc + ew * gsc_es;
  // g_s of the sample each of this thread's dz quads belongs to, fetched per lane next to the quad.  (NOT two wave-uniform values
  // and a select in commit(): a select between two by-reference lambda captures becomes a load through a selected POINTER, the
  // closure's address escapes and every captured array of the kernel lands in scratch -- 12.2 instead of 2.7 ms.)
  float gsj[K::NDZ_J];
  // ---- operand lane bases: MFMA m of a k-block, lane half hi -> fragment fi = 2 m + hi = (sample fi / 3, frag fi % 3)
  int aa[NE][3], bb[2][3], shamt[2], kyhi[2], kylo[2];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int fi = 2 * m + hi, fs = fi / 3, ff = fi % 3;
#pragma unroll
    for (int i = 0; i < NE; ++i) aa[i][m] = ((fs * 3 + ff) * NPL * K::ROWS + i * 32 + l31) * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = wc * 64 + j * 32 + l31, ch = c >> 6, ky = (c >> 3) & 7, kx = c & 7;
      bb[j][m] = K::IMG_OFF + ((fs * 4 + ch) * 16) * 192 + (kx & 3) * 48 + ff * 16;  // + ring row of ky (per block)
      shamt[j] = (kx >> 2) * 16;
      kyhi[j] = ky >> 2;
      kylo[j] = ky & 3;
    }
  }
  f4 dzr[K::NDZ_J];
  u4w mkr[K::NDZ_J];  // the sign-mask words of the four pixels of dzr (written by conv1's forward, common.h Workspace::m1): bit mbit set = a1 is not positive
  u4w imr[2] = {(u4w){0u, 0u, 0u, 0u}, (u4w){0u, 0u, 0u, 0u}};
  bool full = true, imfirst = true;
  int imam = 0;  // ring group (mod 4) of the first group this fetch stages
  auto fetch = [&](int kb) __attribute__((always_inline)) {
    const int pair = kb / 20, oy = kb % 20;
    full = 2 * pair + 1 < n;
    imfirst = kb == kb_begin || oy == 0;        // first block of a pair: both of its row groups are new
    const int gfirst = imfirst ? 0 : 1;
    imam = (21 * pair + oy + gfirst) & 3;
    if (imfirst) {  // wave-uniform: the 20 k-blocks of a sample pair share their scales
#pragma unroll
      for (int j = 0; j < K::NDZ_J; ++j) gsj[j] = ld_gs(gs, min(2 * pair + (int)((dz_s1 >> j) & 1u), n - 1));
    }
    const int64_t sb = ew * dz_es + (int64_t)pair * (2 * 12800) + oy * 20;
    const char* dzb = (const char*)(dz + sb);
    const char* mb = (const char*)(m1 + ew * m1_es + (int64_t)pair * 800 + oy * 20);
    const char* fp = (const char*)frames + (int64_t)pair * (2 * fs) + (oy + gfirst) * 336;
    // odd batch tail: the second sample does not exist -> its slots read the first sample, commit() zeroes its dz
#pragma unroll
    for (int j = 0; j < K::NDZ_J; ++j) {
      const uint32_t o = full ? dzoff[j] : dzoff[j] - ((dz_s1 >> j) & 1u) * (12800u * 4u);
      if (DDRL_W1_KO & 32) { dzr[j] = (f4){(float)o, 1.0f, 2.0f, (float)kb}; mkr[j] = (u4w){o, o, o, o}; continue; }
      dzr[j] = *(const f4*)(dzb + o);
      // the lanes of one (sample, quad) -- all 32 output channels -- read the same 16 bytes: one cache line per wave instruction
      if (!(DDRL_W1_KO & 1)) mkr[j] = *(const u4w*)(mb + (full ? mcol[j] : mcol[j] - ((dz_s1 >> j) & 1u) * 1600u));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (DDRL_W1_KO & 4) continue;
      // wave-uniform skips: a lone group has no k = 1 units and none in wave 3 (their loads were the cost, not the conversion)
      if (!imfirst && (k == 1 || !lone_group_wave)) continue;
      const uint32_t o = full ? imoff[k] : imoff[k] - ((im_s1 >> k) & 1u) * (uint32_t)fs;
      const unsigned* q = (const unsigned*)(fp + o);  // 4-byte aligned
      imr[k] = (u4w){q[0], q[1], q[2], q[3]};
    }
  };
  auto commit = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < K::NDZ_J; ++j) {
      if (j + 1 < K::NDZ_J || ((dz_ok >> j) & 1u)) {
        f4 g = (DDRL_W1_KO & 1) ? dzr[j]
                                : (f4){leaky_bit(mkr[j][0], mbit[j], dzr[j].x), leaky_bit(mkr[j][1], mbit[j], dzr[j].y),
                                       leaky_bit(mkr[j][2], mbit[j], dzr[j].z), leaky_bit(mkr[j][3], mbit[j], dzr[j].w)};
        if (!full && ((dz_s1 >> j) & 1u)) g = zero4();
        const float gj = gsj[j], sdj = sd * gj;
        bacc[j] += ((g.x + g.y) + (g.z + g.w)) * gj;  // the bias gradient rides along (fp32)
        unsigned pa[NPL], pb[NPL];
        if (DDRL_W1_KO & 2) {
          for (int p = 0; p < NPL; ++p) { pa[p] = __float_as_uint(g.x) ^ __float_as_uint(g.y); pb[p] = __float_as_uint(g.z) ^ __float_as_uint(g.w); }
        } else {
        split_planes(g.x, g.y, sdj, pa);
        split_planes(g.z, g.w, sdj, pb);
        }
        char* d = st + adst[j];
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::PLANE) = make_uint2(pa[p], pb[p]);
      }
    }
    const unsigned im_ok = (DDRL_W1_KO & 4) ? 0u : imfirst ? im_ok2 : im_ok1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if ((im_ok >> k) & 1u) {
        char* d = ldsw + bdst[k] + ((imam + (int)((im_g1 >> k) & 1u)) & 3) * (4 * 192);
        if (!((im_tail >> k) & 1u)) {
          // 16 pixels x0 .. x0+15: plane q gets pixels x0+q, +4, +8, +12 = byte q of the four dwords, as 16-bit floats
          *(uint2*)(d + 0 * 48) = make_uint2(pixel_pair_sel<0, 0>(imr[k][0], imr[k][1]), pixel_pair_sel<0, 0>(imr[k][2], imr[k][3]));
          *(uint2*)(d + 1 * 48) = make_uint2(pixel_pair_sel<1, 1>(imr[k][0], imr[k][1]), pixel_pair_sel<1, 1>(imr[k][2], imr[k][3]));
          *(uint2*)(d + 2 * 48) = make_uint2(pixel_pair_sel<2, 2>(imr[k][0], imr[k][1]), pixel_pair_sel<2, 2>(imr[k][2], imr[k][3]));
          *(uint2*)(d + 3 * 48) = make_uint2(pixel_pair_sel<3, 3>(imr[k][0], imr[k][1]), pixel_pair_sel<3, 3>(imr[k][2], imr[k][3]));
        } else {
          const unsigned v = imr[k][3];  // pixels 80..83 -> element 20 of each plane
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(unsigned short*)(d + q * 48) = pixel_one((v >> (8 * q)) & 255u);
        }
      }
    }
  };
  f32x16 acc[NE][2];
#pragma unroll
  for (int i = 0; i < NE; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  __syncthreads();  // zero fill complete
  int kb = kb_begin;
  if (kb < kb_end) {
    fetch(kb);
    commit(ldsw);
    if (kb + 1 < kb_end) fetch(kb + 1);
  }
  __syncthreads();
  for (int buf = 0; kb < kb_end; ++kb, buf ^= 1) {
    const char* cur = ldsw + buf * K::STAGE;
    const int am = (21 * (kb / 20) + kb % 20) & 3;  // ring group of this block's first 4 input rows
    int rowoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) rowoff[j] = ((((am + kyhi[j]) & 3) << 2) | kylo[j]) * 192;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      frag8 b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* q = ldsw + bb[j][m] + rowoff[j];
        const uint2 w01 = *(const uint2*)q, w23 = *(const uint2*)(q + 8);
        const unsigned w4 = *(const unsigned*)(q + 16);
        const u4w o = (DDRL_W1_KO & 16) ? (u4w){w01.x, w01.y, w23.x, w23.y ^ w4}
                                        : (u4w){__builtin_amdgcn_alignbit(w01.y, w01.x, shamt[j]), __builtin_amdgcn_alignbit(w23.x, w01.y, shamt[j]),
                                                __builtin_amdgcn_alignbit(w23.y, w23.x, shamt[j]), __builtin_amdgcn_alignbit(w4, w23.y, shamt[j])};
        b[j] = __builtin_bit_cast(frag8, o);
      }
#pragma unroll
      for (int p = NPL - 1; p >= 0; --p) {  // smallest plane first
        frag8 a[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) a[i] = *(const frag8*)(cur + aa[i][m] + p * K::PLANE);
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DDRL_W1_KO & 8) { acc[i][j][0] += (float)a[i][0] + (float)b[j][0]; } else
            acc[i][j] = mfma_planes(a[i], b[j], acc[i][j]);
          }
      }
    }
    if (kb + 1 < kb_end) {
      commit(ldsw + (buf ^ 1) * K::STAGE);
      if (kb + 2 < kb_end) fetch(kb + 2);
    }
    __syncthreads();
  }
  // slabs [32 oc][64 C taps] + [32] (the arena's conv1.weight, conv1.bias): weights (scaled by the 1/255 of the frame
  // normalisation), then the bias partial
  const int ktaps = 64 * C;
  const int64_t slab_floats = 32 * ktaps + 32;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const float r255 = amax[amax_idx(AMAX_GMAX, i)] * PIXEL_UNIT / (255.0f * WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DZ1, i)]));
    float* slab = part + ((int64_t)split * 2 + i) * slab_floats;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 64 + j * 32 + l31;
      if (wc < C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[acc_row(r, hi) * ktaps + col] = acc[i][j][r] * r255;
      }
    }
  }
  float* red = (float*)ldsw;
#pragma unroll
  for (int j = 0; j < K::NDZ_J; ++j)
    if ((dz_ok >> j) & 1u) red[ew * K::QPE + te + K::TPE * j] = bacc[j];
  __syncthreads();
  if (tid < K::ROWS) {
    const int e = tid >> 5, oc = tid & 31;
    float sum = 0.0f;
#pragma unroll
    for (int smp = 0; smp < 2; ++smp)
#pragma unroll
      for (int q = 0; q < 5; ++q) sum += red[e * K::QPE + (smp * 32 + oc) * 5 + q];
    part[((int64_t)split * 2 + e) * slab_floats + 32 * ktaps + oc] = sum;
  }
}

template <int NE>
static void launch_wgrad1_planes(const EncCall& c, int S, hipStream_t st) {
  using K = Wgrad1B<NE>;
  const Workspace& w = *c.ws;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad1_planes_kernel<NE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL(conv_wgrad1_planes_kernel<NE>, dim3(1, S, 1), dim3(256), K::LDS_BYTES, st, c.frames, w.dz1, w.m1, m1_words(c.max_batch),
                     c.max_batch * 12800, w.amax, w.gsc, c.max_batch, w.wpart, c.n, S, c.L->C);
}

void launch_conv_wgrad1_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const ParamLayout& L = *c.L;
  const int S = c.splits->c1;
  {
    ProfRange pr(c.prof, "ConvWgrad1", st);
    if (L.NE == 2) {
      launch_wgrad1_planes<2>(c, S, st);
    } else {
      launch_wgrad1_planes<1>(c, S, st);
    }
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, (int64_t)32 * 64 * L.C + 32, L.NE, grads, L.enc_base[0] + L.enc.c1w, L.enc_base[1] + L.enc.c1w, st);
}

}  // namespace ddrl
