// The 3136 -> 512 linear layer (forward, data gradient, weight gradient) on the 16-bit matrix pipe (plane products, engine2.h).
// Reference: nn.Linear(3136, 512) in USTC_lab/nn/atari_encoder.py:22,31 and its autograd backward (ppo.py:122-123).
#include "engine2.h"

namespace ddrl {

// ------------------------------------------------------------------------------------------------
// Dense-layer forward:  h[b][n] = sum_k a3[b][k] Wl[n][k] + bl[n],  rows = b, cols = n, reduction = k (3136).
// Both operands are split into NPL 16-bit planes (two scaled fp16 planes, 22 bits; the weights once per optimiser step in
// optim.hip, the activations while they are staged) and the NPROD plane products that matter are accumulated in fp32 (f16x3:
// h1 g0, h0 g1, h0 g0).  128 x 128 tile, k-block 32 = 2 MFMA k-groups; LDS holds ONE stage
// ([plane][row][32 k] 16-bit, row pitch 80 B so that 16 lanes' 16-byte fragments hit distinct banks): the next
// k-block waits in registers and is committed between two barriers while the CU's other workgroup computes.
// ------------------------------------------------------------------------------------------------
using bf8f = __attribute__((ext_vector_type(8))) __bf16;

// xcd_note: XCD-aware tile order of the three dense-layer kernels (-DDDRL_FC_SWZ=0 switches it off).  The hardware deals
// workgroups to the chip's 8 XCDs round-robin by linear id and an XCD runs about 64 of them at a time; each XCD has its own
// 4 MB L2.  In launch order the workgroups that share an operand tile land on different XCDs and every L2 fetches its own copy
// (FETCH_SIZE 2-3 x the algorithmic bytes, profiles/r02_v27_pmc_traffic.json).  The kernels therefore derive their tile from
// (xcd = id mod 8, position = id / 8) so that sharers sit next to each other on one XCD.  FETCH_SIZE per launch at B = 65,536,
// off -> on: forward 3.39 -> 1.23 GB, data gradient 4.63 -> 1.27 GB (blocks of 8 x 5 tiles; 3.38 one batch tile at a time), weight
// gradient 5.04 -> 3.60 GB; the kernels' times move by 0-4 % (they are not bandwidth-bound), the neighbours' clocks gain.
#ifndef DDRL_FC_SWZ
#define DDRL_FC_SWZ 1
#endif

struct FcFwdB {
  static constexpr int PITCH = 80, PLANE = 128 * PITCH, B_OFF = NPL * PLANE, LDS_BYTES = 2 * NPL * PLANE;
};

template <bool SPLIT>  // SPLIT: acting launches, split-K with run-time k-block ranges; training launches keep compile-time loop bounds
__global__ __launch_bounds__(256) void fc_fwd_planes_kernel(const float* __restrict__ a3, int64_t a3_es, const unsigned short* __restrict__ wlb,
                                                            const float* __restrict__ amax, const float* __restrict__ params, int64_t bias_off0,
                                                            int64_t bias_off1, float* __restrict__ h, int64_t h_es, int n, int ne, int nsplit,
                                                            float* __restrict__ part, const float* __restrict__ smax, int smax_es) {
  using K = FcFwdB;
  extern __shared__ __attribute__((aligned(16))) char ldsf[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (!SPLIT && DDRL_FC_SWZ != 0 && (gridDim.y & 7) == 0) {  // XCD-aware tile order, see xcd_note above
    const int lin = bx + 4 * by, xcd = lin & 7, q = lin >> 3;
    bx = q & 3, by = (q >> 2) * 8 + xcd;  // an XCD owns the batch tiles = xcd mod 8; the 4 feature tiles of one run together and share its a3 rows
  }
  if (SPLIT && DDRL_FC_SWZ != 0) {
    // Split launches: the G = 4 x gridDim.y workgroups of one (encoder, split) share that K range of a3 (per batch tile) and of the
    // weights (per feature tile).  In dispatch order (x fastest, XCD = linear id mod 8) they would land on 8 different XCDs and every L2
    // would fetch its own copy from the memory side (51 MB instead of 19 MB at n = 256); here groups 8 i .. 8 i + 7 take XCDs 0..7, one
    // each, and the groups left over when gridDim.z is not a multiple of 8 take two XCDs each
    const int G = 4 * (int)gridDim.y, nz = (int)gridDim.z, lin = bx + 4 * by + G * bz, full = (nz & ~7) * G;
    int g, m;
    if (lin < full) {
      const int xcd = lin & 7, r = lin >> 3;
      g = (r / G) * 8 + xcd, m = r % G;
    } else {
      const int l2 = lin - full, xcd = l2 & 7, rest = nz & 7;  // rest groups over 8 XCDs
      if (rest == 4) {
        g = (nz & ~7) + (xcd >> 1), m = (l2 >> 3) * 2 + (xcd & 1);
      } else {
        g = (nz & ~7) + l2 / G, m = l2 % G;  // other remainders: dispatch order
      }
    }
    bz = g, bx = m & 3, by = m >> 2;
  }
  // nsplit > 1 (acting launches): split-K, bz = e + ne * split, bias-free partial sums part[split][e][n][512] that
  // heads_act adds up (the slab format of the f32-MFMA FcFwd2)
  const int e = SPLIT ? bz % ne : bz, split = SPLIT ? bz / ne : 0, n0 = bx * 128, b0 = by * 128;
  float sa = plane_scale(amax[amax_idx(AMAX_A3, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_WL, e)]));
  // staging maps: activations = 4 quads of 4 k per thread (row rr + 32 j, k4), weights = 2 x 3 fragments of 8 k
  const int k4 = tid & 7, rr = tid >> 3;
  const float* asrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) asrc[j] = a3 + e * a3_es + (int64_t)min(b0 + rr + 32 * j, n - 1) * FLAT + k4 * 4;
  const int k8 = tid & 3, cc = tid >> 2;
  const unsigned short* wsrc = wlb + (int64_t)e * 3 * FLAT * FEAT + (int64_t)(n0 + cc) * FLAT + k8 * 8;
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (wr * 64 + i * 32 + l31) * K::PITCH + hi * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) bB[j] = K::B_OFF + (wc * 64 + j * 32 + l31) * K::PITCH + hi * 16;
  // SPLIT (acting) launches are latency chains of PER = 7 k-blocks per workgroup: DEPTH k-blocks of loads are kept in flight (a
  // round trip to the weight planes in HBM is ~2 us, the 24 MFMAs of a k-block 0.3 us); training launches keep one set of
  // staging registers (three waves per SIMD)
  constexpr int DEPTH = SPLIT ? 4 : 1;
  f4 ar[DEPTH][4], wrg[DEPTH][2][NPL];
  auto fetch = [&](int kb, int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ar[slot][j] = ld4(asrc[j] + kb * 32);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) wrg[slot][j][p] = *(const f4*)(wsrc + (int64_t)p * FLAT * FEAT + (int64_t)j * 64 * FLAT + kb * 32);
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 v = ar[slot][j];
      unsigned pa[NPL], pb[NPL];
      split_planes_c(v.x, v.y, sa, pa);
      split_planes_c(v.z, v.w, sa, pb);
      char* d = ldsf + (rr + 32 * j) * K::PITCH + k4 * 8;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::PLANE) = make_uint2(pa[p], pb[p]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(f4*)(ldsf + K::B_OFF + p * K::PLANE + (cc + 64 * j) * K::PITCH + k8 * 16) = wrg[slot][j][p];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto compute = [&]() {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      frag8 a[NPL][2], b[NPL][2];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = *(const frag8*)(ldsf + aA[i] + p * K::PLANE + kg * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[p][j] = *(const frag8*)(ldsf + bB[j] + p * K::PLANE + kg * 32);
      }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(a[PA[t]][i], b[PB[t]][j], acc[i][j]);
    }
  };
  if constexpr (SPLIT) {
    constexpr int PER = (FLAT / 32) / DDRL_FC_ACT_SPLITS;  // the launcher passes nsplit = DDRL_FC_ACT_SPLITS
    const int kb0 = split * PER;
#pragma unroll
    for (int d = 0; d < DEPTH && d < PER; ++d) fetch(kb0 + d, d);
    if (smax != nullptr) {
      // after the fused acting convolutions (act.hip) a3's maximum is not in the slot: every sample left its own in smax[e][.] and
      // the largest of them is taken here (order-free, hence deterministic; n <= 512 values), behind the loads just requested
      float m = 0.0f;
      for (int i = tid; i < n; i += 256) m = fmaxf(m, smax[e * smax_es + i]);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      float* red = (float*)ldsf;
      if (lane == 0) red[wave] = m;
      __syncthreads();
      const float a3top = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      __syncthreads();  // the first commit below overwrites these words
      sa = plane_scale(a3top), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_WL, e)]));
    }
    commit(0);
    if (DEPTH < PER) fetch(kb0 + DEPTH, 0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      compute();
      __syncthreads();  // every wave is done with the stage
      if (k + 1 < PER) {
        commit((k + 1) % DEPTH);
        if (k + 1 + DEPTH < PER) fetch(kb0 + k + 1 + DEPTH, (k + 1) % DEPTH);
      }
      __syncthreads();
    }
  } else {
    constexpr int NKB = FLAT / 32;  // 98 k-blocks
    fetch(0, 0);
    commit(0);
    fetch(1, 0);
    __syncthreads();
    for (int kb = 0; kb < NKB; ++kb) {
      compute();
      __syncthreads();  // every wave is done with the stage
      if (kb + 1 < NKB) {
        commit(0);
        if (kb + 2 < NKB) fetch(kb + 2, 0);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nn = n0 + wc * 64 + j * 32 + l31;
    const float bias = SPLIT ? 0.0f : params[(e ? bias_off1 : bias_off0) + nn];
    float* dst = SPLIT ? part + ((int64_t)split * 2 + e) * n * FEAT : h + e * h_es;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
        if (b < n) dst[(int64_t)b * FEAT + nn] = acc[i][j][r] * inv + bias;
      }
  }
}

// ------------------------------------------------------------------------------------------------
void launch_fc_forward2(const EncCall& c, bool allow_split, hipStream_t st, bool per_sample_max) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const int nsplit = allow_split ? fc_forward_splits(c.n) : 1;  // 98 k-blocks = 14 x 7
  ProfRange pr(c.prof, nsplit > 1 ? "FcFwdSplit" : "FcFwd", st);
  {
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute((const void*)fc_fwd_planes_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcFwdB::LDS_BYTES);
      (void)hipFuncSetAttribute((const void*)fc_fwd_planes_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcFwdB::LDS_BYTES);
      configured = true;
    }
    const dim3 grid(FEAT / 128, (c.n + 127) / 128, c.L->NE * nsplit);
    if (nsplit > 1)
      hipLaunchKernelGGL(fc_fwd_planes_kernel<true>, grid, dim3(256), FcFwdB::LDS_BYTES, st, w.a3, MB * FLAT, w.wlb, w.amax, c.params,
                         c.L->enc_base[0] + c.L->enc.lb, c.L->enc_base[c.L->NE - 1] + c.L->enc.lb, w.h, MB * FEAT, c.n, c.L->NE, nsplit, w.wpart,
                         per_sample_max ? w.actmax : (const float*)nullptr, DDRL_ACT_FUSED_MAX);
    else
      hipLaunchKernelGGL(fc_fwd_planes_kernel<false>, grid, dim3(256), FcFwdB::LDS_BYTES, st, w.a3, MB * FLAT, w.wlb, w.amax, c.params,
                         c.L->enc_base[0] + c.L->enc.lb, c.L->enc_base[c.L->NE - 1] + c.L->enc.lb, w.h, MB * FEAT, c.n, c.L->NE, 1, w.wpart,
                         (const float*)nullptr, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// Dense-layer data gradient as plane products (same structure as fc_fwd_planes_kernel with the roles K = 512 features,
// N = 3,136 conv3 outputs):  dz3[b][k] = leaky'(a3[b][k]) * sum_n dh[b][n] Wl[n][k].
// dh is split into NPL planes while it is staged, the weights come pre-split and transposed from optim.hip
// (wdlb[e][plane][k 3136][n 512]); 128 x 128 tile, k-block 32 = 2 MFMA k-groups, one LDS stage.
// ------------------------------------------------------------------------------------------------
#ifndef DDRL_FCD_KBK
#define DDRL_FCD_KBK 32
#endif
#ifndef DDRL_FCD_WPE
#define DDRL_FCD_WPE 2
#endif
struct FcDgradB {  // k-block KBK = 32 or 16 features; row pitch = data + 16 B so that 16 lanes' fragments hit distinct banks
  static constexpr int KBK = DDRL_FCD_KBK, WPE = DDRL_FCD_WPE, PITCH = 2 * KBK + 16, PLANE = 128 * PITCH, B_OFF = NPL * PLANE, LDS_BYTES = 2 * NPL * PLANE;
  static constexpr int NAQ = KBK / 4, ARJ = 128 * NAQ / 256, AROWS = 256 / NAQ;  // dh: quads of 4 features per row, per thread, rows per round
  static constexpr int NWF = KBK / 8, WJ = 128 * NWF / 256, WCOLS = 256 / NWF;   // weights: fragments of 8 features per column
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FcDgradB::WPE, FcDgradB::WPE))) void fc_dgrad_planes_kernel(const float* __restrict__ dh, int64_t dh_es, const unsigned short* __restrict__ wdlb,
                                                              float* __restrict__ amax, const unsigned* __restrict__ m3, float* __restrict__ dz3, int64_t a3_es, int n) {
  using K = FcDgradB;
  extern __shared__ __attribute__((aligned(16))) char ldsg[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  if (DDRL_FC_SWZ != 0 && (gridDim.y & 7) == 0) {  // XCD-aware tile order, see xcd_note above; gridDim.x = 25
    const int lin = bx + 25 * by, xcd = lin & 7, q = lin >> 3;
    if ((gridDim.y & 63) != 0) {
      bx = q % 25, by = (q / 25) * 8 + xcd;  // an XCD owns batch tiles = xcd mod 8 and walks their 25 column tiles one batch tile at a time
    } else {                                 // ... in blocks of 8 batch tiles x 5 column tiles (2.1 + 1.3 MB of operands shared by 40 workgroups)
      const int blk = q / 40, r = q % 40;
      bx = (blk % 5) * 5 + r % 5, by = ((blk / 5) * 8 + r / 5) * 8 + xcd;
    }
  }
  const int e = blockIdx.z, k0 = bx * 128, b0 = by * 128;
  const float sa = plane_scale(amax[amax_idx(AMAX_DH, e)]), inv = 1.0f / (sa * plane_scale(amax[amax_idx(AMAX_WL, e)]));
  // staging maps: dh = 4 quads of 4 n per thread (row rr + 32 j, n4), weights = 2 x 3 fragments of 8 n (column cc + 64 j)
  const int n4 = tid % K::NAQ, rr = tid / K::NAQ;
  const float* asrc[K::ARJ];
#pragma unroll
  for (int j = 0; j < K::ARJ; ++j) asrc[j] = dh + e * dh_es + (int64_t)min(b0 + rr + K::AROWS * j, n - 1) * FEAT + n4 * 4;
  const int n8 = tid % K::NWF, cc = tid / K::NWF;
  const unsigned short* wsrc[K::WJ];
#pragma unroll
  for (int j = 0; j < K::WJ; ++j) wsrc[j] = wdlb + (int64_t)e * 3 * FLAT * FEAT + (int64_t)min(k0 + cc + K::WCOLS * j, FLAT - 1) * FEAT + n8 * 8;
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (wr * 64 + i * 32 + l31) * K::PITCH + hi * 16;
#pragma unroll
  for (int j = 0; j < 2; ++j) bB[j] = K::B_OFF + (wc * 64 + j * 32 + l31) * K::PITCH + hi * 16;
  f4 ar[K::ARJ], wrg[K::WJ][NPL];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < K::ARJ; ++j) ar[j] = ld4(asrc[j] + kb * K::KBK);
#pragma unroll
    for (int j = 0; j < K::WJ; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) wrg[j][p] = *(const f4*)(wsrc[j] + (int64_t)p * FLAT * FEAT + kb * K::KBK);
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < K::ARJ; ++j) {
      const f4 v = ar[j];
      unsigned pa[NPL], pb[NPL];
      split_planes_c(v.x, v.y, sa, pa);
      split_planes_c(v.z, v.w, sa, pb);
      char* d = ldsg + (rr + K::AROWS * j) * K::PITCH + n4 * 8;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::PLANE) = make_uint2(pa[p], pb[p]);
    }
#pragma unroll
    for (int j = 0; j < K::WJ; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(f4*)(ldsg + K::B_OFF + p * K::PLANE + (cc + K::WCOLS * j) * K::PITCH + n8 * 16) = wrg[j][p];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  constexpr int NKB = FEAT / K::KBK;
  fetch(0);
  commit();
  fetch(1);
  __syncthreads();
  for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
    for (int kg = 0; kg < K::KBK / 16; ++kg) {
      frag8 a[NPL][2], b[NPL][2];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = *(const frag8*)(ldsg + aA[i] + p * K::PLANE + kg * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[p][j] = *(const frag8*)(ldsg + bB[j] + p * K::PLANE + kg * 32);
      }
      DDRL_PLANE_PRODUCTS;
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(a[PA[t]][i], b[PB[t]][j], acc[i][j]);
    }
    __syncthreads();  // every wave is done with the stage
    if (kb + 1 < NKB) {
      commit();
      if (kb + 2 < NKB) fetch(kb + 2);
    }
    __syncthreads();
  }
  float big = 0.0f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wc * 64 + j * 32 + l31;
    if (k >= FLAT) continue;
    // the sign of a3[b][k] comes from the mask conv3's forward wrote in ITS tile layout (common.h Workspace::m3): k = (channel, pixel),
    // channel = 32 i' + acc_row(r', hi') there -> word (b, pixel, hi'), bit 16 i' + 15 - r'; 51 MB instead of the 1.6 GB of a3
    const int ch = k / 49, pix = k % 49;
    const int mbit = 16 * (ch >> 5) + 15 - ((ch & 3) + 4 * ((ch & 31) >> 3));
    const unsigned* mp = m3 + (int64_t)e * (a3_es / FLAT) * 98 + pix * 2 + ((ch >> 2) & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned mw[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = min(b0 + wr * 64 + i * 32 + acc_row(r, hi), n - 1);
        mw[r] = mp[(int64_t)b * 98];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + wr * 64 + i * 32 + acc_row(r, hi);
        const float g = leaky_bit(mw[r], mbit, acc[i][j][r] * inv);
        if (b < n) {
          dz3[e * a3_es + (int64_t)b * FLAT + k] = g;
          big = fmaxf(big, fabsf(g));
        }
      }
    }
  }
  amax_update(big, amax + amax_idx(AMAX_DZ3, e));
}

// ------------------------------------------------------------------------------------------------
// Dense-layer weight gradient as plane products:  part[s][e][n][k] = sum_{b in split s} dh[b][n] a3[b][k]  (+ the bias partial).
// Both operands are "reduction-major" in memory ([b][n] and [b][k]): they are staged as they lie -- rows = 32 samples of
// the k-block, 128 columns, split into NPL planes on the way into LDS -- and the MFMA fragments (8 consecutive
// SAMPLES of one column per lane) are read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group
// fetches a 4-row x 16-column block and every lane receives one column of it, two reads per fragment.
// Image row pitch 320 B = 256 B of data + 64 B pad: the four rows of a block fall into the four bank quarters, so the
// two groups of a 32-lane half (32 columns x 4 rows = 256 B) read conflict-free.
// 128 x 128 tile, 4 waves as 2 x 2 of 64 x 64, k-block = 32 samples = 2 MFMA k-groups, ONE LDS stage; the next k-block
// waits in registers and is split + committed between two barriers while the CU's other workgroup computes.
// ------------------------------------------------------------------------------------------------
using s4v = __attribute__((ext_vector_type(4))) short;
struct FcWgradB {
  static constexpr int KB = 32, PITCH = 320, PLANE = KB * PITCH, B_OFF = NPL * PLANE, LDS_BYTES = 2 * NPL * PLANE;
  static constexpr int64_t SLAB = (int64_t)FEAT * FLAT + FEAT;  // weights then bias, like the arena (= FcWgradB::SLAB)
};

__device__ __forceinline__ frag8 tr_fragment(const char* lds, int byte_off) {
  typedef s4v __attribute__((address_space(3))) * lds_s4;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + byte_off));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(lds + byte_off + 4 * FcWgradB::PITCH));
  typedef __attribute__((ext_vector_type(8))) short s8v;
  const s8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(frag8, v);
}

__global__ __launch_bounds__(256) void fc_wgrad_planes_kernel(const float* __restrict__ dh, int64_t dh_es, const float* __restrict__ a3,
                                                              int64_t a3_es, const float* __restrict__ amax, const float* __restrict__ gsc,
                                                              int64_t gsc_es, float* __restrict__ part, int n, int nsplit, int ne) {
  using K = FcWgradB;
  extern __shared__ __attribute__((aligned(16))) char ldsw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
#if DDRL_FC_SWZ != 0
  // 1-D grid (launch site): linear id = xcd + 8 (t + 4 m): the four feature tiles t of one (column tile, split, encoder) unit run
  // together on ONE XCD and share the unit's a3 columns through its L2 (6.7 MB per unit, read once instead of four times)
  const int unit = (int)(blockIdx.x >> 5) * 8 + (int)(blockIdx.x & 7);
  if (unit >= 25 * ne * nsplit) return;
  const int bx = unit % 25, by = (int)(blockIdx.x >> 3) & 3, bz = unit / 25;
#else
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#endif
  const int e = bz % ne, split = bz / ne;
  // dh is NORMALISED per sample (common.h Workspace::gsc): sample s is staged with the factor sd g_s / g_max (<= sd) and the sums are
  // multiplied by g_max / (sd sa); the bias gradient sums g_s dh[s] in fp32
  const float sd = WGRAD_HEADROOM * plane_scale(amax[amax_idx(AMAX_DH, e)]) / amax[amax_idx(AMAX_GMAX, e)], sa = plane_scale(amax[amax_idx(AMAX_A3, e)]),
              inv = 1.0f / (sd * sa);
  const float* gs = gsc + e * gsc_es;
  const int k0 = bx * 128, n0 = by * 128;
  const int nkb = (n + K::KB - 1) / K::KB;
  const int per = (nkb + nsplit - 1) / nsplit;
  const int kb_begin = split * per, kb_end = min(nkb, kb_begin + per);
  // staging map: thread -> 4 columns (c4) of rows kk + 8 j of the k-block
  const int c4 = tid & 31, kk = tid >> 5;
  const float* dsrc = dh + e * dh_es + n0 + c4 * 4;
  const float* asrc = a3 + e * a3_es + min(k0 + c4 * 4, FLAT - 4);   // columns past the matrix re-read its last quad (never stored)
  // fragment addresses: 16-lane group g16 covers columns 16 (g16 & 1) .. +15 of the 32-column fragment and samples
  // 8 (g16 >> 1) .. +3 (second read: +4); inside the group lane 4 q + p supplies row q, columns 4 p .. 4 p + 3
  const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int aA[2], bB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aA[i] = (8 * (g16 >> 1) + q) * K::PITCH + (wr * 64 + i * 32 + 16 * (g16 & 1) + 4 * pp) * 2;
#pragma unroll
  for (int j = 0; j < 2; ++j) bB[j] = K::B_OFF + (8 * (g16 >> 1) + q) * K::PITCH + (wc * 64 + j * 32 + 16 * (g16 & 1) + 4 * pp) * 2;
  f4 dr[4], ar[4];
  float gr[4];
  f4 bsum = zero4();
  const bool bias_owner = (bx == 0);
  auto fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = min(kb * K::KB + kk + 8 * j, n - 1);  // unconditional loads from clamped rows; masked at commit
      dr[j] = ld4(dsrc + row * FEAT);
      ar[j] = ld4(asrc + row * FLAT);
      gr[j] = gs[row];
    }
  };
  auto commit = [&](int kb) {
    if (kb * K::KB + K::KB > n) {  // ragged last k-block: samples >= n contribute zero
      rare_path();
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (kb * K::KB + kk + 8 * j >= n) dr[j] = zero4();
      rare_path();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* d = ldsw + (kk + 8 * j) * K::PITCH + c4 * 8;
      unsigned pa[NPL], pb[NPL];
      const float sdj = sd * gr[j];
      split_planes(dr[j].x, dr[j].y, sdj, pa);
      split_planes(dr[j].z, dr[j].w, sdj, pb);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + p * K::PLANE) = make_uint2(pa[p], pb[p]);
      split_planes(ar[j].x, ar[j].y, sa, pa);
      split_planes(ar[j].z, ar[j].w, sa, pb);
#pragma unroll
      for (int p = 0; p < NPL; ++p) *(uint2*)(d + K::B_OFF + p * K::PLANE) = make_uint2(pa[p], pb[p]);
    }
    if (bias_owner) bsum += (dr[0] * gr[0] + dr[1] * gr[1]) + (dr[2] * gr[2] + dr[3] * gr[3]);
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
    commit(kb_begin);
    if (kb_begin + 1 < kb_end) fetch(kb_begin + 1);
    __syncthreads();
    for (int kb = kb_begin; kb < kb_end; ++kb) {
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        frag8 a[NPL][2], b[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[p][i] = tr_fragment(ldsw, aA[i] + p * K::PLANE + kg * 16 * K::PITCH);
#pragma unroll
          for (int j = 0; j < 2; ++j) b[p][j] = tr_fragment(ldsw, bB[j] + p * K::PLANE + kg * 16 * K::PITCH);
        }
        DDRL_PLANE_PRODUCTS;
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_planes(a[PA[t]][i], b[PB[t]][j], acc[i][j]);
      }
      __syncthreads();  // every wave is done with the stage
      if (kb + 1 < kb_end) {
        commit(kb + 1);
        if (kb + 2 < kb_end) fetch(kb + 2);
      }
      __syncthreads();
    }
  }
  float* slab = part + ((int64_t)split * 2 + e) * K::SLAB;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wc * 64 + j * 32 + l31;
    if (k >= FLAT) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(int64_t)(n0 + wr * 64 + i * 32 + acc_row(r, hi)) * FLAT + k] = acc[i][j][r] * inv;
  }
  if (bias_owner) {  // one column tile per (row tile, split, e) owns the bias partial
    float* red = (float*)ldsw;
    st4(red + kk * 128 + c4 * 4, bsum);
    __syncthreads();
    if (tid < 128) {
      float s = 0.0f;
#pragma unroll
      for (int t = 0; t < 8; ++t) s += red[t * 128 + tid];
      slab[(int64_t)FEAT * FLAT + n0 + tid] = s;
    }
  }
}

void launch_fc_backward2(const EncCall& c, float* grads, hipStream_t st, int part) {  // part: 0 = both, 1 = data gradient only, 2 = weight gradient only
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
  const int S = c.splits->fc;
  if (part != 1) {
    {
      ProfRange pr(c.prof, "FcWgrad", st);
      static bool configured_w = false;
      if (!configured_w) {
        (void)hipFuncSetAttribute((const void*)fc_wgrad_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcWgradB::LDS_BYTES);
        configured_w = true;
      }
#if DDRL_FC_SWZ != 0
      const dim3 wgrid((unsigned)((25 * L.NE * S + 7) / 8 * 32));
#else
      const dim3 wgrid((FLAT + 127) / 128, FEAT / 128, L.NE * S);
#endif
      hipLaunchKernelGGL(fc_wgrad_planes_kernel, wgrid, dim3(256), FcWgradB::LDS_BYTES, st, w.dh, MB * FEAT, w.a3, MB * FLAT, w.amax, w.gsc, MB,
                         w.wpart, c.n, S, L.NE);
    }
    ProfRange pr(c.prof, "reduce_partials", st);
    launch_reduce_partials(w.wpart, S, FcWgradB::SLAB, L.NE, grads, L.enc_base[0] + L.enc.lw, L.enc_base[1] + L.enc.lw, st);
  }
  if (part != 2) {
    ProfRange pr(c.prof, "FcDgrad", st);
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute((const void*)fc_dgrad_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcDgradB::LDS_BYTES);
      configured = true;
    }
    hipLaunchKernelGGL(fc_dgrad_planes_kernel, dim3((FLAT + 127) / 128, (c.n + 127) / 128, L.NE), dim3(256), FcDgradB::LDS_BYTES, st, w.dh, MB * FEAT,
                       w.wdlb, w.amax, w.m3, w.dz3, MB * FLAT, c.n);
  }
}

}  // namespace ddrl
