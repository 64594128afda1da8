// HBM-bound support kernels: derived weight layouts, split-K partial reduction, global grad-norm + clip + two-group Adam, the GAE scan and the u8->f32 table.
//
// Reference arithmetic replaced:
//   clip_grad_norm_ + two torch.optim.Adam steps   USTC_lab/nn/ppo.py:40-42,125-129
//   Agents._accumulate_rewards                     USTC_lab/agent/agent.py:124-140
#include "engine2.h"

namespace ddrl {

// --------------------------------------------------------------------------------------------
// derived weight layouts: 16-bit planes of every GEMM operand the weights take part in (rebuilt after every optimiser step)
// --------------------------------------------------------------------------------------------
// conv1 weights as the NPL planes of engine2.h's plane scheme (default: two scaled fp16 planes, h0 + h1 = w S to 22 bits; the
// kernel names keep their history).  Layout: common.h wp1b.
__device__ __forceinline__ void pack_conv1_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst,
                                                       const float* __restrict__ amax, int bx) {
  const int rows = 32 * L.NE;
  const int idx = bx * 256 + threadIdx.x;
  if (idx >= L.C * 4 * 2 * rows * 8) return;  // [channel C][ky pair 4][plane][lane half 2][row][kx 8]
  const int j = idx & 7, row = (idx >> 3) % rows, r = (idx >> 3) / rows;
  const int h = r & 1, g = (r >> 1) & 3, c = r >> 3;
  const int e = row >> 5, oc = row & 31, ky = 2 * g + h;
  const float w = params[L.enc_base[e] + L.enc.c1w + ((oc * L.C + c) * 8 + ky) * 8 + j];
  unsigned short pl[NPL];
  planes_of(w, plane_scale(amax[amax_idx(AMAX_W1, e)]), pl);
  const int base = (c * 4 + g) * NPL;
#pragma unroll
  for (int p = 0; p < NPL; ++p) dst[(((base + p) * 2 + h) * rows + row) * 8 + j] = pl[p];
}

// dense-layer weights as NPL planes (see pack_conv1_planes_kernel): wlb[e][plane][n][k]
__device__ __forceinline__ void pack_fc_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst,
                                                    unsigned short* __restrict__ dst_t, const float* __restrict__ amax,
                                                    unsigned short (&tile)[NPL][32][33], int bx, int by, int e) {
  // one 32 (n) x 32 (k) tile per workgroup: the planes go out row-major as they come in and, through LDS, transposed for the data
  // gradient ([plane][k][n]) -- both in 64-byte runs (element-wise the transposed copy was 2-byte stores 1 KB apart: 36 us per pack)
  const int k0 = bx * 32, n0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float sc = plane_scale(amax[amax_idx(AMAX_WL, e)]);
  const float* src = params + L.enc_base[e] + L.enc.lw;
  unsigned short* d = dst + (int64_t)e * 3 * FLAT * FEAT;
  unsigned short* t = dst_t + (int64_t)e * 3 * FLAT * FEAT;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + ty + 8 * j, k = k0 + tx;
    unsigned short pl[NPL];
    planes_of(src[(int64_t)n * FLAT + k], sc, pl);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      d[p * (int64_t)FLAT * FEAT + (int64_t)n * FLAT + k] = pl[p];
      tile[p][ty + 8 * j][tx] = pl[p];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + ty + 8 * j, n = n0 + tx;
#pragma unroll
    for (int p = 0; p < NPL; ++p) t[p * (int64_t)FLAT * FEAT + (int64_t)k * FEAT + n] = tile[p][tx][ty + 8 * j];
  }
}

// conv2 weights as NPL planes: wp2b[e][in channel][plane][oc][tap = ky * 4 + kx]
__device__ __forceinline__ void pack_conv2_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst, const float* __restrict__ amax, int bx, int e) {
  const int i = bx * 256 + threadIdx.x;  // (oc, ch, tap) in parameter order
  if (i >= 64 * 32 * 16) return;
  const int tap = i & 15, ch = (i >> 4) & 31, oc = i >> 9;
  const float w = params[L.enc_base[e] + L.enc.c2w + i];
  unsigned short pl[NPL];
  planes_of(w, plane_scale(amax[amax_idx(AMAX_W2, e)]), pl);
  unsigned short* d = dst + (((int64_t)(e * 32 + ch) * NPL) * 64 + oc) * 16 + tap;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 64 * 16] = pl[p];
}

// conv3 weights as NPL planes: wp3b[e][k-block = ic / 8][tap pair][plane][oc][tap parity][ic % 8]; the tenth tap is zero
__device__ __forceinline__ void pack_conv3_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst, const float* __restrict__ amax, int bx, int e) {
  const int i = bx * 256 + threadIdx.x;  // (kb, kg, oc, h, c)
  if (i >= 8 * 5 * 64 * 16) return;
  const int c = i & 7, h = (i >> 3) & 1, oc = (i >> 4) & 63, kg = (i >> 10) % 5, kb = (i >> 10) / 5;
  const int tap = 2 * kg + h;
  const float w = tap < 9 ? params[L.enc_base[e] + L.enc.c3w + (oc * 64 + kb * 8 + c) * 9 + tap] : 0.0f;
  unsigned short pl[NPL];
  planes_of(w, plane_scale(amax[amax_idx(AMAX_W3, e)]), pl);
  unsigned short* d = dst + ((((int64_t)(e * 8 + kb) * 5 + kg) * NPL) * 64 + oc) * 16 + h * 8 + c;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 64 * 16] = pl[p];
}

// conv2 weights for the plane-product data gradient: wd2b[e][a][kb 8][u 2][plane][row = c * 32 + ic][v][o]
//   = W2[oc = 8 kb + o][ic][2 u + a][2 v + c]
__device__ __forceinline__ void pack_dgrad2_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst, const float* __restrict__ amax, int bx, int e) {
  const int i = bx * 256 + threadIdx.x;  // (a, kb, u, row, v, o)
  if (i >= 2 * 8 * 2 * 64 * 16) return;
  const int o = i & 7, v = (i >> 3) & 1, row = (i >> 4) & 63, u = (i >> 10) & 1, kb = (i >> 11) & 7, a = i >> 14;
  const int c = row >> 5, ic = row & 31, oc = 8 * kb + o;
  const float w = params[L.enc_base[e] + L.enc.c2w + (oc * 32 + ic) * 16 + (2 * u + a) * 4 + 2 * v + c];
  unsigned short pl[NPL];
  planes_of(w, plane_scale(amax[amax_idx(AMAX_W2, e)]), pl);
  unsigned short* d = dst + (((((int64_t)(e * 2 + a) * 8 + kb) * 2 + u) * NPL) * 64 + row) * 16 + v * 8 + o;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 64 * 16] = pl[p];
}

// conv3 weights for the data gradient on planes (conv2.hip conv_dgrad3_planes_kernel): k-blocks of 16 oc, one k-group per tap
__device__ __forceinline__ void pack_dgrad3_planes_body(const float* __restrict__ params, const ParamLayout& L, unsigned short* __restrict__ dst, const float* __restrict__ amax, int bx, int e) {
  const int i = bx * 256 + threadIdx.x;  // (kb 4, tap 9, ic 64, h 2, o 8): wd3b[e][kb][tap][plane][ic][h][o] = W3[oc = 16 kb + 8 h + o][ic][tap]
  if (i >= 4 * 9 * 64 * 16) return;
  const int o = i & 7, h = (i >> 3) & 1, ic = (i >> 4) & 63, tap = (i >> 10) % 9, kb = (i >> 10) / 9;
  const float w = params[L.enc_base[e] + L.enc.c3w + ((16 * kb + 8 * h + o) * 64 + ic) * 9 + tap];
  unsigned short pl[NPL];
  planes_of(w, plane_scale(amax[amax_idx(AMAX_W3, e)]), pl);
  unsigned short* d = dst + ((((int64_t)(e * 4 + kb) * 9 + tap) * NPL) * 64 + ic) * 16 + h * 8 + o;
#pragma unroll
  for (int p = 0; p < NPL; ++p) d[p * 64 * 16] = pl[p];
}

// Weight magnitudes, ONE launch (round 6: was weights_amax + a1_bound, 24 + 5 us):
//   blocks [0, WA_BLOCKS x NE): largest magnitude of linear.weight / conv2.weight / conv3.weight / conv1.weight per encoder -> Workspace::amax
//     weight slots (the power-of-two scale of their fp16 planes, engine2.h plane scheme), slots zeroed before.  The dense layer's 1.6 M
//     values get 128 workgroups, eight independent loads per thread and round (64 workgroups walking them one dependent maximum after
//     the other: 13.5 MB in 24 us);
//   then 8 blocks per encoder: upper bound of |a1| -> slot AMAX_A1 (common.h): max over oc of sum_k |w1[oc][k]| + |b1[oc]|, one wave per oc.
constexpr int WA_NBLK[4] = {128, 8, 8, 4};  // slots AMAX_WL, AMAX_W2, AMAX_W3, AMAX_W1
constexpr int WA_BLOCKS = 128 + 8 + 8 + 4, A1B_BLOCKS = 8;
__global__ __launch_bounds__(256) void weights_amax_kernel(const float* __restrict__ params, ParamLayout L, float* __restrict__ amax) {
  int bid = blockIdx.x;
  if (bid >= WA_BLOCKS * L.NE) {  // the bound of a1
    bid -= WA_BLOCKS * L.NE;
    const int e = bid / A1B_BLOCKS, oc = (bid % A1B_BLOCKS) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, K = L.C * 64;
    const float* wsrc = params + L.enc_base[e] + L.enc.c1w + (int64_t)oc * K;
    float s = 0.0f;
    for (int k = lane; k < K; k += 64) s += fabsf(wsrc[k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    amax_update(s + fabsf(params[L.enc_base[e] + L.enc.c1b + oc]), amax + amax_idx(AMAX_A1, e));
    return;
  }
  const int e = bid / WA_BLOCKS;
  int b = bid % WA_BLOCKS, t = 0;
  while (b >= WA_NBLK[t]) b -= WA_NBLK[t], ++t;  // t = slot
  const int nb = WA_NBLK[t];
  const int64_t off = t == AMAX_WL ? L.enc.lw : t == AMAX_W2 ? L.enc.c2w : t == AMAX_W3 ? L.enc.c3w : L.enc.c1w;
  const int64_t cnt = t == AMAX_WL ? (int64_t)FEAT * FLAT : t == AMAX_W2 ? (int64_t)C2_OC * C2_K : t == AMAX_W3 ? (int64_t)C3_OC * C3_K
                                                                                                              : (int64_t)C1_OC * L.C * 64;
  const float* src = params + L.enc_base[e] + off;
  float m = 0.0f;
  const int64_t stride = (int64_t)nb * 256;
  int64_t i = (int64_t)b * 256 + threadIdx.x;
  for (; i + 7 * stride < cnt; i += 8 * stride) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, fabsf(x[u]));
  }
  for (; i < cnt; i += stride) m = fmaxf(m, fabsf(src[i]));
  // one atomic per WORKGROUP: every wave of the grid looks at the (freshly zeroed) slot at the same moment, so the "look first" of
  // amax_update saves nothing here and 2,048 atomics on eight addresses took 35 of an earlier form's 42 us
  __shared__ float wm[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax((unsigned*)(amax + amax_idx(t, e)), __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// Every derived weight layout in ONE launch (round 6: six launches of 4-10 us and their boundaries): block ranges in the order below
constexpr int PK_D3 = 4 * 9 * 64 * 16 / 256, PK_D2 = 2 * 8 * 2 * 64 * 16 / 256, PK_C3 = 8 * 5 * 64 * 16 / 256, PK_C2 = 64 * 32 * 16 / 256;
constexpr int PK_FC = (FLAT / 32) * (FEAT / 32);
static_assert(FLAT % 32 == 0 && FEAT % 32 == 0, "whole 32 x 32 tiles");
__global__ __launch_bounds__(256) void pack_all_planes_kernel(const float* __restrict__ params, ParamLayout L, unsigned short* __restrict__ wd3b,
                                                             unsigned short* __restrict__ wd2b, unsigned short* __restrict__ wp3b,
                                                             unsigned short* __restrict__ wp2b, unsigned short* __restrict__ wlb,
                                                             unsigned short* __restrict__ wdlb, unsigned short* __restrict__ wp1b,
                                                             const float* __restrict__ amax) {
  __shared__ unsigned short tile[NPL][32][33];
  int b = blockIdx.x;
  const int ne = L.NE;
  if (b < PK_FC * ne) {  // the dense layer first: its 3,136 workgroups are the bulk
    const int e = b / PK_FC, r = b % PK_FC;
    pack_fc_planes_body(params, L, wlb, wdlb, amax, tile, r % (FLAT / 32), r / (FLAT / 32), e);
    return;
  }
  b -= PK_FC * ne;
  if (b < PK_D3 * ne) return pack_dgrad3_planes_body(params, L, wd3b, amax, b % PK_D3, b / PK_D3);
  b -= PK_D3 * ne;
  if (b < PK_D2 * ne) return pack_dgrad2_planes_body(params, L, wd2b, amax, b % PK_D2, b / PK_D2);
  b -= PK_D2 * ne;
  if (b < PK_C3 * ne) return pack_conv3_planes_body(params, L, wp3b, amax, b % PK_C3, b / PK_C3);
  b -= PK_C3 * ne;
  if (b < PK_C2 * ne) return pack_conv2_planes_body(params, L, wp2b, amax, b % PK_C2, b / PK_C2);
  b -= PK_C2 * ne;
  pack_conv1_planes_body(params, L, wp1b, amax, b);
}

void launch_pack_weights(const Workspace& w, const ParamLayout& L, const float* params, hipStream_t st) {
  static_assert(AMAX_WL == 0 && AMAX_W2 == 1 && AMAX_W3 == 2 && AMAX_W1 == 3 && AMAX_A1 == 4 && AMAX_FIRST_ACT == 5,
                "weight slots and the bound of a1 come first");
  (void)hipMemsetAsync(w.amax, 0, AMAX_FIRST_ACT * 2 * sizeof(float), st);
  hipLaunchKernelGGL(weights_amax_kernel, dim3((WA_BLOCKS + A1B_BLOCKS) * L.NE), dim3(256), 0, st, params, L, w.amax);
  const int c1_blocks = (L.C * 4 * 2 * 32 * L.NE * 8 + 255) / 256;
  hipLaunchKernelGGL(pack_all_planes_kernel, dim3((PK_FC + PK_D3 + PK_D2 + PK_C3 + PK_C2) * L.NE + c1_blocks), dim3(256), 0, st, params, L,
                     w.wd3b, w.wd2b, w.wp3b, w.wp2b, w.wlb, w.wdlb, w.wp1b, w.amax);
}

// --------------------------------------------------------------------------------------------
// grads[off_e + i] = sum_s part[s][e][i]   (fixed order).  The slabs are summed in DOUBLE and rounded once (round 4): up to 1,024
// fp32 partial sums per element added one after the other in fp32 cost a few 1e-7 of the result, as much as the matrix pipe's own
// fp32 accumulation inside a slab; in double this stage adds nothing (the kernels are HBM-bound: 20 us either way).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nsplit, int64_t count,
                                                              float* __restrict__ grads, int64_t off0, int64_t off1) {
  const int e = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  int sp = 0;
  for (; sp + 8 <= nsplit; sp += 8) {  // slab order; the loads of a round are requested before its adds
    float x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = part[((int64_t)(sp + t) * 2 + e) * count + i];
#pragma unroll
    for (int t = 0; t < 8; ++t) s += (double)x[t];
  }
  for (; sp < nsplit; ++sp) s += (double)part[((int64_t)sp * 2 + e) * count + i];
  grads[(e ? off1 : off0) + i] = (float)s;
}
// Many slabs of few elements (conv1: 1,024 slabs x 8,224 values): 16 lane groups walk the slabs in
// parallel (group g takes slabs g, g+16, ...), one lane then adds the 16 partial sums in group order.
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ part, int nsplit, int64_t count,
                                                                   float* __restrict__ grads, int64_t off0, int64_t off1) {
  __shared__ double red[16][17];
  const int e = blockIdx.y;
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + el;
  double s = 0.0;
  if (i < count)
    for (int sp = g; sp < nsplit; sp += 16) s += (double)part[((int64_t)sp * 2 + e) * count + i];
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && i < count) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][el];
    grads[(e ? off1 : off0) + i] = (float)t;
  }
}
void launch_reduce_partials(const float* part, int nsplit, int64_t count, int ne, float* grads, int64_t off0,
                            int64_t off1, hipStream_t st) {
  if (nsplit >= 64 && count < 65536)
    hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)((count + 15) / 16), ne), dim3(256), 0, st, part, nsplit,
                       count, grads, off0, off1);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((count + 255) / 256), ne), dim3(256), 0, st, part, nsplit,
                       count, grads, off0, off1);
}

// --------------------------------------------------------------------------------------------
// grad-norm partials -> clip coefficient -> Adam (actor lr for the first n_actor parameters)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = (double)g[i];
    s += x * x;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

struct AdamArgs {
  float lr_step[2];   // lr / (1 - beta1^t) per group
  float bc2_sqrt;     // sqrt(1 - beta2^t)
  float beta1, beta2, eps;
  float max_norm;
  int clip;
  float v_theta, ent_theta;
};

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, int64_t n_actor,
                                                        const double* __restrict__ part, int nparts, AdamArgs a) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(s);
      float coef = a.max_norm / (norm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
      coef = fminf(coef, 1.0f);
      if (!a.clip) coef = 1.0f;
      s_coef = coef;
      if (blockIdx.x == 0) {
        g[n + 4] = norm;
        g[n + 5] = coef;
        g[n + 3] = g[n + 0] + g[n + 1] * a.v_theta - g[n + 2] * a.ent_theta;  // PpoTotalLoss, ppo.py:108
      }
    }
  }
  __syncthreads();
  const float coef = s_coef;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.0f - a.beta1);            // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * a.beta2 + ((1.0f - a.beta2) * gi) * gi;  // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    const float step = (i < n_actor) ? a.lr_step[0] : a.lr_step[1];
    p[i] = p[i] + (-step) * (mi / denom);              // addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mi;
    v[i] = vi;
    g[i] = gi;  // clip_grad_norm_ scales .grad in place
  }
}

void launch_clip_adam(const ddrl_config& cfg, const ParamLayout& L, const Workspace& w, float* params, float* grads,
                      float* m, float* v, int64_t step, hipStream_t st) {
  hipLaunchKernelGGL(sqnorm_kernel, dim3(NORM_WG), dim3(256), 0, st, grads, L.n_params, w.npart);
  AdamArgs a;
  const double bc1 = 1.0 - pow((double)cfg.adam_beta1, (double)step);
  const double bc2 = 1.0 - pow((double)cfg.adam_beta2, (double)step);
  if (L.NE == 1) {  // shared prenet: self.optim = Adam(self.parameters(), LEARNING_RATE) (ppo.py:39,110-117)
    a.lr_step[0] = a.lr_step[1] = (float)((double)cfg.learning_rate / bc1);
  } else {
    a.lr_step[0] = (float)((double)cfg.actor_lr / bc1);
    a.lr_step[1] = (float)((double)cfg.critic_lr / bc1);
  }
  a.bc2_sqrt = (float)sqrt(bc2);
  a.beta1 = cfg.adam_beta1;
  a.beta2 = cfg.adam_beta2;
  a.eps = cfg.adam_eps;
  a.max_norm = cfg.clip_grad_norm;
  a.clip = cfg.clip_grad;
  a.v_theta = cfg.v_loss_theta;
  a.ent_theta = cfg.ent_loss_theta;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(2048), dim3(256), 0, st, params, grads, m, v, L.n_params, L.n_actor, w.npart,
                     NORM_WG, a);
}

// --------------------------------------------------------------------------------------------
// GAE reverse scan, one lane per env column, reference operation order (no fma contraction)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gae_kernel(const float* __restrict__ values, const float* __restrict__ rewards,
                                                 const uint8_t* __restrict__ dones, int T, int N, float gamma, float landa,
                                                 float* __restrict__ adv, float* __restrict__ ret) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const float gl = __fmul_rn(gamma, landa);  // (discounts * landa)
  float g = 0.0f;
  float nv = values[(int64_t)T * N + n];
#pragma unroll 8
  for (int t = T - 1; t >= 0; --t) {
    const int64_t i = (int64_t)t * N + n;
    const float d = (float)(uint8_t)(1 - dones[i]);  // (1 - dones) evaluated in uint8
    const float v = values[i];
    const float r = rewards[i];
    g = __fmul_rn(g, d);                                        // rewards_sum *= (1 - dones)
    const float t1 = __fmul_rn(gl, g);                          // discounts*landa*rewards_sum
    const float t2 = __fmul_rn(__fmul_rn(gamma, nv), d);        // discounts*next_v*(1-dones)
    const float t3 = __fadd_rn(__fsub_rn(t2, v), r);            // ... - values + rewards
    g = __fadd_rn(t1, t3);
    nv = v;
    ret[i] = __fadd_rn(v, g);
    adv[i] = g;
  }
}
void launch_gae(const float* values, const float* rewards, const uint8_t* dones, int T, int N, float gamma, float landa,
                float* adv, float* ret, hipStream_t st) {
  hipLaunchKernelGGL(gae_kernel, dim3((N + 63) / 64), dim3(64), 0, st, values, rewards, dones, T, N, gamma, landa, adv,
                     ret);
}

// --------------------------------------------------------------------------------------------
// Status.update_reward_status (USTC_lab/agent/statistics.py:118-123) over the T steps of a rollout: one lane per env, forward in
// time, the reference's operation order (sum += r; episode = episode * (1 - d) + sum * d; sum *= (1 - d)); the running sum and the
// latest finished episode's return persist in rsum / rep between calls
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void episode_returns_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                                                             int T, int N, float* __restrict__ rsum, float* __restrict__ rep,
                                                             float* __restrict__ trace, int* __restrict__ finished) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  float s = rsum[n], e = rep[n];
  int fin = finished ? finished[n] : 0;
#pragma unroll 8
  for (int t = 0; t < T; ++t) {
    const int64_t i = (int64_t)t * N + n;
    const float d = dones[i] ? 1.0f : 0.0f;
    const float k = __fsub_rn(1.0f, d);
    s = __fadd_rn(s, rewards[i]);
    e = __fadd_rn(__fmul_rn(e, k), __fmul_rn(s, d));
    s = __fmul_rn(s, k);
    fin += dones[i] ? 1 : 0;
    if (trace) trace[i] = e;
  }
  rsum[n] = s;
  rep[n] = e;
  if (finished) finished[n] = fin;
}
void launch_episode_returns(const float* rewards, const uint8_t* dones, int T, int N, float* rsum, float* rep, float* trace,
                            int* finished, hipStream_t st) {
  hipLaunchKernelGGL(episode_returns_kernel, dim3((N + 63) / 64), dim3(64), 0, st, rewards, dones, T, N, rsum, rep, trace, finished);
}

// float32(u8/255.0) table via the function the conv1 loaders use (engine2.h)
__global__ void fill_lut_kernel(float* lut) { lut[threadIdx.x] = u8_unit(threadIdx.x); }
void launch_fill_lut(float* lut, hipStream_t st) { hipLaunchKernelGGL(fill_lut_kernel, dim3(1), dim3(256), 0, st, lut); }

}  // namespace ddrl
