// Kernels of the GAIL path: the extra value head PPO carries next to its critic, the
// discriminator's WGAN-style loss terms and its clip + RMSprop step.  All HBM-bound, one wavefront per sample for the
// 512-wide dot products (8 features per lane + a wave64 xor-shuffle butterfly), fixed-order reductions (no atomics).
//
// Reference arithmetic replaced:
//   PPO.add_critic / forward: values = [critic(h) for critic in self._critics]      USTC_lab/nn/ppo.py:61-62,72-75
//   gailv_loss = vlossf(data.values[-1], values[-1].squeeze()); v_loss += it        USTC_lab/nn/ppo.py:101-107
//   Discriminator.learn: mean D(generator batch) - mean D(expert batch)              USTC_lab/nn/GAIL.py:76-80
//   clip_grad_norm_(D.parameters(), WGAN_CLIP_GRAD_NUM); RMSprop(alpha=0.9).step()   USTC_lab/nn/GAIL.py:28,83-84
#include "kernels.h"
#include "ops.h"
#include "ppo_math.h"

namespace ddrl {

constexpr int VH_WG = 256;     // workgroups of the value-head loss kernel (fixed -> deterministic partial sums)
constexpr int VH_WAVES = 4;
constexpr int VH_STRIDE = 576; // floats per workgroup partial: [512 dW][db][loss sum][pad]

// v[b] = w . h[b] + bias.  `w` sits anywhere in a flat fp32 arena (4-byte aligned): scalar loads, once per wave.
__global__ __launch_bounds__(256) void value_head_fwd_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                             const float* __restrict__ h, int64_t ld_h, int n,
                                                             float* __restrict__ value) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  float wc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wc[i] = w[lane * 8 + i];
  const float bc = bias[0];
  for (int b = gw; b < n; b += nw) {
    float hc[8];
    load8(h + (int64_t)b * ld_h + lane * 8, hc);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = __builtin_fmaf(hc[i], wc[i], s);
    const float v = wave_sum(s) + bc;
    if (lane == 0) value[b] = v;
  }
}

// Value loss of one extra head on `rets`, its gradient ADDED into dh (the head reads the features the other heads read:
// shared prenet), the head's own weight / bias gradient and the loss sum as per-workgroup partials.
__global__ __launch_bounds__(VH_WAVES * 64) void value_head_loss_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                                        const float* __restrict__ h, int64_t ld_h, int n,
                                                                        const float* __restrict__ rets, ddrl_config cfg,
                                                                        float gscale, float* __restrict__ dh, int64_t ld_dh,
                                                                        float* __restrict__ part) {
  __shared__ float red[VH_WAVES][FEAT + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * VH_WAVES + wave;
  const int nw = gridDim.x * VH_WAVES;
  float wc[8], gwc[8], gb = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    wc[i] = w[lane * 8 + i];
    gwc[i] = 0.0f;
  }
  const float bc = bias[0];
  double s_v = 0.0;
  for (int b = gw; b < n; b += nw) {
    float hc[8], d[8];
    load8(h + (int64_t)b * ld_h + lane * 8, hc);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = __builtin_fmaf(hc[i], wc[i], s);
    const float v = wave_sum(s) + bc;
    const float gv = value_loss_element(rets[b] - v, cfg, s_v) * gscale;
    load8(dh + (int64_t)b * ld_dh + lane * 8, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d[i] += gv * wc[i];
      gwc[i] = __builtin_fmaf(gv, hc[i], gwc[i]);
    }
    store8(dh + (int64_t)b * ld_dh + lane * 8, d);
    gb += gv;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = gwc[i];
  if (lane == 0) {
    red[wave][FEAT] = gb;
    red[wave][FEAT + 1] = (float)s_v;
  }
  __syncthreads();
  float* out = part + (int64_t)blockIdx.x * VH_STRIDE;
  for (int i = threadIdx.x; i < FEAT + 2; i += VH_WAVES * 64) {
    float t = red[0][i];
#pragma unroll
    for (int q = 1; q < VH_WAVES; ++q) t += red[q][i];  // waves in order
    out[i] = t;
  }
}

// partials -> dw[512], db, and vloss_accum[0] += sum * inv_b * (1/2 for the squared error, ppo.py:57)
__global__ __launch_bounds__(256) void value_head_reduce_kernel(const float* __restrict__ part, int nwg, ddrl_config cfg,
                                                                float inv_b, float* __restrict__ dw, float* __restrict__ db,
                                                                float* __restrict__ vloss_accum) {
  __shared__ double sh[8][RED_OUT];
  if (blockIdx.x == gridDim.x - 1) {  // the loss sum, by one wave
    if (threadIdx.x >= 64) return;
    const double s = wave_sum_partials(part, VH_STRIDE, nwg, FEAT + 1);
    if (threadIdx.x == 0 && vloss_accum) vloss_accum[0] += (float)(s * (double)inv_b * (cfg.smooth_l1_loss ? 1.0 : 0.5));
    return;
  }
  // summed in double, rounded once, in the fixed order of ppo_math.h sum_partials8
  const int i = blockIdx.x * RED_OUT + (threadIdx.x & (RED_OUT - 1));
  const float s = sum_partials8(part, VH_STRIDE, nwg, min(i, FEAT), sh);
  if (threadIdx.x >= RED_OUT || i > FEAT) return;
  if (i < FEAT) {
    if (dw) dw[i] = s;
  } else if (db) {
    db[0] = s;
  }
}

void launch_value_head_fwd(const float* w, const float* b, const float* h, int64_t ld_h, int n, float* value, hipStream_t st) {
  int wgs = (n + 3) / 4;
  if (wgs > 1024) wgs = 1024;
  hipLaunchKernelGGL(value_head_fwd_kernel, dim3(wgs), dim3(256), 0, st, w, b, h, ld_h, n, value);
}

void launch_value_head_loss(const ddrl_config& cfg, bool shared, const float* w, const float* b, const float* h, int64_t ld_h,
                            int n, const float* rets, float inv_b, float* dh, int64_t ld_dh, float* dw, float* db,
                            float* vloss_accum, float* part, hipStream_t st) {
  // shared prenet: total_loss.backward() -> the value gradient carries v_loss_theta (ppo.py:108,111)
  const float gscale = shared ? inv_b * cfg.v_loss_theta : inv_b;
  hipLaunchKernelGGL(value_head_loss_kernel, dim3(VH_WG), dim3(VH_WAVES * 64), 0, st, w, b, h, ld_h, n, rets, cfg, gscale, dh,
                     ld_dh, part);
  hipLaunchKernelGGL(value_head_reduce_kernel, dim3((FEAT + 1 + RED_OUT - 1) / RED_OUT + 1), dim3(256), 0, st, part, VH_WG, cfg, inv_b, dw, db,
                     vloss_accum);
}

// loss[0] (+)= sign * mean(score[:, 0]);  dscore[i][0] = sign / n_total, dscore[i][1..width) = 0 (padded columns).
// One workgroup: the n scores are summed in a fixed tree order.
__global__ __launch_bounds__(256) void wgan_terms_kernel(const float* __restrict__ score, int64_t ld, int n, int64_t n_total,
                                                         float sign, float* __restrict__ dscore, int64_t ld_d, int width,
                                                         float* __restrict__ loss, int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)score[(int64_t)i * ld];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const float g = sign / (float)n_total;
  for (int64_t i = threadIdx.x; i < (int64_t)n * width; i += 256) {
    const int64_t r = i / width, c = i - r * width;
    dscore[r * ld_d + c] = (c == 0) ? g : 0.0f;
  }
  if (threadIdx.x == 0) {
    const float term = (float)((double)sign * red[0] / (double)n_total);
    loss[0] = accumulate ? loss[0] + term : term;
  }
}

// out[c] = sum_i x[i][c], accumulated in double in a fixed tree order and rounded once: the bias gradient of a narrow dense
// layer.  (The discriminator's score bias gets sum_i sign/n from each term of the WGAN loss: the two terms cancel exactly
// only if each sum is correctly rounded, and RMSprop would turn a 6e-8 residue into a step of ~lr.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int n, float* __restrict__ out) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)x[(int64_t)i * ld + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)red[0];
}

struct RmsArgs {
  float lr, alpha, one_minus_alpha, eps, max_norm;  // one_minus_alpha = float(1.0 - alpha) in DOUBLE, as torch casts `value = 1 - alpha`
};

// clip_grad_norm_ over the whole arena, then torch.optim.RMSprop (no momentum, not centered):
//   square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha); avg = square_avg.sqrt().add_(eps); param.addcdiv_(grad, avg, value=-lr)
__global__ __launch_bounds__(256) void clip_rmsprop_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq,
                                                           int64_t n, const double* __restrict__ part, int nparts, RmsArgs a) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(s);
      const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
      s_coef = coef;
      if (blockIdx.x == 0) {
        g[n + 4] = norm;
        g[n + 5] = coef;
      }
    }
  }
  __syncthreads();
  const float coef = s_coef;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float si = sq[i] * a.alpha + (a.one_minus_alpha * gi) * gi;
    const float avg = sqrtf(si) + a.eps;
    p[i] = p[i] + ((-a.lr) * gi) / avg;
    sq[i] = si;
    g[i] = gi;
  }
}

__global__ __launch_bounds__(256) void sqnorm2_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
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

}  // namespace ddrl

using namespace ddrl;

static int32_t g_check() { return hipGetLastError() == hipSuccess ? DDRL_OK : DDRL_ERR_HIP; }
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" {

int32_t ddrl_op_value_head_forward(const float* w, const float* b, const float* h, int64_t ld_h, int32_t n, float* value,
                                   void* stream) {
  if (!w || !b || !h || !value || n < 1 || ld_h < FEAT || (ld_h & 3) || !al16(h)) return DDRL_ERR_INVALID_ARG;
  launch_value_head_fwd(w, b, h, ld_h, n, value, (hipStream_t)stream);
  return g_check();
}

int32_t ddrl_op_value_head_ws_floats(int64_t* floats) {
  if (!floats) return DDRL_ERR_INVALID_ARG;
  *floats = (int64_t)VH_WG * VH_STRIDE;
  return DDRL_OK;
}

int32_t ddrl_op_value_head_loss(const ddrl_config* cfg, int32_t shared, const float* w, const float* b, const float* h,
                                int64_t ld_h, int32_t n, const float* rets, int64_t B_global, float* dh, int64_t ld_dh, float* dw,
                                float* db, float* vloss_accum, float* ws, void* stream) {
  if (!cfg || !w || !b || !h || !rets || !dh || !ws || n < 1 || B_global < n) return DDRL_ERR_INVALID_ARG;
  if (ld_h < FEAT || ld_dh < FEAT || (ld_h & 3) || (ld_dh & 3) || !al16(h) || !al16(dh)) return DDRL_ERR_INVALID_ARG;
  launch_value_head_loss(*cfg, shared != 0, w, b, h, ld_h, n, rets, (float)(1.0 / (double)B_global), dh, ld_dh, dw, db,
                         vloss_accum, ws, (hipStream_t)stream);
  return g_check();
}

int32_t ddrl_op_wgan_terms(const float* score, int64_t ld, int32_t n, int64_t n_total, float sign, float* dscore, int64_t ld_d,
                           int32_t width, float* loss, int32_t accumulate, void* stream) {
  if (!score || !dscore || !loss || n < 1 || n_total < n || ld < 1 || width < 1 || ld_d < width) return DDRL_ERR_INVALID_ARG;
  hipLaunchKernelGGL(wgan_terms_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, score, ld, n, n_total, sign, dscore, ld_d,
                     width, loss, accumulate);
  return g_check();
}

int32_t ddrl_op_colsum(const float* x, int64_t ld, int32_t n, int32_t width, float* out, void* stream) {
  if (!x || !out || n < 1 || width < 1 || ld < width) return DDRL_ERR_INVALID_ARG;
  hipLaunchKernelGGL(colsum_kernel, dim3(width), dim3(256), 0, (hipStream_t)stream, x, ld, n, out);
  return g_check();
}

int32_t ddrl_op_clip_rmsprop(float* params, float* grads, float* square_avg, int64_t n_params, float lr, double alpha, float eps,
                             float max_norm, void* ws, void* stream) {
  if (!params || !grads || !square_avg || !ws || n_params < 1 || !(lr > 0.0f) || !(alpha >= 0.0) || !(max_norm > 0.0f))
    return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm2_kernel, dim3(NORM_WG), dim3(256), 0, st, grads, n_params, (double*)ws);
  // torch.optim.RMSprop: square_avg.mul_(alpha).addcmul_(grad, grad, value=1 - alpha) -- alpha and 1 - alpha are Python doubles, each
  // cast to float32 once: float(1 - 0.9) = 0.1f, whereas 1.0f - 0.9f = 0.100000024f (2.4e-7 off in every step)
  RmsArgs a{lr, (float)alpha, (float)(1.0 - alpha), eps, max_norm};
  hipLaunchKernelGGL(clip_rmsprop_kernel, dim3(2048), dim3(256), 0, st, params, grads, square_avg, n_params, (const double*)ws,
                     NORM_WG, a);
  return g_check();
}

}  // extern "C"
