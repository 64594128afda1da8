// extern "C" operator-level entry points (include/ddrl.h, "ddrl_op_*"): generic convolution,
// max-pool, dense layer, Gaussian / categorical heads on caller-owned buffers and a stand-alone
// clip + Adam step.  The Python host composes the non-Atari encoders of the reference from them
// (ddrl4nav_amd/nn/generic.py), mirroring USTC_lab/nn/nav_encoder.py and mlp_encoder.py.
#include "kernels.h"
#include "ops.h"

using namespace ddrl;

static int32_t op_check() { return hipGetLastError() == hipSuccess ? DDRL_OK : DDRL_ERR_HIP; }

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static bool fill_geom(const ddrl_conv_desc* d, ConvGeom& g) {
  if (!d) return false;
  g.n = d->n; g.cin = d->cin; g.h = d->h; g.w = d->w; g.cout = d->cout; g.kh = d->kh; g.kw = d->kw;
  g.stride = d->stride; g.pad_h = d->pad_h; g.pad_w = d->pad_w;
  if (!conv_geom_fill(g)) return false;
  g.in_sn = d->in_sn > 0 ? d->in_sn : (int64_t)g.cin * g.h * g.w;
  g.out_sn = d->out_sn > 0 ? d->out_sn : (int64_t)g.cout * g.oh * g.ow;
  return true;
}

// the direct kernels stage whole planes with 16-byte loads: strides and bases must allow that
static bool direct_ok(const ConvGeom& g, const void* in_side, const void* out_side) {
  return (g.in_sn & 3) == 0 && (g.out_sn & 3) == 0 && aligned16(in_side) && aligned16(out_side);
}

// regions of a layer's packed buffer: 0-4 gather layouts + tables (gconv.hip), 5 / 6 the specialised forward / data-gradient layouts
// of the layers that have them (pconv.hip / fconv.hip fp16 planes + header, c1d.hip transposed weights).  Read-only after ddrl_op_conv_pack;
// nothing in it depends on the batch size of a call (the per-sample scales of a launch live in the caller's scales_scratch).
struct PackView {
  int64_t off[7], total;
};
static PackView pack_view(const ConvGeom& g) {
  int64_t sz[7];
  conv_pack_sizes(g, sz);
  sz[5] = sz[6] = 0;
  if (conv_has_planes(g)) sz[5] = sz[6] = conv_planes_pack_floats(g);
  if (conv_has_c1d(g)) sz[5] = sz[6] = conv_c1d_pack_floats(g);  // c1d.hip: transposed weights for the scalar cache
  if (conv_has_first(g)) {  // fconv.hip: forward weight planes only (a first layer has no data gradient)
    sz[5] = conv_first_pack_floats(g);
    sz[6] = 0;
  }
  PackView v;
  int64_t o = 0;
  for (int i = 0; i < 7; ++i) {
    v.off[i] = o;
    o += align_up(sz[i], 64);
  }
  v.total = o;
  return v;
}

extern "C" {

int32_t ddrl_op_conv_out_shape(const ddrl_conv_desc* d, int32_t* oh, int32_t* ow) {
  ConvGeom g;
  if (!fill_geom(d, g) || !oh || !ow) return DDRL_ERR_INVALID_ARG;
  *oh = g.oh;
  *ow = g.ow;
  return DDRL_OK;
}

int32_t ddrl_op_conv_pack_floats(const ddrl_conv_desc* d, int64_t* floats) {
  ConvGeom g;
  if (!fill_geom(d, g) || !floats) return DDRL_ERR_INVALID_ARG;
  *floats = pack_view(g).total;
  return DDRL_OK;
}

int32_t ddrl_op_conv_pack(const ddrl_conv_desc* d, const float* w, float* packed, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !w || !packed || !aligned16(packed)) return DDRL_ERR_INVALID_ARG;
  const PackView v = pack_view(g);
  launch_conv_pack(g, w, packed + v.off[0], (int2*)(packed + v.off[1]), packed + v.off[2], (int2*)(packed + v.off[3]),
                   (int*)(packed + v.off[4]), (hipStream_t)stream);
  if (conv_has_c1d(g))
    launch_conv_c1d_pack(g, w, packed + v.off[5], packed + v.off[6], (hipStream_t)stream);
  else if (conv_has_first(g))
    launch_conv_first_pack(g, w, packed + v.off[5], (hipStream_t)stream);
  else if (conv_has_planes(g))
    launch_conv_planes_pack(g, w, packed + v.off[5], packed + v.off[6], (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_conv_scratch_floats(const ddrl_conv_desc* d, int64_t* floats) {
  ConvGeom g;
  if (!fill_geom(d, g) || !floats) return DDRL_ERR_INVALID_ARG;
  *floats = conv_has_planes(g) ? (int64_t)g.n : 0;
  return DDRL_OK;
}

int32_t ddrl_op_conv_ws_floats(const ddrl_conv_desc* d, int64_t* floats) {
  ConvGeom g;
  if (!fill_geom(d, g) || !floats) return DDRL_ERR_INVALID_ARG;
  int splits = conv_wgrad_splits(g);
  if (conv_planes_wgrad_splits(g) > splits) splits = conv_planes_wgrad_splits(g);
  if (conv_first_wgrad_splits(g) > splits) splits = conv_first_wgrad_splits(g);
  if (conv_c1d_wgrad_splits(g) > splits) splits = conv_c1d_wgrad_splits(g);
  // + the per-sample scales of the plane kernels (pconv.hip): 2 n floats behind the slabs
  *floats = (int64_t)splits * ((int64_t)g.cout * g.cin * g.kh * g.kw + g.cout) + (conv_has_planes(g) ? 2 * (int64_t)g.n + 64 : 0);
  return DDRL_OK;
}

int32_t ddrl_op_conv_forward(const ddrl_conv_desc* d, const float* in, const float* packed, const float* bias, int32_t act,
                             float* out, float* scales_scratch, float* out_amax, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !in || !packed || !bias || !out || act < 0 || act > 1) return DDRL_ERR_INVALID_ARG;
  const PackView v = pack_view(g);
  if (conv_has_planes(g) && direct_ok(g, in, out) && !scales_scratch) return DDRL_ERR_INVALID_ARG;  // g.n floats (ddrl_op_conv_scratch_floats)
  if (out_amax && !conv_has_c1d(g)) {
    // every other kernel family: the magnitudes come from a pass over the output just written (needs 16-byte loads of whole samples)
    const int64_t elems = (int64_t)g.cout * g.oh * g.ow;
    if ((elems & 3) || (g.out_sn & 3) || !aligned16(out)) return DDRL_ERR_INVALID_ARG;
  }
  if (conv_has_c1d(g))
    launch_conv_c1d_fwd(g, in, packed + v.off[5], bias, act, out, out_amax, (hipStream_t)stream);
  else if (conv_has_first(g) && direct_ok(g, in, out))
    launch_conv_first_fwd(g, in, packed + v.off[5], bias, act, out, (hipStream_t)stream);
  else if (conv_has_first(g))
    launch_conv_fwd(g, in, packed + v.off[0], (const int2*)(packed + v.off[1]), bias, act, out, (hipStream_t)stream);
  else if (conv_has_planes(g) && direct_ok(g, in, out))
    launch_conv_planes_fwd(g, in, packed + v.off[5], scales_scratch, bias, act, out, (hipStream_t)stream);
  else   // any geometry, any stride / alignment (also the plane layers' when a strided view rules out their 16-byte loads)
    launch_conv_fwd(g, in, packed + v.off[0], (const int2*)(packed + v.off[1]), bias, act, out, (hipStream_t)stream);
  if (out_amax && !conv_has_c1d(g)) launch_sample_amax(out, g.out_sn, g.cout * g.oh * g.ow, g.n, out_amax, (hipStream_t)stream, 1);
  return op_check();
}

int32_t ddrl_op_conv_has_forward_pool(const ddrl_conv_desc* d) {
  ConvGeom g;
  if (!fill_geom(d, g)) return 0;
  return (conv_has_first(g) || (conv_has_planes(g) && conv_planes_has_pool(g))) ? 1 : 0;
}

int32_t ddrl_op_sample_amax(const float* x, int64_t sn, int32_t elems, int32_t n, float* amax, void* stream) {
  if (!x || !amax || n < 1 || elems < 4 || (elems & 3) || (sn & 3) || sn < elems || !aligned16(x)) return DDRL_ERR_INVALID_ARG;
  launch_sample_amax(x, sn, elems, n, amax, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_conv_pooled_uses_scales(const ddrl_conv_desc* d) {
  ConvGeom g;
  if (!fill_geom(d, g)) return 0;
  return (!conv_has_first(g) && conv_has_planes(g) && conv_planes_has_pool(g)) ? 1 : 0;
}

int32_t ddrl_op_conv_forward_pool(const ddrl_conv_desc* d, const float* in, const float* packed, const float* bias, float* pooled,
                                  uint8_t* code, const float* in_amax, float* scales_scratch, float* out_amax, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !in || !packed || !bias || !pooled || !code) return DDRL_ERR_INVALID_ARG;
  if ((g.in_sn & 3) || !aligned16(in)) return DDRL_ERR_INVALID_ARG;
  if (!conv_has_first(g) && conv_has_planes(g) && !in_amax && !scales_scratch) return DDRL_ERR_INVALID_ARG;
  const PackView v = pack_view(g);
  if (conv_has_first(g))
    launch_conv_first_fwd_pool(g, in, packed + v.off[5], bias, pooled, code, out_amax, (hipStream_t)stream);
  else if (conv_has_planes(g) && conv_planes_has_pool(g))
    launch_conv_planes_fwd_pool(g, in, packed + v.off[5], scales_scratch, in_amax, bias, pooled, code, out_amax, (hipStream_t)stream);
  else
    return DDRL_ERR_UNSUPPORTED;  // the caller runs ddrl_op_conv_forward + ddrl_op_maxpool2_forward_idx
  return op_check();
}

int32_t ddrl_op_conv_dgrad_pooled(const ddrl_conv_desc* d, const float* dpool, const uint8_t* code, const float* packed, float* din,
                                  const float* dpool_amax, float* scales_scratch, float* din_amax, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !dpool || !code || !packed || !din || (!dpool_amax && !scales_scratch)) return DDRL_ERR_INVALID_ARG;
  if ((g.in_sn & 3) || !aligned16(din) || !aligned16(dpool)) return DDRL_ERR_INVALID_ARG;
  if (!(conv_has_planes(g) && conv_planes_has_pool(g))) return DDRL_ERR_UNSUPPORTED;
  const PackView v = pack_view(g);
  launch_conv_planes_dgrad_pooled(g, dpool, code, packed + v.off[6], scales_scratch, dpool_amax, din, din_amax, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_conv_wgrad_pooled(const ddrl_conv_desc* d, const float* in, const float* dpool, const uint8_t* code, const float* packed,
                                  float* ws, float* dw, float* db, const float* in_amax, const float* dpool_amax, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !in || !dpool || !code || !packed || !ws || !dw || !db) return DDRL_ERR_INVALID_ARG;
  if ((g.in_sn & 3) || !aligned16(in) || !aligned16(dpool) || ((uintptr_t)code & 1)) return DDRL_ERR_INVALID_ARG;
  if (conv_has_first(g))
    launch_conv_first_wgrad_pooled(g, in, dpool, code, ws, dw, db, (hipStream_t)stream);
  else if (conv_has_planes(g) && conv_planes_has_pool(g))
    launch_conv_planes_wgrad_pooled(g, in, dpool, code, in_amax, dpool_amax, ws, dw, db, (hipStream_t)stream);
  else
    return DDRL_ERR_UNSUPPORTED;
  return op_check();
}

int32_t ddrl_op_conv_dgrad(const ddrl_conv_desc* d, const float* dz, const float* packed, float* din, float* scales_scratch, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !dz || !packed || !din) return DDRL_ERR_INVALID_ARG;
  const PackView v = pack_view(g);
  if (conv_has_planes(g) && direct_ok(g, din, dz) && !conv_has_c1d_backward(g) && !scales_scratch) return DDRL_ERR_INVALID_ARG;
  if (conv_has_c1d_backward(g))
    launch_conv_c1d_dgrad(g, dz, packed + v.off[6], din, (hipStream_t)stream);
  else if (conv_has_planes(g) && direct_ok(g, din, dz))
    launch_conv_planes_dgrad(g, dz, packed + v.off[6], scales_scratch, din, (hipStream_t)stream);
  else
    launch_conv_dgrad(g, dz, packed + v.off[2], (const int2*)(packed + v.off[3]), din, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_conv_wgrad(const ddrl_conv_desc* d, const float* in, const float* dz, const float* packed, float* ws,
                           float* dw, float* db, void* stream) {
  ConvGeom g;
  if (!fill_geom(d, g) || !in || !dz || !packed || !ws || !dw || !db) return DDRL_ERR_INVALID_ARG;
  if (g.oh * g.ow < 32) return DDRL_ERR_UNSUPPORTED;
  const PackView v = pack_view(g);
  if (conv_has_c1d_backward(g))
    launch_conv_c1d_wgrad(g, in, dz, ws, dw, db, (hipStream_t)stream);
  else if (conv_has_first(g) && direct_ok(g, in, dz))
    launch_conv_first_wgrad(g, in, dz, ws, dw, db, (hipStream_t)stream);
  else if (conv_has_planes(g) && direct_ok(g, in, dz))
    launch_conv_planes_wgrad(g, in, dz, ws, dw, db, (hipStream_t)stream);
  else
    launch_conv_wgrad(g, in, dz, (const int*)(packed + v.off[4]), ws, dw, db, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_maxpool2_forward(const float* in, int64_t planes, int32_t h, int32_t w, float* out, void* stream) {
  if (!in || !out || planes < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || ((uintptr_t)in & 7)) return DDRL_ERR_INVALID_ARG;
  launch_maxpool2_fwd(in, planes, h, w, out, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_maxpool2_relu_backward(const float* a, const float* dpool, int64_t planes, int32_t h, int32_t w, float* dz,
                                       void* stream) {
  if (!a || !dpool || !dz || planes < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || (((uintptr_t)a | (uintptr_t)dz) & 7))
    return DDRL_ERR_INVALID_ARG;
  launch_maxpool2_relu_bwd(a, dpool, planes, h, w, dz, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_maxpool2_forward_idx(const float* in, int64_t planes, int32_t h, int32_t w, float* out, uint8_t* code, void* stream) {
  if (!in || !out || !code || planes < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || ((uintptr_t)in & 7)) return DDRL_ERR_INVALID_ARG;
  launch_maxpool2_fwd_idx(in, planes, h, w, out, code, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_maxpool2_backward_idx(const float* dpool, const uint8_t* code, int64_t planes, int32_t h, int32_t w, float* dz,
                                      void* stream) {
  if (!dpool || !code || !dz || planes < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || ((uintptr_t)dz & 15)) return DDRL_ERR_INVALID_ARG;
  launch_maxpool2_bwd_idx(dpool, code, planes, h, w, dz, (hipStream_t)stream);
  return op_check();
}

// ---- dense layer --------------------------------------------------------------------------------
static bool lin_ok(int32_t n, int32_t K, int32_t N) { return n >= 1 && K >= 1 && N >= 4 && (N & 3) == 0; }

int32_t ddrl_op_linear_pack_floats(int32_t K, int32_t N, int64_t* wt_floats, int64_t* wn_floats) {
  if (!lin_ok(1, K, N) || !wt_floats || !wn_floats) return DDRL_ERR_INVALID_ARG;
  *wt_floats = (int64_t)((K + 31) / 32 * 32) * N;
  *wn_floats = (int64_t)N * ((K + 3) / 4 * 4);
  if (linear_has_planes(K, N)) {  // the fp16 plane layouts of plin.hip follow the f32 layouts (both regions start 16-byte aligned)
    *wt_floats += linear_planes_fwd_floats(K, N);
    *wn_floats += linear_planes_dgrad_floats(K, N);
  }
  return DDRL_OK;
}

int32_t ddrl_op_linear_pack(const float* w, int32_t K, int32_t N, float* wt, float* wn, void* stream) {
  if (!lin_ok(1, K, N) || !w || !wt || !wn || !aligned16(wt) || !aligned16(wn)) return DDRL_ERR_INVALID_ARG;
  launch_linear_pack(w, K, N, wt, wn, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_linear_uses_planes(int32_t n, int32_t K, int32_t N) { return lin_ok(n, K, N) && linear_uses_planes(n, K, N) ? 1 : 0; }

int32_t ddrl_op_row_amax(const float* x, int64_t ld, int32_t width, int32_t n, float* amax, int32_t accumulate, void* stream) {
  if (!x || !amax || n < 1 || width < 1 || (ld & 3) || ld < (width + 3) / 4 * 4 || !aligned16(x)) return DDRL_ERR_INVALID_ARG;
  launch_row_amax(x, ld, width, n, amax, accumulate ? 1 : 0, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_linear_forward(const float* in, int64_t ld_in, const float* wt, const float* bias, int32_t act, float* out,
                               int64_t ld_out, int32_t n, int32_t K, int32_t N, float* ws, const float* in_amax, void* stream) {
  if (!lin_ok(n, K, N) || !in || !wt || !bias || !out || act < 0 || act > 1) return DDRL_ERR_INVALID_ARG;
  if ((ld_in & 3) || ld_in < (K + 3) / 4 * 4 || ld_out < N || !aligned16(in) || !aligned16(wt)) return DDRL_ERR_INVALID_ARG;
  launch_linear_fwd(in, ld_in, wt, bias, out, ld_out, n, K, N, act, ws, in_amax, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_linear_dgrad(const float* dout, int64_t ld_dout, const float* wn, const float* mask_src, int64_t ld_mask,
                             float* din, int64_t ld_din, int32_t n, int32_t K, int32_t N, float* ws, const float* dout_amax,
                             float* din_amax, int32_t amax_lo, int32_t amax_hi, void* stream) {
  if (!lin_ok(n, K, N) || !dout || !wn || !din) return DDRL_ERR_INVALID_ARG;
  if ((ld_dout & 3) || ld_dout < N || ld_din < K || !aligned16(dout) || !aligned16(wn)) return DDRL_ERR_INVALID_ARG;
  if (din_amax) {
    if (amax_hi <= 0) amax_hi = K;   // default: every column
    if (amax_lo < 0 || amax_lo >= amax_hi || amax_hi > K || (amax_lo & 3) || (ld_din & 3) || !aligned16(din)) return DDRL_ERR_INVALID_ARG;
  }
  launch_linear_dgrad(dout, ld_dout, wn, mask_src, ld_mask, din, ld_din, n, K, N, ws, dout_amax, din_amax, amax_lo, amax_hi, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_linear_ws_floats(int32_t n, int32_t K, int32_t N, int64_t* floats) {
  if (!lin_ok(n, K, N) || !floats) return DDRL_ERR_INVALID_ARG;
  int64_t wg = (int64_t)linear_wgrad_splits(n, K, N) * ((int64_t)N * K + N);   // non-decreasing in n
  if (linear_has_planes(K, N)) {  // plin.hip's split counts (non-decreasing in n as well)
    const int64_t wp = (int64_t)linear_planes_wgrad_splits(n, K, N) * ((int64_t)N * K + N);
    if (wp > wg) wg = wp;
  }
  // split-K partials of the forward: splits(n') * n' * N floats for a launch of n' <= n samples.  FEWER samples take MORE splits
  // (the split count fills the chip), so the product is NOT largest at n' = n: with tn = ceil(N / 128) column tiles,
  // splits(n') <= min(cap_K, ceil(512 / (tn ceil(n' / 128)))) gives splits(n') n' <= min(cap_K n, 65536 / tn + n).  (Sizing by n alone
  // once let a 40,000-sample launch of a 65,536-sample layer write 2 x 40,000 x N partials into 1 x 65,536 x N floats.)
  const int64_t tn = (N + 127) / 128;
  int64_t cap_k = ((K + 31) / 32) / 8;
  if (cap_k < 1) cap_k = 1;
  const int64_t by_k = cap_k * (int64_t)n, by_fill = 65536 / tn + (int64_t)n + 128;
  const int64_t fw = (by_k < by_fill ? by_k : by_fill) * N;
  // + the per-row plane scales of plin.hip: n floats (rounded up to 64) in front of the forward's partials, 2 n behind the slabs
  *floats = (wg > fw ? wg : fw) + 2 * (int64_t)n + 128;
  return DDRL_OK;
}

int32_t ddrl_op_linear_wgrad(const float* in, int64_t ld_in, const float* dout, int64_t ld_dout, float* ws, float* dw,
                             float* db, int32_t n, int32_t K, int32_t N, const float* in_amax, const float* dout_amax, void* stream) {
  if (!lin_ok(n, K, N) || !in || !dout || !ws || !dw || !db) return DDRL_ERR_INVALID_ARG;
  if ((ld_in & 3) || (ld_dout & 3) || ld_in < (K + 3) / 4 * 4 || ld_dout < N || !aligned16(in) || !aligned16(dout))
    return DDRL_ERR_INVALID_ARG;
  launch_linear_wgrad(in, ld_in, dout, ld_dout, ws, n, K, N, dw, db, in_amax, dout_amax, (hipStream_t)stream);
  return op_check();
}

// ---- heads + optimiser on caller-owned arenas --------------------------------------------------
static bool heads_ok(const ddrl_heads_desc* d) {
  if (!d || d->n_params < 1) return false;
  if (d->continuous) return d->n_actions >= 1 && d->n_actions <= 8;
  return d->n_actions >= 2 && d->n_actions <= 18;
}
static ParamLayout cat_layout(const ddrl_heads_desc* d) {
  ParamLayout L = make_layout(d->n_actions, 4, d->shared != 0);  // encoder offsets are unused by the head kernels
  L.actor_w = d->actor_w; L.actor_b = d->actor_b; L.critic_w = d->critic_w; L.critic_b = d->critic_b;
  L.n_params = d->n_params;
  return L;
}
static GaussLayout gauss_layout(const ddrl_heads_desc* d) {
  GaussLayout L;
  L.D = d->n_actions; L.shared = d->shared != 0;
  L.actor_w = d->actor_w; L.actor_b = d->actor_b; L.log_std = d->log_std; L.critic_w = d->critic_w;
  L.critic_b = d->critic_b; L.n_params = d->n_params;
  return L;
}
struct HeadsWs {
  float *dlogits, *dvalue, *hpart;
  int64_t total;
};
static HeadsWs heads_ws(const ddrl_heads_desc* d, int64_t max_n, float* base) {
  HeadsWs w;
  int64_t o = 0;
  auto take = [&](int64_t floats) { float* p = base ? base + o : nullptr; o += align_up(floats, 64); return p; };
  w.dlogits = take(max_n * d->n_actions);
  w.dvalue = take(max_n);
  const int64_t hs = d->continuous ? gauss_hpart_stride(d->n_actions) : hpart_stride(d->n_actions);
  w.hpart = take((int64_t)HEAD_WG * hs);
  w.total = o;
  return w;
}

int32_t ddrl_op_heads_ws_floats(const ddrl_heads_desc* d, int32_t max_n, int64_t* floats) {
  if (!heads_ok(d) || max_n < 1 || !floats) return DDRL_ERR_INVALID_ARG;
  *floats = heads_ws(d, max_n, nullptr).total;
  return DDRL_OK;
}

int32_t ddrl_op_heads_act(const ddrl_heads_desc* d, const float* params, const float* h_actor, const float* h_critic,
                          int32_t n, const float* act_in, uint64_t seed, uint64_t stream_id, float* dist_out, float* value,
                          float* action_out, float* logp_out, void* stream) {
  if (!heads_ok(d) || !params || !h_actor || !h_critic || !value || n < 1) return DDRL_ERR_INVALID_ARG;
  if (!aligned16(h_actor) || !aligned16(h_critic)) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (d->continuous) {
    launch_gauss_act(gauss_layout(d), params, h_actor, h_critic, n, act_in, seed, stream_id, dist_out, value, action_out,
                     logp_out, st);
    return op_check();
  }
  ParamLayout L = cat_layout(d);
  Workspace w{};
  w.h = const_cast<float*>(h_actor);
  ddrl_config cfg;
  ddrl_config_default(&cfg);
  HeadsCall hc{&w, &L, &cfg, params, n, n};
  hc.h_es = d->shared ? 0 : (int64_t)(h_critic - h_actor);
  hc.plain_features = true;
  launch_heads_act(hc, act_in, seed, stream_id, dist_out, value, action_out, logp_out, st);
  return op_check();
}

int32_t ddrl_op_heads_loss(const ddrl_heads_desc* d, const ddrl_config* cfg, const float* params, const float* h_actor,
                           const float* h_critic, int32_t n, const float* actions, const float* old_logps, const float* advs,
                           const float* rets, int64_t B_global, float* dh_actor, float* dh_critic, float* grads, float* ws,
                           void* stream) {
  if (!heads_ok(d) || !cfg || !params || !h_actor || !h_critic || !actions || !old_logps || !advs || !rets || !dh_actor ||
      !grads || !ws || n < 1 || B_global < n)
    return DDRL_ERR_INVALID_ARG;
  if (!d->shared && !dh_critic) return DDRL_ERR_INVALID_ARG;
  if (!aligned16(h_actor) || !aligned16(h_critic) || !aligned16(dh_actor) || !aligned16(ws)) return DDRL_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  HeadsWs hw = heads_ws(d, n, ws);
  const float inv_b = (float)(1.0 / (double)B_global);
  if (d->continuous) {
    launch_gauss_loss(gauss_layout(d), *cfg, params, h_actor, d->shared ? h_actor : h_critic, n, actions, old_logps, advs,
                      rets, inv_b, dh_actor, dh_critic, hw.dlogits, hw.dvalue, hw.hpart, grads, st);
    return op_check();
  }
  ParamLayout L = cat_layout(d);
  Workspace w{};
  w.h = const_cast<float*>(h_actor);
  w.dh = dh_actor;
  w.dlogits = hw.dlogits;
  w.dvalue = hw.dvalue;
  w.hpart = hw.hpart;
  HeadsCall hc{&w, &L, cfg, params, n, n};
  hc.h_es = d->shared ? 0 : (int64_t)(h_critic - h_actor);
  hc.dh_es = d->shared ? 0 : (int64_t)(dh_critic - dh_actor);
  hc.plain_features = true;
  launch_heads_loss(hc, actions, old_logps, advs, rets, inv_b, grads, st);
  return op_check();
}

int32_t ddrl_op_clip_adam(const ddrl_config* cfg, float* params, float* grads, float* m, float* v, int64_t n_params,
                          int64_t n_actor, int32_t shared, int64_t step, void* ws, void* stream) {
  if (!cfg || !params || !grads || !m || !v || !ws || n_params < 1 || n_actor < 0 || n_actor > n_params || step < 1)
    return DDRL_ERR_INVALID_ARG;
  ParamLayout L = make_layout(2, 4, shared != 0);
  L.n_params = n_params;
  L.n_actor = shared ? n_params : n_actor;
  Workspace w{};
  w.npart = (double*)ws;
  launch_clip_adam(*cfg, L, w, params, grads, m, v, step, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_clip_adam_ws_bytes(int64_t* bytes) {
  if (!bytes) return DDRL_ERR_INVALID_ARG;
  *bytes = (int64_t)NORM_WG * sizeof(double);
  return DDRL_OK;
}

int32_t ddrl_op_relu_mask(float* d, int64_t ld_d, const float* act, int64_t ld_act, int32_t n, int32_t width, void* stream) {
  if (!d || !act || n < 1 || width < 1 || ld_d < width || ld_act < width) return DDRL_ERR_INVALID_ARG;
  launch_relu_mask(d, ld_d, act, ld_act, n, width, (hipStream_t)stream);
  return op_check();
}

int32_t ddrl_op_accumulate(float* dst, const float* src, int64_t count, void* stream) {
  if (!dst || !src || count < 1) return DDRL_ERR_INVALID_ARG;
  launch_accumulate(dst, src, count, (hipStream_t)stream);
  return op_check();
}

}  // extern "C"
