// v2 convolution weight-gradient kernels on the pipelined f32-MFMA engine (engine2.h).
//
//   part[split][e][oc][tap] = sum over the split's samples and output pixels of
//                             dz[b][oc][pix] * in[b][ic][oy*S+ky][ox*S+kx]
//   rows = oc (MFMA A operand, from the staged dz block), cols = taps (ic,ky,kx) (B operand, read
//   straight out of the staged RAW input planes), reduction = (sample, pixel).
// The two k indices of one 32x32x2 MFMA are the SAME pixel of two consecutive samples, so the
// lane halves differ by a constant (one staged sample) and every operand address is again
// lane_base + compile-time immediate.  The bias gradient (sum of dz) rides along: conv1 adds up the dz
// quads it stages in registers, conv2/conv3 add up their dz rows from LDS once per k-block.  Slabs are laid out like the arena
// (weights then bias) so that one reduce_partials launch finishes both.
//
// Reference: autograd weight/bias gradients of conv1..conv3 (atari_encoder.py:16-18 through
// actor_loss.backward(); v_loss.backward(), ppo.py:122-123).
#include "engine2.h"

namespace ddrl {


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
// conv1, both encoders fused: rows = (e, oc) = 64, cols = 256 taps (4 ch x 8 x 8),
// k-block = (sample pair, output row oy): 20 k-steps (ox).
// ================================================================================================
template <int NE>
struct ConvWgrad1v2 {
  static constexpr int COMMIT_FIRST = 1;  // 7.68 -> 7.38 ms
  static constexpr int THREADS = 256, TM = NE, TN = 2, KSTEPS = 20, ROWS = 32 * NE;  // rows = (e, oc)
  // dz staging: the threads are split by encoder (TPE each) so that the encoder's base pointer is
  // wave-uniform; an encoder's k-block is 2 samples x 32 oc x 5 quads of one output row.
  static constexpr int TPE = 256 / NE, QPE = 2 * 32 * 5, NDZ_J = (QPE + TPE - 1) / TPE;
  static constexpr int LDA = 22;  // dz row stride: even (8-byte aligned quad halves), 22 l mod 64 distinct over 32 lanes
  static constexpr int A_FLOATS = 2 * ROWS * LDA, B_OFF = A_FLOATS, B_FLOATS = 2 * 4 * 672, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int64_t SLAB = 32 * 256 + 32;
  struct Params {
    const uint8_t* frames;
    const float* dz;   // da1 [e][n][32][400], raw output of ConvDgrad2 (no leaky' yet)
    const float* act;  // a1, same layout: dz1 = leaky'(a1) * da1 is formed while staging
    int64_t dz_es;
    float* part;  // [nsplit][e][SLAB]
    int n, nsplit;
  };
  struct Regs {
    f4 dzr[NDZ_J], actr[NDZ_J];
    unsigned im[6];
    bool full;  // wave-uniform: both samples of the pair exist
  };
  int abase[NE], bbase[2], kb_begin, kb_end;
  int split, l31, hi, wc, ew;
  // k-block independent lane parts (bytes) of this thread's source addresses, see ld4_so()
  uint32_t dzoff[NDZ_J], imoff[6];
  int ldsoff[NDZ_J];
  unsigned dz_s1, dz_ok, im_s1;  // bit j: slot j belongs to the pair's second sample / exists
  float bacc[NDZ_J];             // bias gradient: running sums of this thread's dz quads
  static constexpr int aoff(int s) { return s; }
  static constexpr int boff(int s) { return 4 * s; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    const int lane = tid & 63;
    wc = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
    split = blockIdx.y;
    WgradSplit sp;
    sp.set(p.n, p.nsplit, split);
    kb_begin = sp.pair_begin * 20;
    kb_end = sp.pair_end * 20;
    ew = __builtin_amdgcn_readfirstlane(tid / TPE);
    const int te = tid % TPE;
    dz_s1 = dz_ok = im_s1 = 0;
#pragma unroll
    for (int j = 0; j < NDZ_J; ++j) {
      const int idx = te + TPE * j, c = min(idx, QPE - 1);
      const int row = c / 5, q4 = c % 5, smp = row >> 5, oc = row & 31;
      dzoff[j] = (uint32_t)((smp * 12800 + oc * 400 + q4 * 4) * 4);
      ldsoff[j] = (smp * ROWS + ew * 32 + oc) * LDA + q4 * 4;
      dz_s1 |= (unsigned)smp << j;
      dz_ok |= (idx < QPE ? 1u : 0u) << j;
      bacc[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int idx = min(tid + 256 * j, 1343);
      const int smp = idx / 672, rr = idx % 672;
      imoff[j] = (uint32_t)(smp * 28224 + (rr / 168) * 7056 + (rr % 168) * 4);
      im_s1 |= (unsigned)smp << j;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) abase[i] = hi * (ROWS * LDA) + (i * 32 + l31) * LDA;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 64 + j * 32 + l31;
      bbase[j] = B_OFF + hi * 2688 + (col >> 6) * 672 + ((col >> 3) & 7) * 84 + (col & 7);
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int pair = kb / 20, oy = kb % 20;
    r.full = 2 * pair + 1 < p.n;
    pin_offsets(dzoff);
    pin_offsets(imoff);
    // All loads are unconditional (a guarded load makes hipcc branch and wait per load).
    const int64_t sb = ew * p.dz_es + (int64_t)pair * (2 * 12800) + oy * 20;
    const float* dzb = p.dz + sb;
    const float* acb = p.act + sb;
    const uint8_t* fp = p.frames + (int64_t)pair * (2 * 28224) + oy * 336;
    if (r.full) {
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j) {
        r.dzr[j] = ld4_so(dzb, dzoff[j]);
        r.actr[j] = ld4_so(acb, dzoff[j]);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) r.im[j] = ld1u_so(fp, imoff[j]);
    } else {
      // last, half-filled pair of an odd batch: the second sample does not exist -> its slots read
      // the first sample instead and commit() zeroes their dz
      rare_path();
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j) {
        const uint32_t o = dzoff[j] - ((dz_s1 >> j) & 1u) * (12800u * 4u);
        r.dzr[j] = ld4_so(dzb, o);
        r.actr[j] = ld4_so(acb, o);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) r.im[j] = ld1u_so(fp, imoff[j] - ((im_s1 >> j) & 1u) * 28224u);
      rare_path();
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NDZ_J; ++j) {
      if (j + 1 < NDZ_J || ((dz_ok >> j) & 1u)) {
        f4 g = (f4){leaky_g(r.actr[j].x, r.dzr[j].x), leaky_g(r.actr[j].y, r.dzr[j].y), leaky_g(r.actr[j].z, r.dzr[j].z),
                    leaky_g(r.actr[j].w, r.dzr[j].w)};
        if (!r.full) {
          rare_path();
          if ((dz_s1 >> j) & 1u) g = zero4();
        }
        f2* d = (f2*)(buf + ldsoff[j]);  // 8-byte aligned: two ds_write_b64 with immediate offsets
        d[0] = (f2){g.x, g.y};
        d[1] = (f2){g.z, g.w};
        bacc[j] += (g.x + g.y) + (g.z + g.w);  // the bias gradient rides along
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int idx = tid + 256 * j;
      if (idx < 1344) {
        const unsigned v = r.im[j];
        st4(buf + B_OFF + idx * 4, (f4){u8_unit(v & 255u), u8_unit((v >> 8) & 255u), u8_unit((v >> 16) & 255u), u8_unit(v >> 24)});
      }
    }
  }
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[NE][2], float* lds) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      float* slab = p.part + ((int64_t)split * 2 + i) * SLAB;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wc * 64 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[acc_row(r, hi) * 256 + col] = acc[i][j][r];
      }
    }
    // bias partial of (e, oc) = its 2 samples x 5 quad sums
    const int te = threadIdx.x % TPE;
#pragma unroll
    for (int j = 0; j < NDZ_J; ++j)
      if ((dz_ok >> j) & 1u) lds[ew * QPE + te + TPE * j] = bacc[j];
    __syncthreads();
    if (threadIdx.x < ROWS) {
      const int e = threadIdx.x >> 5, oc = threadIdx.x & 31;
      float s = 0.0f;
#pragma unroll
      for (int smp = 0; smp < 2; ++smp)
#pragma unroll
        for (int q = 0; q < 5; ++q) s += lds[e * QPE + (smp * 32 + oc) * 5 + q];
      p.part[((int64_t)split * 2 + e) * SLAB + 8192 + oc] = s;
    }
  }
};

// ================================================================================================
// conv2: rows = oc (64), cols = 256 taps = 16 input channels x 4 x 4 (blockIdx.x = channel group),
// k-block = (sample pair, band of 3 output rows): 27 k-steps.
// ================================================================================================
struct ConvWgrad2v2 {
  static constexpr int COMMIT_FIRST = 1;  // 5.73 -> 5.64 ms
  static constexpr int THREADS = 256, TM = 2, TN = 2, KSTEPS = 27;
  static constexpr int A_FLOATS = 2 * 64 * 27, B_OFF = A_FLOATS, B_FLOATS = 2 * 16 * 160, STAGE = A_FLOATS + B_FLOATS;
  // dz staging: the k-block's 128 rows (sample, oc) x 27 floats are copied element by element (the rows are
  // only 4-byte aligned in global memory; 16-byte loads at that alignment measured slower than dword loads)
  static constexpr int NDZ = 128 * 27, NDZ_J = (NDZ + 255) / 256;
  static constexpr int64_t SLAB = 64 * 512 + 64;
  struct Params {
    const float* in;  // a1
    int64_t in_es;
    const float* dz;  // dz2
    int64_t dz_es;
    float* part;
    int n, nsplit;
  };
  struct Regs {
    float dzr[NDZ_J];
    f4 im[5];
    bool full;  // wave-uniform: both samples of the pair exist
  };
  int abase[2], bbase[2], kb_begin, kb_end;
  int e, g, split, l31, hi, wc;
  // k-block independent lane parts (bytes) of this thread's source addresses, see ld4_so()
  uint32_t dzoff[NDZ_J], imoff[5];
  unsigned dz_s1, dz_ok, im_s1;  // bit j: slot j belongs to the pair's second sample / exists
  float bacc[NDZ_J];             // bias gradient: running sums of the dz elements this thread stages
  const float* in;
  const float* dz;
  static constexpr int aoff(int s) { return s; }
  static constexpr int boff(int s) { return (s / 9) * 40 + (s % 9) * 2; }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    const int lane = tid & 63;
    wc = tid >> 6;
    l31 = lane & 31;
    hi = lane >> 5;
    g = blockIdx.x;
    split = blockIdx.y;
    e = blockIdx.z;
    WgradSplit sp;
    sp.set(p.n, p.nsplit, split);
    kb_begin = sp.pair_begin * 3;
    kb_end = sp.pair_end * 3;
    in = p.in + e * p.in_es + g * 16 * 400;
    dz = p.dz + e * p.dz_es;
    dz_s1 = dz_ok = im_s1 = 0;
#pragma unroll
    for (int j = 0; j < NDZ_J; ++j) {
      const int idx = tid + 256 * j, c = min(idx, NDZ - 1);
      const int row = c / 27;
      dzoff[j] = (uint32_t)(((row >> 6) * 5184 + (row & 63) * 81 + c % 27) * 4);
      dz_s1 |= (unsigned)(row >> 6) << j;
      dz_ok |= (idx < NDZ ? 1u : 0u) << j;
      bacc[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int idx = tid + 256 * j;
      const int rr = idx % 640;
      imoff[j] = (uint32_t)(((idx / 640) * 12800 + (rr / 40) * 400 + (rr % 40) * 4) * 4);
      im_s1 |= (unsigned)(idx / 640) << j;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = hi * 1728 + (i * 32 + l31) * 27;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wc * 64 + j * 32 + l31;
      bbase[j] = B_OFF + hi * 2560 + (col >> 4) * 160 + ((col >> 2) & 3) * 20 + (col & 3);
    }
  }
  __device__ __forceinline__ void fetch(const Params& p, int kb, Regs& r) {
    const int pair = kb / 3, band = kb % 3;
    r.full = 2 * pair + 1 < p.n;
    pin_offsets(dzoff);
    pin_offsets(imoff);
    // Unconditional loads (a guarded load makes hipcc branch and wait per load).
    const float* dzp = dz + (int64_t)pair * (2 * 5184) + band * 27;
    const float* inp = in + (int64_t)pair * (2 * 12800) + band * 120;
    if (r.full) {
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j) r.dzr[j] = ld1f_so(dzp, dzoff[j]);
#pragma unroll
      for (int j = 0; j < 5; ++j) r.im[j] = ld4_so(inp, imoff[j]);
    } else {
      // last, half-filled pair of an odd batch: the second sample does not exist -> its slots read
      // the first sample instead and commit() zeroes their dz
      rare_path();
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j) r.dzr[j] = ld1f_so(dzp, dzoff[j] - ((dz_s1 >> j) & 1u) * (5184u * 4u));
#pragma unroll
      for (int j = 0; j < 5; ++j) r.im[j] = ld4_so(inp, imoff[j] - ((im_s1 >> j) & 1u) * (12800u * 4u));
      rare_path();
    }
  }
  __device__ __forceinline__ void commit(const Regs& r, float* buf) {
    const int tid = threadIdx.x;
    if (r.full) {
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j)
        if (j + 1 < NDZ_J || ((dz_ok >> j) & 1u)) {
          buf[tid + 256 * j] = r.dzr[j];
          if (g == 0) bacc[j] += r.dzr[j];  // the bias gradient rides along
        }
    } else {
      rare_path();
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j)
        if (j + 1 < NDZ_J || ((dz_ok >> j) & 1u)) {
          const float v = ((dz_s1 >> j) & 1u) ? 0.0f : r.dzr[j];
          buf[tid + 256 * j] = v;
          if (g == 0) bacc[j] += v;
        }
      rare_path();
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) st4(buf + B_OFF + (tid + 256 * j) * 4, r.im[j]);
  }
  __device__ __forceinline__ void extra(const float*) {}
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[2][2], float* lds) {
    float* slab = p.part + ((int64_t)split * 2 + e) * SLAB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = g * 256 + wc * 64 + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(i * 32 + acc_row(r, hi)) * 512 + col] = acc[i][j][r];
    }
    if (g == 0) {  // bias partial of oc = the sums of its 2 x 27 staged elements
#pragma unroll
      for (int j = 0; j < NDZ_J; ++j)
        if ((dz_ok >> j) & 1u) lds[threadIdx.x + 256 * j] = bacc[j];
      __syncthreads();
      if (threadIdx.x < 64) {
        float s = 0.0f;
#pragma unroll
        for (int smp = 0; smp < 2; ++smp)
#pragma unroll
          for (int q = 0; q < 27; ++q) s += lds[(smp * 64 + threadIdx.x) * 27 + q];
        slab[32768 + threadIdx.x] = s;
      }
    }
  }
};

// ================================================================================================
// conv3: rows = oc (64, split over 2 wave rows), cols = 192 taps per workgroup (blockIdx.x = 0..2
// over the 576 = 64 ch x 3 x 3 taps; 24 staged channels cover any 192-tap window),
// k-block = one sample pair: 49 k-steps.
// ================================================================================================
struct ConvWgrad3v2 {
  static constexpr int COMMIT_FIRST = 1;  // 4.25 -> 4.19 ms
  static constexpr int THREADS = 256, TM = 1, TN = 3, KSTEPS = 49;
  static constexpr int A_FLOATS = 2 * 64 * 49, B_OFF = A_FLOATS, B_FLOATS = 2 * 24 * 81, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int64_t SLAB = 64 * 576 + 64;
  struct Params {
    const float* in;  // a2
    int64_t in_es;
    const float* dz;  // dz3
    int64_t dz_es;
    float* part;
    int n, nsplit;
  };
  // both tiles are plain copies of global memory: staged LDS-direct (no staging registers, no commit)
  static constexpr int DIRECT_PENDING = 0;
  struct Regs {};
  int abase[1], bbase[3], kb_begin, kb_end;
  int e, g, split, l31, hi, wr, wc;
  // k-block independent lane parts (bytes) of this thread's source addresses, see ld4_so()
  uint32_t dzoff[7], imoff[4];
  unsigned dz_s1, im_s1;  // bit j: slot j belongs to the pair's second sample
  const float* in;
  const float* dz;
  float bacc;
  static constexpr int aoff(int s) { return s; }
  static constexpr int boff(int s) { return (s / 7) * 9 + (s % 7); }
  __device__ __forceinline__ void init(const Params& p, int tid, float*) {
    const int lane = tid & 63, wave = tid >> 6;
    wr = wave >> 1;
    wc = wave & 1;
    l31 = lane & 31;
    hi = lane >> 5;
    g = blockIdx.x;
    split = blockIdx.y;
    e = blockIdx.z;
    WgradSplit sp;
    sp.set(p.n, p.nsplit, split);
    kb_begin = sp.pair_begin;
    kb_end = sp.pair_end;
    const int ch0 = g * 20;
    in = p.in + e * p.in_es + ch0 * 81;
    dz = p.dz + e * p.dz_es;
    bacc = 0.0f;
    dz_s1 = im_s1 = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int idx = min(tid + 256 * j, 1567);
      dzoff[j] = (uint32_t)(idx * 16);
      dz_s1 |= (unsigned)(idx / 784) << j;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = min(tid + 256 * j, 971);
      imoff[j] = (uint32_t)(((idx / 486) * 5184 + (idx % 486) * 4) * 4);
      im_s1 |= (unsigned)(idx / 486) << j;
    }
    abase[0] = hi * 3136 + (wr * 32 + l31) * 49;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = g * 192 + wc * 96 + j * 32 + l31;
      const int ch = col / 9, t = col % 9;
      bbase[j] = B_OFF + hi * 1944 + (ch - ch0) * 81 + (t / 3) * 9 + (t % 3);
    }
  }
  __device__ __forceinline__ void direct(const Params& p, int kb, float* stage) {
    pin_offsets(dzoff);
    pin_offsets(imoff);
    const float* dzp = dz + (int64_t)kb * (2 * FLAT);  // the pair's dz3 is one contiguous run
    const float* inp = in + (int64_t)kb * (2 * 5184);
    const int wave = wave_u(), tid = threadIdx.x;
    if (2 * kb + 1 < p.n) {
      direct_copy(dzp, dzoff, stage, wave, tid, 1568);
      direct_copy(inp, imoff, stage + B_OFF, wave, tid, 972);
    } else {
      // last, half-filled pair of an odd batch: the second sample does not exist -> its quads read the
      // first sample instead and direct_done() zeroes them in the dz tile
      rare_path();
      uint32_t dzo[7], imo[4];
#pragma unroll
      for (int q = 0; q < 7; ++q) dzo[q] = dzoff[q] - ((dz_s1 >> q) & 1u) * (uint32_t)(FLAT * 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) imo[q] = imoff[q] - ((im_s1 >> q) & 1u) * (5184u * 4u);
      direct_copy(dzp, dzo, stage, wave, tid, 1568);
      direct_copy(inp, imo, stage + B_OFF, wave, tid, 972);
      rare_path();
    }
  }
  __device__ __forceinline__ void direct_done(const Params& p, int kb, float* stage) {
    if (2 * kb + 1 >= p.n) {  // each thread clears the second-sample quads it loaded itself
      rare_path();
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int idx = threadIdx.x + 256 * q;
        if (idx < 1568 && ((dz_s1 >> q) & 1u)) st4(stage + idx * 4, zero4());
      }
      rare_path();
    }
  }
  __device__ __forceinline__ void fetch(const Params&, int, Regs&) {}
  __device__ __forceinline__ void commit(const Regs&, float*) {}
  __device__ __forceinline__ void extra(const float* cur) {
    // bias gradient: every thread of the column-tile-0 workgroups adds up half a dz row (25 / 24 values) per
    // k-block, so that the LDS reads are spread over all four waves
    if (g == 0) {
      const float* row = cur + (threadIdx.x >> 1) * 49 + (threadIdx.x & 1) * 25;
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < 24; ++q) s += row[q];
      if (!(threadIdx.x & 1)) s += row[24];
      bacc += s;
    }
  }
  __device__ __forceinline__ void epilogue(const Params& p, f32x16 (&acc)[1][3], float* lds) {
    float* slab = p.part + ((int64_t)split * 2 + e) * SLAB;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = g * 192 + wc * 96 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(wr * 32 + acc_row(r, hi)) * 576 + col] = acc[0][j][r];
    }
    if (g == 0) {
      lds[threadIdx.x] = bacc;  // [sample of the pair][oc][half row]
      __syncthreads();
      if (threadIdx.x < 64) {
        const float* q = lds + 2 * threadIdx.x;
        slab[36864 + threadIdx.x] = (q[0] + q[1]) + (q[128] + q[129]);
      }
    }
  }
};

// ================================================================================================
void launch_conv_wgrad3_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
  const int S = c.splits->c3;
  {
    ConvWgrad3v2::Params p{w.a2, MB * 5184, w.dz3, MB * FLAT, w.wpart, c.n, S};
    ProfRange pr(c.prof, "ConvWgrad3", st);
    launch_engine2<ConvWgrad3v2>(dim3(3, S, L.NE), p, st);
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, ConvWgrad3v2::SLAB, L.NE, grads, L.enc_base[0] + L.enc.c3w, L.enc_base[1] + L.enc.c3w, st);
}

void launch_conv_wgrad2_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
  const int S = c.splits->c2;
  {
    ConvWgrad2v2::Params p{w.a1, MB * 12800, w.dz2, MB * 5184, w.wpart, c.n, S};
    ProfRange pr(c.prof, "ConvWgrad2", st);
    launch_engine2<ConvWgrad2v2>(dim3(2, S, L.NE), p, st);
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, ConvWgrad2v2::SLAB, L.NE, grads, L.enc_base[0] + L.enc.c2w, L.enc_base[1] + L.enc.c2w, st);
}

void launch_conv_wgrad1_2(const EncCall& c, float* grads, hipStream_t st) {
  const Workspace& w = *c.ws;
  const int64_t MB = c.max_batch;
  const ParamLayout& L = *c.L;
  const int S = c.splits->c1;
  {
    ProfRange pr(c.prof, "ConvWgrad1", st);
    if (L.NE == 2) {
      ConvWgrad1v2<2>::Params p{c.frames, w.dz1, w.a1, MB * 12800, w.wpart, c.n, S};
      launch_engine2<ConvWgrad1v2<2>>(dim3(1, S, 1), p, st);
    } else {
      ConvWgrad1v2<1>::Params p{c.frames, w.dz1, w.a1, MB * 12800, w.wpart, c.n, S};
      launch_engine2<ConvWgrad1v2<1>>(dim3(1, S, 1), p, st);
    }
  }
  ProfRange pr(c.prof, "reduce_partials", st);
  launch_reduce_partials(w.wpart, S, ConvWgrad1v2<2>::SLAB, L.NE, grads, L.enc_base[0] + L.enc.c1w, L.enc_base[1] + L.enc.c1w, st);
}

}  // namespace ddrl
