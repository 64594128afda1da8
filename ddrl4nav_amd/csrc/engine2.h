// Pipelined f32-MFMA tile engine shared by every GEMM-shaped kernel of the library.
//
//   C[row][col] = sum_k A(row,k) * B(k,col),  rows -> MFMA A operand -> accumulator registers,
//   cols -> MFMA B operand -> lanes (stores are coalesced along cols).
//
// Design points (the first build of this library gathered im2col tiles element by element and
// ran at a third of this engine's rate, profiles/README.md):
//   * operands are read from LDS as  lds[lane_base + compile-time immediate]  (one ds_read_b32
//     with an offset field per operand, zero address VALU inside the k loop);
//   * LDS is double buffered and the next k-block is prefetched global -> registers while the
//     current one feeds the matrix pipe (one barrier per k-block);
//   * ops stage RAW tensors (input planes, weight slabs) with wide coalesced loads instead of
//     gathering an im2col tile element by element (the generic gather ops of gconv.hip are the
//     exception, for layers without a specialised kernel);
//   * the order of LDS reads and MFMAs inside a k-block is pinned (compute_block), and ops choose
//     by trait whether commit/fetch run before or after the MFMA block (COMMIT_FIRST, IGLP, OCC,
//     EXTRA, PRE_EPILOGUE below).
//
// An Op provides:
//   constants  THREADS, TM, TN, KSTEPS, STAGE (floats per LDS buffer)
//   struct Params, struct Regs (prefetch registers)
//   bool  init(P, tid, lds)            tile coordinates, lane bases abase[TM], bbase[TN], kb range
//   void  fetch(P, kb, regs)           issue global loads of k-block kb
//   void  commit(regs, buf)            registers -> LDS buffer
//   void  extra(cur)                   optional side work on the published buffer (bias sums)
//   static constexpr int aoff(s), boff(s)   immediates of k-step s (in floats)
//   void  epilogue(P, acc)
#pragma once
#include "kernels.h"

namespace ddrl {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// 16-byte register vector for staged data.  NOT HIP's float4: that is a struct, a whole-struct copy
// lowers to llvm.memcpy, and an alloca touched only by memcpys is never promoted to registers
// (the prefetch buffers then live in scratch and every load is waited for immediately).
using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f4 ld4(const float* p) { return *(const f4*)p; }
__device__ __forceinline__ void st4(float* p, f4 v) { *(f4*)p = v; }
__device__ __forceinline__ f4 zero4() { return (f4){0.0f, 0.0f, 0.0f, 0.0f}; }
// Load through the  scalar base + 32-bit lane offset  addressing form (global_load v, v_off, s[base]): the
// k-block dependent part of the address is a wave-uniform pointer, the lane part a loop-invariant byte offset
// kept in ONE register, so no address arithmetic runs on the vector ALU inside the k loop.  That matters
// because VALU instructions do not overlap f32 MFMAs on a SIMD (tools/mfma_peak.hip: ~3.5 cycles of
// matrix-pipe time lost per VALU instruction).  The empty asm keeps the zero-extension next to the load;
// hoisted out of the loop it would turn back into a 64-bit vector add per load.  pin_offsets() "redefines" the
// loop-carried offset registers in place (no copy) and must run once per fetch, BEFORE any branch that
// selects between load paths (a redefinition inside one arm costs a register copy per offset at the join).
// LDS-direct staging: one global_load_lds_dwordx4 moves 16 bytes per lane from global memory straight into
// LDS (no staging registers, no ds_write): the 64 lanes of a wave fill the 1 KB that starts at `lds_wave_base`
// (wave-uniform) in lane order.  Completion is tracked by vmcnt; the engine waits for it before the barrier
// that publishes the stage (engine2_step).
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
__device__ __forceinline__ void ld16_to_lds(const void* uniform_base, uint32_t lane_bytes, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gvoid_t*)((const char*)uniform_base + lane_bytes), (lvoid_t*)lds_wave_base, 16, 0, 0);
}
// Contiguous tile copy, LDS-direct: quad (tid + 256 j) of the tile comes from uniform_base + off[j]; the 64
// quads a wave moves per j land 1 KB contiguous at tile + (64 wave + 256 j) quads.  Quads >= nquads are
// skipped (the lanes are masked off).
// wave index inside the workgroup as a scalar (the LDS base of an LDS-direct load must be wave-uniform)
__device__ __forceinline__ int wave_u() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
template <int NJ>
__device__ __forceinline__ void direct_copy(const void* uniform_base, const uint32_t (&off)[NJ], float* tile, int wave, int tid, int nquads) {
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (256 * (j + 1) <= nquads || tid + 256 * j < nquads) ld16_to_lds(uniform_base, off[j], tile + (64 * wave + 256 * j) * 4);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// First and last statement of a rarely taken arm (ragged tails): an un-speculatable marker, so that the
// compiler keeps the arm behind its wave-uniform branch instead of hoisting / sinking its instructions
// into the common path (which costs per-lane selects or register copies there).
__device__ __forceinline__ void rare_path() { asm volatile("; rare path"); }
template <int N>
__device__ __forceinline__ void pin_offsets(uint32_t (&off)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(off[j]));
}
__device__ __forceinline__ f4 ld4_so(const void* uniform_base, uint32_t lane_bytes) {
  return *(const f4*)((const char*)uniform_base + lane_bytes);
}
#ifndef DDRL_ST_SBASE
#define DDRL_ST_SBASE 0  // measured: 61 fewer vector instructions per ConvFwd1 workgroup, ConvFwd2 2.36 -> 2.51 ms, iteration 25.18 -> 25.40: off
#endif
__device__ __forceinline__ void st1_so(void* uniform_base, uint32_t lane_bytes, float v) {
#if DDRL_ST_SBASE
  // the row offset that callers fold into `uniform_base` (k x 1,600 B, k x 324 B ...) fits the store's 12-bit immediate only for the
  // first few rows of an epilogue; beyond that the compiler rebuilds a 64-bit VECTOR address per store (v_add_co + v_addc, or
  // v_lshl_add_u64 when the base is merely opaque).  Written out, the sum stays on the scalar ALU:
  //   global_store_dword v_off, v_data, s[base:base+1]
  // (a dword store reads its data register at issue: no hazard the compiler would have to know about)
  asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_bytes), "v"(v), "s"(uniform_base) : "memory");
#else
  *(float*)((char*)uniform_base + lane_bytes) = v;
#endif
}
__device__ __forceinline__ float ld1f_so(const void* uniform_base, uint32_t lane_bytes) {
  return *(const float*)((const char*)uniform_base + lane_bytes);
}
__device__ __forceinline__ unsigned ld1u_so(const void* uniform_base, uint32_t lane_bytes) {
  return *(const unsigned*)((const char*)uniform_base + lane_bytes);
}

// leaky_relu(v) = max(v, LEAKY * v) for a slope below one: one multiply + one max (bit-identical to the
// compare/select form, signed zeros included)
__device__ __forceinline__ float leaky_f(float v) { return __builtin_fmaxf(v, v * LEAKY); }
__device__ __forceinline__ float leaky_g(float act, float g) { return act > 0.0f ? g : g * LEAKY; }
// the same with the decision taken from bit `bit` of a sign mask (bit SET = act is not positive): g * LEAKY or g * 1 without a
// compare (bfe, bfi, mul)
__device__ __forceinline__ float leaky_bit(unsigned mask, int bit, float g) {
  const unsigned sel = (unsigned)__builtin_amdgcn_sbfe((int)mask, bit, 1);  // 0 or ~0
  return g * __uint_as_float((__float_as_uint(LEAKY) & sel) | (0x3F800000u & ~sel));
}

// float32(u8/255.0), correctly rounded, in three vector instructions (convert, multiply, fma): 1/255 is
// split into float32 hi + lo parts and x*hi + fl(x*lo) is rounded once.  Exhaustively equal to the
// correctly rounded quotient for x = 0..255 (tests/test_gpu_parity.py checks ddrl_u8_table, which runs this
// function, bit for bit against numpy).
__device__ __forceinline__ float u8_unit(unsigned b) {
  const float x = (float)b;
  const float r_hi = 0x1.010102p-8f, r_lo = -0x1.fdfdfep-33f;
  return __builtin_fmaf(x, r_hi, x * r_lo);
}

// Ops may declare `static constexpr int IGLP = 1`: use __builtin_amdgcn_iglp_opt(0) instead of the
// pinned read-ahead order (measured per op: better for the three dense-layer kernels only).
template <class Op, class = void>
struct IglpOf {
  static constexpr bool v = false;
};
template <class Op>
struct IglpOf<Op, decltype((void)Op::IGLP)> {
#ifdef DDRL_NO_IGLP  // A/B switch for tools/ablate_iter.py
  static constexpr bool v = false;
#else
  static constexpr bool v = Op::IGLP != 0;
#endif
};

template <class Op>
__device__ __forceinline__ void compute_block(const Op& op, const float* __restrict__ cur, const float* lds_base,
                                              f32x16 (&acc)[Op::TM][Op::TN]) {
#ifdef DDRL_SETPRIO
  __builtin_amdgcn_s_setprio(DDRL_SETPRIO);
#endif
#ifndef DDRL_NO_LDS_PREFETCH
  // Operand pointers of this k-block.  When the two stage buffers together exceed the 64 KB reach of the
  // DS immediate offset, the (compile-time) offset of the second buffer is folded into the lane registers
  // here, once per k-block, and hidden from the compiler; otherwise it would re-base the address with a
  // vector add in front of every far read.
  const float* pa[Op::TM];
  const float* pb[Op::TN];
  if constexpr (2 * Op::STAGE * sizeof(float) > 65536) {
    // (the pinned quantity is the byte offset from the start of LDS, not a pointer: an opaque pointer
    // would lose its LDS address space and turn the reads into flat loads)
    const unsigned cur_bytes = (unsigned)((const char*)cur - (const char*)lds_base);
#pragma unroll
    for (int i = 0; i < Op::TM; ++i) {
      unsigned o = cur_bytes + 4u * (unsigned)op.abase[i];
      asm volatile("" : "+v"(o));
      pa[i] = (const float*)((const char*)lds_base + o);
    }
#pragma unroll
    for (int j = 0; j < Op::TN; ++j) {
      unsigned o = cur_bytes + 4u * (unsigned)op.bbase[j];
      asm volatile("" : "+v"(o));
      pb[j] = (const float*)((const char*)lds_base + o);
    }
  } else {
#pragma unroll
    for (int i = 0; i < Op::TM; ++i) pa[i] = cur + op.abase[i];
#pragma unroll
    for (int j = 0; j < Op::TN; ++j) pb[j] = cur + op.bbase[j];
  }
  // operands of k-step s+1 are read from LDS before the MFMAs of k-step s are issued (measured:
  // -0.75 ms per PPO iteration over the un-pinned schedule, profiles/README.md)
  float a[Op::TM], b[Op::TN];
#pragma unroll
  for (int i = 0; i < Op::TM; ++i) a[i] = pa[i][Op::aoff(0)];
#pragma unroll
  for (int j = 0; j < Op::TN; ++j) b[j] = pb[j][Op::boff(0)];
#pragma unroll
  for (int s = 0; s < Op::KSTEPS; ++s) {
    float an[Op::TM], bn[Op::TN];
    if (s + 1 < Op::KSTEPS) {
#pragma unroll
      for (int i = 0; i < Op::TM; ++i) an[i] = pa[i][Op::aoff(s + 1 < Op::KSTEPS ? s + 1 : s)];
#pragma unroll
      for (int j = 0; j < Op::TN; ++j) bn[j] = pb[j][Op::boff(s + 1 < Op::KSTEPS ? s + 1 : s)];
    }
#pragma unroll
    for (int i = 0; i < Op::TM; ++i)
#pragma unroll
      for (int j = 0; j < Op::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    if (s + 1 < Op::KSTEPS) {
#pragma unroll
      for (int i = 0; i < Op::TM; ++i) a[i] = an[i];
#pragma unroll
      for (int j = 0; j < Op::TN; ++j) b[j] = bn[j];
    }
    // pin the order the machine scheduler would otherwise undo: the LDS reads of the NEXT k-step, then
    // this k-step's MFMAs (0x100 = DS read, 0x008 = MFMA)
    if constexpr (IglpOf<Op>::v) {
      if (s == 0) __builtin_amdgcn_iglp_opt(0);  // LLVM's built-in MFMA / DS interleave for small GEMMs
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, Op::TM + Op::TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, Op::TM * Op::TN, 0);
    }
  }
#else
#pragma unroll
  for (int s = 0; s < Op::KSTEPS; ++s) {
    float a[Op::TM], b[Op::TN];
#pragma unroll
    for (int i = 0; i < Op::TM; ++i) a[i] = cur[op.abase[i] + Op::aoff(s)];
#pragma unroll
    for (int j = 0; j < Op::TN; ++j) b[j] = cur[op.bbase[j] + Op::boff(s)];
#pragma unroll
    for (int i = 0; i < Op::TM; ++i)
#pragma unroll
      for (int j = 0; j < Op::TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
#endif
#ifdef DDRL_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Ops may declare `static constexpr int OCC` = waves per SIMD the register allocator must leave
// room for (second __launch_bounds__ argument); default 1.
template <class Op, class = void>
struct OccOf {
  static constexpr int v = 1;
};
template <class Op>
struct OccOf<Op, decltype((void)Op::OCC)> {
  static constexpr int v = Op::OCC;
};

// Ops may declare `static constexpr int COMMIT_FIRST = 1`: commit + fetch run before the MFMA block
// of each iteration instead of after it (measured per op: helps the conv weight gradients, whose
// commits carry the ReLU masks / u8 conversions; hurts the forward and data-gradient kernels).
template <class Op, class = void>
struct CommitFirstOf {
  static constexpr bool v = false;
};
template <class Op>
struct CommitFirstOf<Op, decltype((void)Op::COMMIT_FIRST)> {
#ifdef DDRL_NO_COMMIT_FIRST  // A/B switch for tools/ablate_iter.py
  static constexpr bool v = false;
#else
  static constexpr bool v = Op::COMMIT_FIRST != 0;
#endif
};

// Ops may stage (part of) a k-block with LDS-direct loads: `static constexpr int DIRECT_PENDING` = number of
// register-prefetch global loads fetch() issues (they are issued AFTER the direct loads of an iteration
// and may stay in flight across the barrier), and direct(P, kb, stage) issues the direct loads of k-block kb.
template <class Op, class = void>
struct HasDirect {
  static constexpr bool v = false;
};
template <class Op>
struct HasDirect<Op, decltype((void)Op::DIRECT_PENDING)> {
  static constexpr bool v = true;
};
template <class Op>
__device__ __forceinline__ void wait_direct(bool fetch_in_flight) {
  if constexpr (HasDirect<Op>::v) {
    if constexpr (Op::DIRECT_PENDING == 0) {
      wait_vmcnt<0>();
    } else {
      if (fetch_in_flight) wait_vmcnt<Op::DIRECT_PENDING>();
      else wait_vmcnt<0>();
    }
  }
}

// Ops may define pre_epilogue(P): issued before the last k-block (see engine2_kernel).
template <class Op, class = void>
struct HasPreEpilogue {
  static constexpr bool v = false;
};
template <class Op>
struct HasPreEpilogue<Op, decltype((void)Op::PRE_EPILOGUE)> {
  static constexpr bool v = true;
};

// One k-block of the pipeline with the LDS buffer index as a compile-time constant: every LDS address of
// the block is then  lane register + immediate  (no per-iteration buffer arithmetic on the vector ALU).
// BUF = -1: the buffer index is the run-time argument `rbuf` (the tail of the k loop, which also hosts the
// pre-epilogue hook, is compiled once this way).
template <class Op, int BUF>
__device__ __forceinline__ void engine2_step(Op& op, const typename Op::Params& P, int kb, int kbe, typename Op::Regs& regs,
                                             f32x16 (&acc)[Op::TM][Op::TN], float* lds2, int rbuf = 0) {
  float* cur = lds2 + (BUF < 0 ? rbuf : BUF) * Op::STAGE;
  float* nxt = lds2 + (BUF < 0 ? rbuf ^ 1 : BUF ^ 1) * Op::STAGE;
  if constexpr (HasDirect<Op>::v) {
    // LDS-direct part of the next stage: in flight under this block's MFMAs (the other buffer was released by
    // the barrier that ended the previous block)
    if (kb + 1 < kbe) op.direct(P, kb + 1, nxt);
  }
  op.extra(cur);
  if constexpr (HasPreEpilogue<Op>::v && BUF < 0) {
    // global loads the epilogue needs (e.g. the activations for the leaky-ReLU mask) are issued
    // before the last k-block, so their latency hides under its MFMAs instead of being exposed
    if (kb == kbe - 1) op.pre_epilogue(P);
  }
  // DDRL_ABL_* are timing-only knock-outs for tools/ablate_engine.sh (results are WRONG with any of
  // them set): they show what each phase of the loop costs on top of the bare LDS->MFMA stream.
  if constexpr (CommitFirstOf<Op>::v) {
    // ops with a VALU-heavy commit (masks, u8 conversion): write the NEXT stage and issue the
    // following fetch before this block's MFMAs, so the scheduler can run them under the MFMAs
    if (kb + 1 < kbe) {
#ifndef DDRL_ABL_NOCOMMIT
      op.commit(regs, nxt);
#endif
#ifndef DDRL_ABL_NOFETCH
      if (kb + 2 < kbe) op.fetch(P, kb + 2, regs);
#endif
    }
    compute_block<Op>(op, cur, lds2, acc);
  } else {
    compute_block<Op>(op, cur, lds2, acc);
    if (kb + 1 < kbe) {
#ifndef DDRL_ABL_NOCOMMIT
      op.commit(regs, nxt);
#endif
#ifndef DDRL_ABL_NOFETCH
      if (kb + 2 < kbe) op.fetch(P, kb + 2, regs);
#endif
    }
  }
  wait_direct<Op>(kb + 2 < kbe);
  if constexpr (HasDirect<Op>::v) {
    if (kb + 1 < kbe) op.direct_done(P, kb + 1, nxt);  // e.g. zero-fill of a ragged last k-block
  }
#ifndef DDRL_ABL_NOBARRIER
  __syncthreads();
#endif
}

template <class Op>
__global__ __launch_bounds__(Op::THREADS, OccOf<Op>::v) void engine2_kernel(typename Op::Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  Op op;
  const int tid = threadIdx.x;
#ifdef DDRL_EDGE_PRIO
  __builtin_amdgcn_s_setprio(DDRL_EDGE_PRIO);
#endif
  op.init(P, tid, lds2);
  typename Op::Regs regs;
  f32x16 acc[Op::TM][Op::TN];
#pragma unroll
  for (int i = 0; i < Op::TM; ++i)
#pragma unroll
    for (int j = 0; j < Op::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  int kb = op.kb_begin;
  const int kbe = op.kb_end;
  if (kb < kbe) {
    if constexpr (HasDirect<Op>::v) op.direct(P, kb, lds2);
    op.fetch(P, kb, regs);
    op.commit(regs, lds2);
    if (kb + 1 < kbe) op.fetch(P, kb + 1, regs);
    wait_direct<Op>(kb + 1 < kbe);
    if constexpr (HasDirect<Op>::v) op.direct_done(P, kb, lds2);
  }
  __syncthreads();
#ifdef DDRL_EDGE_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  // The loop is unrolled by the two LDS buffers (engine2_step) and leaves the last one or two k-blocks to a
  // tail with a run-time buffer index (an exit in the middle of the unrolled loop would cost a copy of every
  // accumulator register per iteration; the tail also keeps the pre-epilogue registers out of the loop).
  for (; kb + 2 < kbe; kb += 2) {
    engine2_step<Op, 0>(op, P, kb, kbe, regs, acc, lds2);
    engine2_step<Op, 1>(op, P, kb + 1, kbe, regs, acc, lds2);
  }
  for (int rbuf = 0; kb < kbe; ++kb, rbuf ^= 1) engine2_step<Op, -1>(op, P, kb, kbe, regs, acc, lds2, rbuf);
#ifdef DDRL_EDGE_PRIO
  __builtin_amdgcn_s_setprio(DDRL_EDGE_PRIO);
#endif
#ifndef DDRL_ABL_NOEPILOGUE
  op.epilogue(P, acc, lds2);
#else
  float sink = 0.0f;
#pragma unroll
  for (int i = 0; i < Op::TM; ++i)
#pragma unroll
    for (int j = 0; j < Op::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
  if (sink == 123.456f) lds2[0] = sink;  // keeps the accumulators live without an epilogue
#endif
}

// Ops may declare `static constexpr int EXTRA` = floats of LDS behind the two stage buffers that
// live for the whole kernel (e.g. the bias vector for the epilogue); default 0.
template <class Op, class = void>
struct ExtraOf {
  static constexpr int v = 0;
};
template <class Op>
struct ExtraOf<Op, decltype((void)Op::EXTRA)> {
  static constexpr int v = Op::EXTRA;
};

template <class Op>
inline void launch_engine2(dim3 grid, const typename Op::Params& p, hipStream_t st) {
  constexpr size_t bytes = (size_t)(2 * Op::STAGE + ExtraOf<Op>::v) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    if (bytes > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)engine2_kernel<Op>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    configured = true;
  }
  hipLaunchKernelGGL(engine2_kernel<Op>, grid, dim3(Op::THREADS), bytes, st, p);
}

// accumulator element r of tile (i,j) of this lane -> (row, col) inside the wave tile
// ---- fp32 -> three bf16 planes (only the -DDDRL_PLANES_BF16 build uses them; the default planes are fp16, below) ------
// x = p0 + p1 + p2 to 24 bits: p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1), round-to-nearest-even
// (v_cvt_pk_bf16_f32); the two subtractions are exact in fp32.  Two values per dword: x in the low half, y in the high.
using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ unsigned pack_bf16x2(float x, float y) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){x, y}, bf16x2_t));
}
__device__ __forceinline__ void split_bf16x3(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack_bf16x2(x, y);
  const float r1x = x - __uint_as_float(p0 << 16), r1y = y - __uint_as_float(p0 & 0xFFFF0000u);
  p1 = pack_bf16x2(r1x, r1y);
  const float r2x = r1x - __uint_as_float(p1 << 16), r2y = r1y - __uint_as_float(p1 & 0xFFFF0000u);
  p2 = pack_bf16x2(r2x, r2y);
}
// the six plane products of a k-group under three bf16 planes per operand, smallest first: (a plane, b plane)
#define DDRL_BF16X6_PRODUCTS constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0}

// the value the neighbouring lane (lane ^ 1) holds: one DPP move (quad_perm [1, 0, 3, 2]), no LDS traffic
__device__ __forceinline__ float lane_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- plane scheme of the kernels with TWO fp32 operands -----------------------------------------------------------
// Default "f16x3": every operand as TWO scaled fp16 planes, x S = h0 + h1 to 22 bits (h0 = fp16(x S), h1 = fp16(x S - h0),
// round-to-nearest-even; S = f16_scale(largest magnitude of the tensor), a power of two, so scaling and un-scaling are
// exact), and the three products h0 g0, h0 g1, h1 g0 on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  What is left out
// (h1 g1 and the representation error) is <= 3 x 2^-22 |a b| per term with zero mean; measured against float64 it does not
// show next to the rounding of the fp32 accumulation itself (tests/test_gpu_parity.py::*_is_at_least_fp32_accurate).  Half the
// matrix instructions and two thirds of the LDS planes of "bf16x6" (three bf16 planes per operand, six products), which
// -DDDRL_PLANES_BF16 keeps: bf16 has fp32's exponent range and needs no scale.
// Range: with the tensor's maximum at [2^12, 2^13) every element down to 2^-16 of it keeps its 22 bits (fp16 normal range
// 2^-14), below that the absolute error stays <= 2^-25 / S, i.e. <= 2^-38 of the maximum -- far below 2^-24 of any sum that
// the large elements take part in.
using f16x2_t = __attribute__((ext_vector_type(2))) _Float16;
using h8v = __attribute__((ext_vector_type(8))) _Float16;
using b8v = __attribute__((ext_vector_type(8))) __bf16;
#ifndef DDRL_PLANES_BF16
constexpr int NPL = 2, NPROD = 3;
using frag8 = h8v;
#define DDRL_PLANE_PRODUCTS constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0} /* smallest first: h1 g0, h0 g1, h0 g0 */
// FOUR VALU instructions per PAIR: v_fma_mixlo_f16 / v_fma_mixhi_f16 compute fma(x, S, c) in fp32 and write the result as fp16 into
// the low / high half of the destination, with c an fp32 or (op_sel_hi) an fp16 source: h0 = fp16(x S + 0), h1 = fp16(x S - h0)
// straight from the h0 halves.  x S and x S - h0 are exact in fp32, so each value is rounded once, as in the plain form
// (v_pk_mul_f32, v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_fma_f32, v_cvt_pk_f16_f32: six; element-wise C compiled to ten).
// Bit-identical to it on 12.6 M values over 60 binades and three scales; only -0 comes out as +0.  The kernels issue 3 - 8 VALU
// instructions per MFMA, most of them this split.
using f32x2_t = __attribute__((ext_vector_type(2))) float;
// the plain form (six instructions, but visible to the scheduler): same-box A/B of the two forms per kernel -- the dense forward /
// data gradient and the conv2 weight gradient are 1-3 % faster with it, the other six with the four-instruction form
__device__ __forceinline__ void split_planes_c(float x, float y, float scale, unsigned (&p)[NPL]) {
  const f32x2_t xs = {x * scale, y * scale};
  const f16x2_t a = __builtin_convertvector(xs, f16x2_t);
  const f32x2_t r = {__builtin_fmaf(x, scale, -(float)a[0]), __builtin_fmaf(y, scale, -(float)a[1])};
  const f16x2_t b = __builtin_convertvector(r, f16x2_t);
  p[0] = __builtin_bit_cast(unsigned, a);
  p[1] = __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ void split_planes(float x, float y, float scale, unsigned (&p)[NPL]) {
#ifndef DDRL_SPLIT_PLAIN
  unsigned h0, h1;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(x), "v"(scale));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(y), "v"(scale));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(h1) : "v"(x), "v"(scale), "v"(h0));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h1) : "v"(y), "v"(scale), "v"(h0));
  p[0] = h0;
  p[1] = h1;
#else
  const f32x2_t xs = {x * scale, y * scale};
  const f16x2_t a = __builtin_convertvector(xs, f16x2_t);
  const f32x2_t r = {__builtin_fmaf(x, scale, -(float)a[0]), __builtin_fmaf(y, scale, -(float)a[1])};
  const f16x2_t b = __builtin_convertvector(r, f16x2_t);
  p[0] = __builtin_bit_cast(unsigned, a);
  p[1] = __builtin_bit_cast(unsigned, b);
#endif
}
__device__ __forceinline__ f32x16 mfma_planes(frag8 a, frag8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__host__ __device__ inline float plane_scale(float amax) { return f16_scale(amax); }
// frame bytes 0..255 are exact in fp16 (and in bf16): the pixel operand of conv1 is ONE plane, "f16x2" = NPL products.
// -DDDRL_PIX_SUBNORMAL=1 (measured, NOT used) feeds them as fp16 SUBNORMALS: the 16-bit pattern 0x00bb is the fp16 value b x 2^-24, so
// a byte becomes an operand by zero extension -- one v_perm_b32 per pair of pixels instead of two byte->float conversions and a
// pack, with 2^24 (PIXEL_UNIT) folded into the epilogue's scale.  Same-box A/B: 110 M fewer vector-ALU instructions per iteration
// (4 %), ConvWgrad1 2.83 -> 2.78 ms, the PPO iteration 25.43 -> 25.37 ms (0.2 %) -- and the results are NOT the same bits: the matrix
// pipe takes subnormal operands, but the conv1 weight gradient's mean error against float64 grows from 1.0 x to 1.6 x that of an
// fp32 evaluation (products of a subnormal pixel with the small plane of dz1 lose bits inside the pipe), which breaks
// tests/test_gpu_parity.py::test_conv1_weight_gradient_is_at_least_fp32_accurate.  Vector-ALU count is not what bounds these kernels.
#ifndef DDRL_PIX_SUBNORMAL
#define DDRL_PIX_SUBNORMAL 0
#endif
#if DDRL_PIX_SUBNORMAL
constexpr float PIXEL_UNIT = 16777216.0f;
// [byte QA of a][0][byte QB of b][0] = the two pixels as 16-bit operands
template <int QA, int QB>
__device__ __forceinline__ unsigned pixel_pair_sel(unsigned a, unsigned b) {
  return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (unsigned)QA + ((unsigned)QB << 16));
}
__device__ __forceinline__ unsigned short pixel_one(unsigned a) { return (unsigned short)a; }
#else
constexpr float PIXEL_UNIT = 1.0f;
__device__ __forceinline__ unsigned pixel_pair(unsigned a, unsigned b) {
  const f16x2_t v = {(_Float16)(float)a, (_Float16)(float)b};
  return __builtin_bit_cast(unsigned, v);
}
template <int QA, int QB>
__device__ __forceinline__ unsigned pixel_pair_sel(unsigned a, unsigned b) { return pixel_pair((a >> (8 * QA)) & 255u, (b >> (8 * QB)) & 255u); }
__device__ __forceinline__ unsigned short pixel_one(unsigned a) { return __builtin_bit_cast(unsigned short, (_Float16)(float)a); }
#endif
// the four pixels of one dword as two operand pairs
__device__ __forceinline__ uint2 pixel_quad(unsigned v) { return make_uint2(pixel_pair_sel<0, 1>(v, v), pixel_pair_sel<2, 3>(v, v)); }
__host__ __device__ inline void planes_of(float w, float scale, unsigned short (&p)[NPL]) {
  const float ws = w * scale;
  const _Float16 h0 = (_Float16)ws;
  const _Float16 h1 = (_Float16)(ws - (float)h0);
  p[0] = __builtin_bit_cast(unsigned short, h0);
  p[1] = __builtin_bit_cast(unsigned short, h1);
}
#else
constexpr int NPL = 3, NPROD = 6;
using frag8 = b8v;
#define DDRL_PLANE_PRODUCTS constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0}
__device__ __forceinline__ void split_planes(float x, float y, float, unsigned (&p)[NPL]) { split_bf16x3(x, y, p[0], p[1], p[2]); }
__device__ __forceinline__ void split_planes_c(float x, float y, float s, unsigned (&p)[NPL]) { split_planes(x, y, s, p); }
__device__ __forceinline__ f32x16 mfma_planes(frag8 a, frag8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__host__ __device__ inline float plane_scale(float) { return 1.0f; }
// float(byte) has at most 8 significant bits: its upper 16 bits ARE the bf16 value
constexpr float PIXEL_UNIT = 1.0f;
__device__ __forceinline__ unsigned pixel_pair(unsigned a, unsigned b) {
  return (__float_as_uint((float)a) >> 16) | (__float_as_uint((float)b) & 0xFFFF0000u);
}
template <int QA, int QB>
__device__ __forceinline__ unsigned pixel_pair_sel(unsigned a, unsigned b) { return pixel_pair((a >> (8 * QA)) & 255u, (b >> (8 * QB)) & 255u); }
__device__ __forceinline__ unsigned short pixel_one(unsigned a) { return (unsigned short)(__float_as_uint((float)a) >> 16); }
__device__ __forceinline__ uint2 pixel_quad(unsigned v) { return make_uint2(pixel_pair_sel<0, 1>(v, v), pixel_pair_sel<2, 3>(v, v)); }
__host__ __device__ inline void planes_of(float w, float, unsigned short (&p)[NPL]) {
  auto rne = [](float v) { unsigned u = __builtin_bit_cast(unsigned, v); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
  p[0] = rne(w);
  const float r1 = w - __builtin_bit_cast(float, (unsigned)p[0] << 16);
  p[1] = rne(r1);
  p[2] = rne(r1 - __builtin_bit_cast(float, (unsigned)p[1] << 16));
}
#endif
// largest magnitude of the values a thread wrote -> the tensor's running maximum (float bits of non-negative values order like
// unsigned integers; the maximum does not depend on the order of the atomics, so the result is deterministic).
// One atomic per wave on ONE address serialises in the memory system (25,600 of them cost conv1's forward 6 ms): a wave
// first LOOKS at the slot and only sends the atomic when it would raise it -- after the first few waves almost none does.
// A stale look can only be too small (the slot never decreases), i.e. it costs an atomic, never a missed maximum.
__device__ __forceinline__ void amax_update(float m, float* slot) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load((unsigned*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned*)slot, bits);
  }
}

// ---- per-sample plane scales of the operator kernels (pconv.hip, plin.hip) ----------------------------------------------------
// What travels between operators is a sample's LARGEST MAGNITUDE (any upper bound works): a producer's epilogue raises it with
// amax_raise (non-negative floats order like unsigned integers: a maximum is deterministic whatever the order of the atomics), the
// consumer turns it into the power-of-two scale of the sample's fp16 planes.  An all-zero sample (the gradient of a sample whose advantage
// is exactly 0) gets the largest scale there is -- it must not pin the batch scale of a weight gradient --; 2^60 also bounds the
// products of two scales.
__device__ __forceinline__ float scale_of_amax(float m) { return m > 0.0f ? fminf(plane_scale(m), 0x1p60f) : 0x1p60f; }
__device__ __forceinline__ void amax_raise(float m, float* slot) {  // one lane; m >= 0; the slot never decreases
  const unsigned bits = __float_as_uint(m);
  if (bits > __hip_atomic_load((unsigned*)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned*)slot, bits);
}
__device__ __forceinline__ void lds_amax_raise(float m, float* lds_slot) { atomicMax((unsigned*)lds_slot, __float_as_uint(m)); }

}  // namespace ddrl
