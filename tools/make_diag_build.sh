#!/bin/bash
# Diagnostic only: build an instrumented copy of the library (s_memtime stamps around the phases
# of the engine loop) as ddrl4nav_amd/csrc/libddrl_hip_diag.so; tools/diag_stamps.py reads it.
# The stamps' fences forbid overlaps the real kernel has: read SHARES, never the run time.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/ddrl_diag
rm -rf $D && mkdir -p $D/ddrl4nav_amd/csrc $D/include
cp $ROOT/ddrl4nav_amd/csrc/*.hip $ROOT/ddrl4nav_amd/csrc/*.h $ROOT/ddrl4nav_amd/csrc/*.cpp $ROOT/ddrl4nav_amd/csrc/Makefile $D/ddrl4nav_amd/csrc/
cp $ROOT/include/ddrl.h $D/include/
cd $D/ddrl4nav_amd/csrc
python3 - <<'PY'
import re
s=open('engine2.h').read()
s=s.replace('''template <class Op>
__global__ __launch_bounds__(Op::THREADS, OccOf<Op>::v) void engine2_kernel(typename Op::Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  Op op;
  const int tid = threadIdx.x;
  op.init(P, tid, lds2);''','''static __device__ unsigned long long g_stamps[256];
template <class Op, class = void> struct StampOf { static constexpr int v = 13; };
template <class Op> struct StampOf<Op, decltype((void)Op::STAMP_ID)> { static constexpr int v = Op::STAMP_ID; };
#define DDRL_STAMP_READER(tu) extern "C" int32_t ddrl_debug_stamps_##tu(unsigned long long* out, int reset) { hipMemcpyFromSymbol(out, HIP_SYMBOL(ddrl::g_stamps), 256 * 8); if (reset) { unsigned long long z[256] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ddrl::g_stamps), z, 256 * 8); } return 0; }
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <class Op>
__global__ __launch_bounds__(Op::THREADS, OccOf<Op>::v) void engine2_kernel(typename Op::Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds2[];
  Op op;
  const int tid = threadIdx.x;
  unsigned long long T0 = stamp(), tc = 0, tm = 0, tb = 0, tw = 0, tf = 0, tp, te;
  op.init(P, tid, lds2);''')
a=s.index('  __syncthreads();\n  int buf = 0;')
b=s.index('// Ops may declare `static constexpr int EXTRA`')
s=s[:a]+'''  __syncthreads();
  tp = stamp();
  int buf = 0;
  for (; kb < kbe; ++kb) {
    unsigned long long a = stamp(), b0, b, c0, c, m0, m1;
    op.extra(lds2 + buf * Op::STAGE);
    if constexpr (HasPreEpilogue<Op>::v) {
      if (kb == kbe - 1) op.pre_epilogue(P);
    }
    if constexpr (!CommitFirstOf<Op>::v) {
      m0 = stamp();
      compute_block<Op>(op, lds2 + buf * Op::STAGE, acc);
      __builtin_amdgcn_sched_barrier(0);
      m1 = stamp();
    }
    b0 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    b = stamp();
    c0 = b;
    if (kb + 1 < kbe) {
      op.commit(regs, lds2 + (buf ^ 1) * Op::STAGE);
      __builtin_amdgcn_sched_barrier(0);
      c0 = stamp();
      if (kb + 2 < kbe) op.fetch(P, kb + 2, regs);
    }
    __builtin_amdgcn_sched_barrier(0);
    c = stamp();
    if constexpr (CommitFirstOf<Op>::v) {
      m0 = c;
      compute_block<Op>(op, lds2 + buf * Op::STAGE, acc);
      __builtin_amdgcn_sched_barrier(0);
      m1 = stamp();
    }
    unsigned long long c2 = stamp();
    __syncthreads();
    unsigned long long d = stamp();
    tc += m1 - m0; tw += b - b0; tm += c0 - b; tf += c - c0; tb += d - c2;
    (void)a;
    buf ^= 1;
  }
  te = stamp();
  op.epilogue(P, acc, lds2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long tz = stamp();
  if ((tid & 63) == 0) {
    const int o = StampOf<Op>::v * 8;
    atomicAdd(&g_stamps[o + 0], tp - T0);
    atomicAdd(&g_stamps[o + 1], tc);
    atomicAdd(&g_stamps[o + 2], tm);
    atomicAdd(&g_stamps[o + 3], tb);
    atomicAdd(&g_stamps[o + 4], tz - te);
    atomicAdd(&g_stamps[o + 5], tz - T0);
    atomicAdd(&g_stamps[o + 6], 1ull);
    atomicAdd(&g_stamps[o + 7], tw);
    atomicAdd(&g_stamps[200 + StampOf<Op>::v], tf);
  }
}

'''+s[b:]
assert "g_stamps[200" in s
open('engine2.h','w').write(s)
order=['ConvFwd2v2','ConvFwd3v2','ConvFwd1v2','ConvDgrad3v2','ConvDgrad2v2','ConvWgrad1v2','ConvWgrad2v2','ConvWgrad3v2','FcFwd2','FcDgrad2','FcWgrad2']
for f,tu in (('conv2.hip','conv2'),('wgrad2.hip','wgrad2'),('fc2.hip','fc2')):
    t=open(f).read()
    for i,name in enumerate(order):
        t=re.sub(r"(struct %s[^{]*\{)"%name, r"\1\n  static constexpr int STAMP_ID = %d;"%i, t)
    t+="\nDDRL_STAMP_READER(%s)\n"%tu
    open(f,'w').write(t)
PY
make -j4 2>&1 | grep -E "error" -A5 || true
cp libddrl_hip.so $ROOT/ddrl4nav_amd/csrc/libddrl_hip_diag.so
echo built $ROOT/ddrl4nav_amd/csrc/libddrl_hip_diag.so
