#!/bin/bash
# rocprofv3 counter passes for the hot-path kernels (run on the GPU box through gpurun).
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; SQ has 8 slots).
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$REPO/$OUT" -o "$name" -- python3 "$REPO/tools/profile_iter.py" 2 4 > "$REPO/$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
ls "$REPO/$OUT"
