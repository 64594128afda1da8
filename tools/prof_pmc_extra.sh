#!/bin/bash
# One more rocprofv3 counter pass over the Pong iteration (tools/profile_iter.py), into the pmc directory of an existing profile:
# where the waves' cycles go besides the matrix pipe (VERDICT r5 item 1's list).  Counter names are checked against `rocprofv3 -L`
# first: an unknown name would fail the whole pass.
# usage (GPU box):  bash tools/prof_pmc_extra.sh gpurun_out/r06_v2/pmc
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$REPO/$OUT/counters_available.txt" 2>&1
WANT="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES"
HAVE=""
for c in $WANT; do if grep -qw "$c" "$REPO/$OUT/counters_available.txt"; then HAVE="$HAVE $c"; else echo "not on this build: $c"; fi; done
echo "collecting:$HAVE"
rocprofv3 --kernel-trace --pmc $HAVE --output-format csv -d "$REPO/$OUT" -o sq3 -- python3 "$REPO/tools/profile_iter.py" 2 4 > "$REPO/$OUT/sq3.log" 2>&1; echo "sq3 rc=$?"
ls "$REPO/$OUT" | grep sq3
