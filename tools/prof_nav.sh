#!/bin/bash
# rocprofv3 passes over the robot_nav PPO iteration (tools/bench_nav.py 4096 4096 2): kernel trace + stats, then separate PMC passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  usage (GPU box): bash tools/prof_nav.sh r04_nav
set -u
TAG=${1:-rXX_nav}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
# the build the summaries are stamped with (tools/pmc_to_profiles.py refuses "unknown"): fail before the passes, not after them
BUILD=${DDRL_PROFILE_BUILD:-$(git rev-parse --short HEAD 2>/dev/null || true)}
if [ -z "$BUILD" ]; then echo "set DDRL_PROFILE_BUILD=<git short hash> (the GPU box has no .git)" >&2; exit 2; fi
echo "{\"batch\": 4096, \"build\": \"$BUILD ($TAG)\"}" > "$OUT/meta.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$REPO/tools/bench_nav.py" 4096 4096 2 > "$OUT/nav_under_rocprof.json" 2> "$OUT/kt.err"; echo "kernel-trace rc=$?"
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc" -o "$name" -- python3 "$REPO/tools/bench_nav.py" 4096 4096 1 > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?"; }
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
hostname > "$OUT/box.txt"; rocm-smi --showproductname --showuniqueid 2>/dev/null | grep -E "Card Model|Card SKU|Unique ID" | head -6 >> "$OUT/box.txt"
find "$OUT" -name "*.db" -delete; find "$OUT" -size +8M -delete
ls "$OUT" "$OUT/pmc" | head -30
