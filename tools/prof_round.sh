#!/bin/bash
# One gpurun call's worth of profiling for a round tag (e.g. r03_v1): the bench line, the same command under
# rocprofv3 --kernel-trace --stats, and the PMC passes of tools/prof_pmc.sh.  Results land in gpurun_out/<tag>/;
# tools/pmc_to_profiles.py + a copy of the summaries into profiles/ follow on the build host.
# usage (on the GPU box, through gpurun):  bash tools/prof_round.sh r03_v1
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
# the build the summaries are stamped with (tools/pmc_to_profiles.py refuses "unknown"): fail before the passes, not after them
BUILD=${DDRL_PROFILE_BUILD:-$(git rev-parse --short HEAD 2>/dev/null || true)}
if [ -z "$BUILD" ]; then echo "set DDRL_PROFILE_BUILD=<git short hash> (the GPU box has no .git)" >&2; exit 2; fi
# what the summaries must carry (tools/pmc_to_profiles.py): the profiled build and the batch of the PMC passes (tools/profile_iter.py: 65,536)
echo "{\"batch\": 65536, \"build\": \"$BUILD ($TAG)\"}" > "$OUT/meta.json"
# socket power / shader clock / junction temperature of every hwmon node every ~25 ms WHILE the bench line is measured: the headline runs
# at the socket's power cap, and this trace is the evidence; tools/hwmon_trace.py --summary picks this GPU's node
rm -f "$OUT/.bench_done"
python3 tools/hwmon_trace.py "$OUT/hwmon_all_nodes.txt" "$OUT/.bench_done" &
SAMPLER=$!
python3 bench.py --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
touch "$OUT/.bench_done"; wait $SAMPLER 2>/dev/null; rm -f "$OUT/.bench_done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-async --no-ingest --no-nav > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err" ); echo "kernel-trace rc=$?"
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
bash tools/prof_pmc.sh "gpurun_out/$TAG/pmc"; echo "pmc rc=$?"
hostname > "$OUT/box.txt"; rocm-smi --showproductname --showuniqueid 2>/dev/null | grep -E "Card Model|Card SKU|Unique ID" | head -6 >> "$OUT/box.txt"
PCI=$(rocm-smi --showbus 2>/dev/null | grep -oE "[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9]" | head -1); echo "pci $PCI" >> "$OUT/box.txt"
python3 tools/hwmon_trace.py --summary "$OUT/hwmon_all_nodes.txt" "$PCI" > "$OUT/hwmon_summary.json" 2>&1; cat "$OUT/hwmon_summary.json" | head -30
ls "$OUT" "$OUT/pmc" | head -40
