#!/bin/bash
# rocprofv3 kernel stats of the robot_nav PPO iteration, one stream (tools/bench_nav.py 4096 4096 4 - nav1d - one-stream): per-kernel
# time per iteration without the two-stream overlap.  usage (GPU box): bash tools/nav_kstats.sh <tag>
OUT=$(pwd)/gpurun_out/${1:-nav_kstats}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$REPO/tools/bench_nav.py" 4096 4096 4 - nav1d - one-stream > "$OUT/run.json" 2> "$OUT/kt.err"
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*.db" -delete; find "$OUT/kt" -name "*kernel_trace.csv" -delete
python3 - "$OUT/kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# 8 iterations (warm-up learn + timed learn, 4 each) + 1 forward
for r in rows[:32]:
    print("%-100s calls %4s  per-iter %7.3f ms  avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 8e6, float(r["AverageNs"]) / 1e3))
P
cut -c1-200 "$OUT/run.json" | tail -1
