#!/bin/bash
# How much of a one-stream robot_nav PPO iteration is GPU idle time between kernels?  rocprofv3 kernel trace of
# `bench_nav.py 4096 4096 4 - nav1d - one-stream`, then sum of kernel durations / iterations against the wall time per iteration.
OUT=$(pwd)/gpurun_out/${1:-nav_gap}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 "$REPO/tools/bench_nav.py" 4096 4096 4 - nav1d - one-stream > "$OUT/run.json" 2> "$OUT/kt.err"
python3 - "$OUT" <<'P'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST learn call = the last 4 iterations: take kernels from the last 4 clip_adam launches backwards
idx = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r["Kernel_Name"]]
lo, hi = idx[-5] + 1, idx[-1]
seg = rows[lo:hi + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])), a["Kernel_Name"][:60], b["Kernel_Name"][:60]) for a, b in zip(seg, seg[1:]))
print(json.dumps({"iterations": 4, "kernels": len(seg), "span_ms_per_iter": span / 4e6, "busy_ms_per_iter": busy / 4e6,
                  "idle_ms_per_iter": (span - busy) / 4e6, "largest_gaps_us": [(g[0] / 1e3, g[1], g[2]) for g in gaps[-8:]]}))
P
tail -c 600 "$OUT/run.json"
