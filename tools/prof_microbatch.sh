#!/bin/bash
# HBM traffic of the Pong PPO iteration as ONE launch set of 65,536 samples against 32 launch sets of 2,048 ("record the
# PMC FETCH_SIZE of both forms"): separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/ab_microbatch.py, summed per kernel
# and per pass over the 65,536 samples.  usage (GPU box): bash tools/prof_microbatch.sh <tag>
OUT=$(pwd)/gpurun_out/${1:-mb_pmc}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mb in 65536 2048; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc" -o "${c}_$mb" -- python3 "$REPO/tools/ab_microbatch.py" 65536 $mb > "$OUT/${c}_$mb.log" 2>&1
    echo "pmc $c mb=$mb rc=$?"
  done
done
python3 - "$OUT" <<'P'
import csv, glob, json, re, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for mb in (65536, 2048):
    per = defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/pmc/**/%s_%d_counter_collection.csv" % (out, c, mb), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            m = re.search(r"ddrl::(\w+?)(?:_kernel)?[<(]", r["Kernel_Name"])
            if m and r["Counter_Name"] == c:
                per[m.group(1)][c] += float(r["Counter_Value"]) * 1024.0      # KB -> bytes
    passes = 5.0    # tools/ab_microbatch.py: 1 warm-up + 3 timed + 1 event-profiled pass over the 65,536 samples
    res[mb] = {k: {"fetch_gb": round(v["FETCH_SIZE"] / passes / 1e9, 3), "write_gb": round(v["WRITE_SIZE"] / passes / 1e9, 3),
                   "corrected_gb": round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / passes / 1e9, 3)} for k, v in per.items()
               if v["FETCH_SIZE"] + v["WRITE_SIZE"] > passes * 5e7}
    res[mb]["_iteration_corrected_gb"] = round(sum(v["corrected_gb"] for v in res[mb].values()), 2)
json.dump(res, open(out + "/microbatch_pmc.json", "w"), indent=1, sort_keys=True)
for mb in res:
    print(mb, "corrected GB per 65,536 samples:", res[mb]["_iteration_corrected_gb"])
    for k, v in sorted(res[mb].items()):
        if not k.startswith("_"):
            print("   %-28s fetch %7.3f  write %7.3f  corrected %7.3f" % (k, v["fetch_gb"], v["write_gb"], v["corrected_gb"]))
P
find "$OUT/pmc" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete
