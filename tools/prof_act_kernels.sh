#!/bin/bash
# kernel durations (min / p10 / median / p90) of the acting forward for library builds kept under tools/_scratch_abl/<name>.so: rocprofv3 kernel trace of tools/ab_bench.py
# usage (GPU box):  bash tools/prof_act_kernels.sh name1 name2 ...   (profiles/r06_act_convs_ab.txt, r06_heads_act_waves_ab.txt)
ROOT=$(pwd)
export TMPDIR=/tmp
for n in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/pa_$n -o pa -- python3 $ROOT/tools/ab_bench.py $ROOT/tools/_scratch_abl/$n.so > $ROOT/gpurun_out/pa_$n.log 2>&1 < /dev/null
  echo "== $n"
  python3 - $ROOT/gpurun_out/pa_$n <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no kernel trace"); sys.exit(0)
d = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "act_convs" in k or "heads_act" in k or "fc_fwd_planes_kernel<true>" in k:
        d.setdefault((k, r.get("Grid_Size_X") or r.get("Workgroup_Size_X")), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g), v in sorted(d.items()):
    v.sort()
    print("%-45s grid %-8s calls %5d  min %.1f  p10 %.1f  median %.1f  p90 %.1f us" % (k[-45:], g, len(v), v[0] / 1e3, v[len(v) // 10] / 1e3, statistics.median(v) / 1e3, v[len(v) * 9 // 10] / 1e3))
PY
  rm -rf $ROOT/gpurun_out/pa_$n
done
