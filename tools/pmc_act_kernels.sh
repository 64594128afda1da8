#!/bin/bash
# counter passes over the acting convolutions for library builds under tools/_scratch_abl/<name>.so (40 forwards at n = 256 through tools/profile_iter.py)
# usage (GPU box):  bash tools/pmc_act_kernels.sh name1 name2 ...   (profiles/r06_act_convs_ab.txt)
ROOT=$(pwd)
export TMPDIR=/tmp
for n in "$@"; do
  export DDRL_ABL_LIB=$ROOT/tools/_scratch_abl/$n.so
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
    tag=$(echo $set | cut -d' ' -f1)
    echo "== $n : $set"
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/gpurun_out/pm_${n}_$tag -o pm -- python3 $ROOT/tools/profile_iter.py 0 40 1024 > $ROOT/gpurun_out/pm_$n.log 2>&1 < /dev/null
    python3 - $ROOT/gpurun_out/pm_${n}_$tag <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
d = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "act_convs" in k:
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for c, v in sorted(d.items()):
    print("  %-28s median %.4g  (n %d)" % (c, statistics.median(v), len(v)))
PY
    rm -rf $ROOT/gpurun_out/pm_${n}_$tag
  done
done
