#!/bin/bash
# FETCH_SIZE of the dense kernels for the given library variants
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = real ]; then unset DDRL_ABL_LIB; else export DDRL_ABL_LIB=$REPO/tools/_scratch_abl/$n.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/r2s2/fetch_$n -o f -- python3 $REPO/tools/profile_iter.py 2 0 > /dev/null 2>&1
  python3 - <<P
import csv,glob,collections
d=collections.defaultdict(lambda:[0,0])
for f in glob.glob("$REPO/gpurun_out/r2s2/fetch_$n/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for t in ("fc_fwd_planes","fc_dgrad_planes","fc_wgrad_planes","conv_dgrad3_planes","conv_wgrad1_planes"):
            if t in k: d[t][0]+=float(r["Counter_Value"]); d[t][1]+=1
print("$n", {k:round(v[0]/max(1,v[1])*1024/1e9,2) for k,v in d.items()}, "GB FETCH_SIZE per launch")
P
done
