#!/bin/bash
# kernel durations of heads_loss / heads_reduce for library builds under tools/_scratch_abl/<name>.so (rocprofv3 --kernel-trace --stats of tools/ablate_iter.py)
# usage (GPU box):  bash tools/prof_heads_kernels.sh name1 name2 ...   (profiles/r06_heads_loss_tr_ab.txt)
ROOT=$(pwd)
export TMPDIR=/tmp
for n in "$@"; do
  export DDRL_ABL_LIB=$ROOT/tools/_scratch_abl/$n.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ph_$n -o ph -- python3 $ROOT/tools/ablate_iter.py $n > $ROOT/gpurun_out/ph_$n.log 2>&1
  f=$(find $ROOT/gpurun_out/ph_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n"
  if [ -n "$f" ]; then grep -E "heads_loss|heads_reduce" "$f" < /dev/null; else echo "no kernel_stats.csv"; fi
done
