#!/bin/bash
# Same-box A/B of per-kernel times: alternates library builds (A B A B) after one warm-up run, because boxes
# differ by up to 5 % and the first run on a fresh box is slower.  Builds are copies of libddrl_hip.so kept
# under tools/_scratch_abl/<name>.so (git-ignored, they travel with gpurun).
# usage: bash tools/ab_kernels.sh name1 name2 ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 $ROOT/tools/ablate_iter.py warmup > /dev/null 2>&1
for rep in 1 2; do
  for n in "$@"; do
    DDRL_ABL_LIB=$ROOT/tools/_scratch_abl/$n.so python3 $ROOT/tools/ablate_iter.py $n 2>&1 | grep -v amdgpu.ids
  done
done
