#!/bin/bash
# Timing-only knock-out study of the tile engine (engine2.h): builds copies of the library with one phase
# of the k loop removed (-DDDRL_ABL_NOFETCH / NOCOMMIT / NOBARRIER / NOEPILOGUE; the results of such a
# build are WRONG, only its kernel times mean something) and prints the per-kernel times of each next to
# the real build's.  `bash tools/ablate_engine.sh build` here (hipcc cross-compiles, ~90 s per variant) puts
# the copies under tools/_scratch_abl/ (git-ignored, travels with gpurun); `bash tools/ablate_engine.sh run`
# on the GPU box times them.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result -Xclang -target-feature -Xclang -load-store-opt"  # = the Makefile's flags
variant() {  # name, extra flags
  D=/tmp/ddrl_abl_$1
  rm -rf $D && mkdir -p $D/ddrl4nav_amd/csrc $D/include
  cp $ROOT/ddrl4nav_amd/csrc/*.hip $ROOT/ddrl4nav_amd/csrc/*.h $ROOT/ddrl4nav_amd/csrc/*.cpp $ROOT/ddrl4nav_amd/csrc/Makefile $D/ddrl4nav_amd/csrc/
  cp $ROOT/include/ddrl.h $D/include/
  make -C $D/ddrl4nav_amd/csrc -j8 CXXFLAGS="$BASE $2" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
  mkdir -p $ROOT/tools/_scratch_abl && cp $D/ddrl4nav_amd/csrc/libddrl_hip.so $ROOT/tools/_scratch_abl/$1.so
}
if [ "$1" = run ]; then
  python3 $ROOT/tools/ablate_iter.py warmup > /dev/null
  python3 $ROOT/tools/ablate_iter.py real
fi
for v in "nofetch:-DDDRL_ABL_NOFETCH" "nocommit:-DDDRL_ABL_NOFETCH -DDDRL_ABL_NOCOMMIT" \
         "nobarrier:-DDDRL_ABL_NOFETCH -DDDRL_ABL_NOCOMMIT -DDDRL_ABL_NOBARRIER" \
         "bare:-DDDRL_ABL_NOFETCH -DDDRL_ABL_NOCOMMIT -DDDRL_ABL_NOBARRIER -DDDRL_ABL_NOEPILOGUE" \
         "noepi:-DDDRL_ABL_NOEPILOGUE" "nobar_only:-DDDRL_ABL_NOBARRIER"; do
  name=${v%%:*}; flags=${v#*:}
  if [ "$1" = build ]; then variant $name "$flags"; fi
  if [ "$1" = run ]; then DDRL_ABL_LIB=$ROOT/tools/_scratch_abl/$name.so python3 $ROOT/tools/ablate_iter.py $name; fi
done
if [ "$1" = run ]; then python3 $ROOT/tools/ablate_iter.py real; fi
