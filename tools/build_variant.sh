#!/bin/bash
# Builds a copy of libddrl_hip.so with extra compiler flags into tools/_scratch_abl/<name>.so (git-ignored, travels with
# gpurun); tools that accept DDRL_ABL_LIB load it instead of the in-tree library.
# usage: bash tools/build_variant.sh <name> "<extra flags>"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result -Xclang -target-feature -Xclang -load-store-opt"  # = the Makefile's flags
D=/tmp/ddrl_abl_$1
rm -rf $D && mkdir -p $D/ddrl4nav_amd/csrc $D/include
cp $ROOT/ddrl4nav_amd/csrc/*.hip $ROOT/ddrl4nav_amd/csrc/*.h $ROOT/ddrl4nav_amd/csrc/*.cpp $ROOT/ddrl4nav_amd/csrc/Makefile $D/ddrl4nav_amd/csrc/
cp $ROOT/include/ddrl.h $D/include/
make -C $D/ddrl4nav_amd/csrc -j8 CXXFLAGS="$BASE $2" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
mkdir -p $ROOT/tools/_scratch_abl && cp $D/ddrl4nav_amd/csrc/libddrl_hip.so $ROOT/tools/_scratch_abl/$1.so
echo built tools/_scratch_abl/$1.so
