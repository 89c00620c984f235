#!/bin/bash
# A/B build of the exact-mode kernels with extra compiler flags, beside the in-tree build:
#   tools/build_variant.sh NAME "-DSDEMPC_VAR_X=1"   ->  build/libsdempc_NAME.so   (select with SDEMPC_LIB=build/libsdempc_NAME.so)
# Only the exact-mode kernel objects (sdempc_kernels.o, _duo2.o, _duo4.o, _duo6.o) are recompiled; the other objects are taken from the in-tree build (which must be current).
set -e
name=${1:?name}; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
dir=/tmp/sdempc_var_$name
rm -rf $dir; mkdir -p $dir/pkg/csrc $dir/include
cp -p $root/include/*.h $dir/include/
cp -p $root/sde4mbrl_px4_amd/csrc/* $dir/pkg/csrc/
rm -f $dir/pkg/csrc/sdempc_kernels.o $dir/pkg/csrc/sdempc_kernels_duo2.o $dir/pkg/csrc/sdempc_kernels_duo4.o $dir/pkg/csrc/sdempc_kernels_duo6.o $dir/pkg/csrc/libsdempc.so
# ONLY_MAIN=1: the knob only concerns translation unit 0 (latency layouts, rollout / gradient kernels): keep the in-tree duo objects
if [ "${ONLY_MAIN:-0}" = 1 ]; then cp -p $root/sde4mbrl_px4_amd/csrc/sdempc_kernels_duo*.o $dir/pkg/csrc/; touch $dir/pkg/csrc/sdempc_kernels_duo*.o; fi
touch $dir/pkg/csrc/sdempc_kernels_fast*.o $dir/pkg/csrc/sdempc_prng.o $dir/pkg/csrc/sdempc_api.o   # (the fast-mode objects keep the in-tree knobs)
make -j4 -C $dir/pkg/csrc EXTRA="$extra" libsdempc.so > $dir/build.log 2>&1
mkdir -p $root/build && cp $dir/pkg/csrc/libsdempc.so $root/build/libsdempc_$name.so
ls -la $root/build/libsdempc_$name.so
