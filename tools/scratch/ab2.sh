#!/bin/bash
# tools/scratch/ab2.sh <mlp> <variant>... : third launch of three, C2 12,288 instances, math_mode fast
mkdir -p gpurun_out/ab2; out=gpurun_out/ab2/out_$1.txt; : > $out; mlp=$1; shift
for rep in 1 2; do for v in "$@"; do
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  echo "== $v ($mlp)" >> $out
  SDEMPC_LIB=$lib timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 3 --mlp-dtype $mlp --math-mode fast 2>&1 | grep "rep 2" | sed -e 's/work:.*//' >> $out
done; done
cat $out
