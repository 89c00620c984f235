#!/bin/bash
# A/B of library variants in build/: fixed-work timing and the full 200-iteration solve, f32x3
mkdir -p gpurun_out/r3k
out=gpurun_out/r3k/out.txt
for v in "$@"; do
  echo "== $v" >> $out
  SDEMPC_LIB=build/libsdempc_$v.so timeout -k 10 200 python tools/prof_solve.py --batch 12288 --max-iter 50 --fixed-work --reps 2 --mlp-dtype f32x3 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED $v" >> $out; cat $out; exit 1; }
  SDEMPC_LIB=build/libsdempc_$v.so timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 2 --mlp-dtype f32x3 2>&1 | grep -v amdgpu.ids >> $out || { echo "FAILED $v" >> $out; cat $out; exit 1; }
done
cat $out
