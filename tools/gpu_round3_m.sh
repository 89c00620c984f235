#!/bin/bash
# (1) the K = 8 bf16 matrix instruction against the model (one group of eight products?)  (2) timing of layer 1 on the matrix pipe
mkdir -p gpurun_out/mfma8 gpurun_out/r3m
for f in build/mfma8/bf16_*.bin; do
  n=$(basename "$f" .bin)
  timeout -k 10 120 tools/mfma16_study/mfma16_probe "$f" "gpurun_out/mfma8/$n.k8.out" 8 || exit 1
done
out=gpurun_out/r3m/out.txt
for rep in 1 2; do
  timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 2 --mlp-dtype f32x3 2>&1 | grep -v amdgpu.ids >> $out || { echo FAILED >> $out; cat $out; exit 1; }
done
timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 2 --mlp-dtype f32 2>&1 | grep -v amdgpu.ids >> $out
timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 1 --mlp-dtype f16 2>&1 | grep -v amdgpu.ids >> $out
cat $out
