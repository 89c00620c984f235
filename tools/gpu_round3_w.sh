#!/bin/bash
# mid-size batches: one group per wave (SDEMPC_DUO=0) against the duo layout (SDEMPC_DUO=1), three arithmetics
mkdir -p gpurun_out/r3w
out=gpurun_out/r3w/out.txt
for mlp in f32x3 f32; do for B in 384 512 768 1024 1280; do
  for duo in 0 1; do
    export SDEMPC_DUO=$duo
    echo "== $mlp B=$B duo=$duo" >> $out
    timeout -k 10 120 python tools/prof_solve.py --batch $B --reps 2 --mlp-dtype $mlp 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
  done
done; done
cat $out
