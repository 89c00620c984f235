#!/bin/bash
# clocks and power while the throughput kernel runs (is the held shader clock a power limit?)
mkdir -p gpurun_out/r3q
out=gpurun_out/r3q/out.txt
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > gpurun_out/r3q/idle.txt 2>&1
for mlp in f32x3 f32 f16; do
  echo "== $mlp" >> $out
  ( for i in $(seq 1 16); do sleep 1; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; done ) >> gpurun_out/r3q/smi_$mlp.txt &
  spid=$!
  timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 3 --mlp-dtype $mlp 2>&1 | grep -v amdgpu.ids >> $out
  wait $spid
done
cat $out; head -30 gpurun_out/r3q/idle.txt; for mlp in f32x3 f32 f16; do echo "-- $mlp"; cat gpurun_out/r3q/smi_$mlp.txt | cut -c1-400 | sed -n 6,14p; done
