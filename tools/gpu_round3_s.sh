#!/bin/bash
# A/B of library variants at the power cap: solves/s, shader clock, package power   usage: gpu_round3_s.sh mlp variant...   (variant "intree" = the in-tree build)
mkdir -p gpurun_out/r3s
mlp=$1; shift
out=gpurun_out/r3s/out.txt
n=0
for v in "$@"; do
  n=$((n+1))
  echo "== $v ($mlp)" >> $out
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  ( for i in $(seq 1 13); do sleep 1; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: [^ ]* (\([0-9]*\)Mhz).*/\1 MHz/' -e 's/.*Power (W): \([0-9.]*\).*/\1 W/' | tr '\n' ' '; echo; done ) > gpurun_out/r3s/smi_${n}_$v.txt &
  spid=$!
  SDEMPC_LIB=$lib timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 3 --mlp-dtype $mlp 2>&1 | grep -v amdgpu.ids >> $out || { echo FAILED >> $out; }
  wait $spid
  awk '$3 > 1300 {c += $1; w += $3; k++} END {if (k) printf "   at the cap: %d samples, sclk mean %.0f MHz, power mean %.0f W\n", k, c / k, w / k}' gpurun_out/r3s/smi_${n}_$v.txt >> $out
done
cat $out
