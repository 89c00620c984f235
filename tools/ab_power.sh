#!/bin/bash
# A/B of library builds AT THE POWER CAP on one GPU box: solves/s of full C2 throughput launches with shader clock and package power sampled beside each.
#   [PROF_ARGS="--math-mode fast --fixed-work"] tools/ab_power.sh <mlp_dtype> <variant>...      variant = NAME of build/libsdempc_NAME.so (tools/build_variant.sh), or "intree" for the in-tree build
# Output: gpurun_out/ab_power/out.txt (one block per variant: prof_solve.py's lines + the mean clock / power of the samples above 1,300 W)
mkdir -p gpurun_out/ab_power
mlp=${1:?mlp_dtype}; shift
out=gpurun_out/ab_power/out.txt
n=0
for v in "$@"; do
  n=$((n+1))
  echo "== $v ($mlp)" >> $out
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  ( for i in $(seq 1 13); do sleep 1; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*sclk clock level: [^ ]* (\([0-9]*\)Mhz).*/\1 MHz/' -e 's/.*Power (W): \([0-9.]*\).*/\1 W/' | tr '\n' ' '; echo; done ) > gpurun_out/ab_power/smi_${n}_$v.txt &
  spid=$!
  SDEMPC_LIB=$lib timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 3 --mlp-dtype $mlp $PROF_ARGS 2>&1 | grep -v amdgpu.ids >> $out || { echo FAILED >> $out; }
  wait $spid
  awk '$3 > 1300 {c += $1; w += $3; k++} END {if (k) printf "   at the cap: %d samples, sclk mean %.0f MHz, power mean %.0f W\n", k, c / k, w / k}' gpurun_out/ab_power/smi_${n}_$v.txt >> $out
done
cat $out
