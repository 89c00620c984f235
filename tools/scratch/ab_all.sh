#!/bin/bash
# A/B of two library builds over the arithmetic modes of C2 (12,288 instances, third launch of three) and the secondary configurations
mkdir -p gpurun_out/ab_all; out=gpurun_out/ab_all/out.txt; : > $out
run() { # lib mlp math config batch reps
  SDEMPC_LIB=$1 timeout -k 10 250 python tools/prof_solve.py --config configs/$4 --batch $5 --reps $6 --mlp-dtype $2 --math-mode $3 2>&1 | grep "rep $(( $6 - 1 ))" | sed -e 's/work:.*//' >> $out
}
for v in r5a intree r5a intree; do
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  for spec in "f32x3 fast" "f32 fast" "f16 fast" "f32x3 exact" "f32 exact"; do
    set -- $spec; echo "== $v c2 $1/$2" >> $out; run $lib $1 $2 c2_iris_traj_h50_p128.yaml 12288 3
  done
done
for v in r5a intree; do
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  echo "== $v c3 f32x3/fast" >> $out; run $lib f32x3 fast c3_hexa_traj_h50_p256.yaml 6144 2
  echo "== $v c5 f32x3/fast" >> $out; run $lib f32x3 fast c5_iris_traj_h200_p1024.yaml 768 2
  echo "== $v c5 f16/fast" >> $out; run $lib f16 fast c5_iris_traj_h200_p1024.yaml 768 2
done
cat $out
