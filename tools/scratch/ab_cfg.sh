#!/bin/bash
# A/B of library builds on the secondary configurations (C3, C5; math_mode fast, f32x3 / f16): tools/scratch/ab_cfg.sh <variant>...
mkdir -p gpurun_out/ab_cfg; out=gpurun_out/ab_cfg/out.txt; : > $out
for rep in 1 2; do
for v in "$@"; do
  lib=build/libsdempc_$v.so; [ "$v" = intree ] && lib=sde4mbrl_px4_amd/csrc/libsdempc.so
  for spec in "c3_hexa_traj_h50_p256.yaml 6144 f32x3" "c5_iris_traj_h200_p1024.yaml 768 f32x3" "c5_iris_traj_h200_p1024.yaml 768 f16"; do
    set -- $spec
    echo "== $v $1 $3" >> $out
    SDEMPC_LIB=$lib timeout -k 10 200 python tools/prof_solve.py --config configs/$1 --batch $2 --reps 2 --mlp-dtype $3 --math-mode fast 2>&1 | grep "rep 1" >> $out
  done
done
done
cat $out
