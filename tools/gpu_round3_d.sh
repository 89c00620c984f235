#!/bin/bash
# round 3: A/B of the four-waves-per-SIMD experiment (TeamPair, 128 registers, control table in global memory) + PMC evidence of the f32x3 mode
mkdir -p gpurun_out/r3d
bash tools/ab_solve.sh "--batch 12288 --reps 2 --mlp-dtype f32x3" - build/libsdempc_w4.so 2>&1 | tee gpurun_out/r3d/ab_w4.log
SDEMPC_LIB=build/libsdempc_w4.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_f32x3 or ticketed" 2>&1 | tail -3 | tee -a gpurun_out/r3d/ab_w4.log
bash tools/profile_round.sh r3d_prof 12288 f32x3 2>&1 | tail -5
