#!/bin/bash
# round 3: A/B of scheduling experiments on the f32x3 throughput kernel (C2, 12,288 instances per launch)
mkdir -p gpurun_out/r3e
SDEMPC_LIB=build/libsdempc_w4.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_f32x3 or ticketed or full_size_throughput" 2>&1 | tail -3 | tee gpurun_out/r3e/w4_tests.log
if grep -q "passed" gpurun_out/r3e/w4_tests.log && ! grep -q "failed" gpurun_out/r3e/w4_tests.log; then W4=build/libsdempc_w4.so; else W4=""; fi
bash tools/ab_solve.sh "--batch 12288 --reps 2 --mlp-dtype f32x3" - build/libsdempc_pf.so build/libsdempc_plc.so build/libsdempc_pfa.so $W4 2>&1 | tee gpurun_out/r3e/ab.log
if [ -n "$W4" ]; then bash tools/ab_solve.sh "--batch 16384 --reps 2 --mlp-dtype f32x3" - $W4 2>&1 | tee -a gpurun_out/r3e/ab.log; fi
