#!/bin/bash
# round 3, second GPU pass: A/B of the f32x3 chain placement (overlap beside the tanh of an independent tile), then the GPU suite on the in-tree build
mkdir -p gpurun_out/r3b
bash tools/ab_solve.sh "--batch 12288 --reps 2 --mlp-dtype f32x3" build/libsdempc_x3o0.so - build/libsdempc_x3s10.so build/libsdempc_x3fwd.so 2>&1 | tee gpurun_out/r3b/ab.log
timeout -k 10 200 python tools/prof_solve.py --mode solve --batch 12288 --reps 2 --mlp-dtype f32 2>&1 | grep -v amdgpu | tail -2 | tee -a gpurun_out/r3b/ab.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3b/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r3b/pytest.log
tail -5 gpurun_out/r3b/pytest.log
