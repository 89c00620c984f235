#!/bin/bash
# round 3, first GPU pass of the f32x3 mode: parity of the matrix-pipe modes, then interleaved timing f32 / f32x3 / f16 at C2
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "matrix_pipe or f32x3 or c5_f16" > gpurun_out/r3a/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
for rep in 1 2; do
  for mlp in f32 f32x3 f16; do
    timeout -k 10 200 python tools/prof_solve.py --mode solve --batch 12288 --reps 2 --mlp-dtype $mlp 2>&1 | grep -v amdgpu | tail -2 | tee -a gpurun_out/r3a/timing.log
  done
done
