#!/bin/bash
# small batches of the matrix-pipe modes on one group per wave: latency of B = 1, 8, 64, 256 (f32x3, f16), then the parity tests of those modes
mkdir -p gpurun_out/r3v
out=gpurun_out/r3v/out.txt
for mlp in f32x3 f16; do for B in 1 8 64 256; do
  for duo in auto 1; do
    if [ $duo = 1 ]; then export SDEMPC_DUO=1; else unset SDEMPC_DUO; fi
    echo "== $mlp B=$B duo=$duo" >> $out
    timeout -k 10 120 python tools/prof_solve.py --batch $B --reps 2 --mlp-dtype $mlp 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
  done
done; done
unset SDEMPC_DUO
cat $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix_pipe or f32x3 or f16 or duo or golden" > gpurun_out/r3v/pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r3v/pytest.log
