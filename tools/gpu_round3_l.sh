#!/bin/bash
# six-team workgroups (TeamHex) against two-team workgroups: timing of the C2 throughput launch + the large-batch parity tests
mkdir -p gpurun_out/r3l
out=gpurun_out/r3l/out.txt
for hex in 1 0 1 0; do
  echo "== SDEMPC_HEX=$hex" >> $out
  SDEMPC_HEX=$hex timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 2 --mlp-dtype f32x3 2>&1 | grep -v amdgpu.ids >> $out || { echo FAILED >> $out; cat $out; exit 1; }
done
echo "== f32 hex=1" >> $out
SDEMPC_HEX=1 timeout -k 10 200 python tools/prof_solve.py --batch 12288 --reps 2 --mlp-dtype f32 2>&1 | grep -v amdgpu.ids >> $out
cat $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ticketed or workspaces_scale" > gpurun_out/r3l/pytest.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r3l/pytest.log
