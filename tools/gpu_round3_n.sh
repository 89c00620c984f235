#!/bin/bash
# round 3, final library (six-team workgroups): whole GPU suite, then the profiling round (kernel trace + PMC passes)
mkdir -p gpurun_out/r3n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3n/pytest.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/r3n/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/profile_round.sh r3_final 12288 f32x3
