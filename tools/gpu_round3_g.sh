#!/bin/bash
# round 3: GPU suite on the final tree, then the rocprofv3 evidence (kernel trace + PMC passes) of the reported arithmetic on the same library
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r3g/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r3g/pytest.log
tail -4 gpurun_out/r3g/pytest.log
if grep -q " failed" gpurun_out/r3g/pytest.log; then exit 1; fi
bash tools/profile_round.sh r3_final 12288 f32x3 2>&1 | tail -3
