#!/bin/bash
# round 3, third GPU pass: the GPU suite on the in-tree build
mkdir -p gpurun_out/r3c
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r3c/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r3c/pytest.log
tail -8 gpurun_out/r3c/pytest.log
