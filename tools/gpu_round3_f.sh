#!/bin/bash
# round 3: GPU suite + the default bench line (wall time recorded) on the final tree
mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r3f/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r3f/pytest.log
tail -4 gpurun_out/r3f/pytest.log
t0=$(date +%s)
timeout -k 10 800 python bench.py --steps 5 --warmup 1 > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err
echo "bench exit $? wall $(( $(date +%s) - t0 )) s" | tee gpurun_out/r3f/bench_wall.txt
tail -c 1500 gpurun_out/r3f/bench.json
tail -5 gpurun_out/r3f/bench.err
