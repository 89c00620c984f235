#!/bin/bash
# round 3: kernel trace of the bench's timed configuration alone (for profiles/), then a randomised soak over the three contraction modes
mkdir -p gpurun_out/r3h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sha256sum sde4mbrl_px4_amd/csrc/libsdempc.so > gpurun_out/r3h/lib_sha.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3h/trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-modes --no-other-configs --verify 0 --latency-reps 0 > gpurun_out/r3h/bench_trace.log 2>&1
find gpurun_out/r3h -name "*.db" -delete 2>/dev/null
tail -c 600 gpurun_out/r3h/bench_trace.log
timeout -k 10 700 python tests/tools/soak.py 500 7000 > gpurun_out/r3h/soak.log 2>&1
echo "soak exit $?"; tail -3 gpurun_out/r3h/soak.log
