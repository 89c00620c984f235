#!/bin/bash
# the whole default bench under rocprofv3 (every leg's kernel and average duration), the arithmetic drift tool with its float64 column, a bench run on 8 CPU threads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_extras; rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/trace_all --output-format csv -- python3 bench.py --steps 5 --warmup 1 > $out/bench_all.json 2> $out/bench_all.err
echo "bench under rocprofv3 rc=$?"
timeout -k 10 400 python tools/mode_drift.py --batch 48 > $out/mode_drift.txt 2>&1; echo "mode_drift rc=$?"; tail -8 $out/mode_drift.txt
timeout -k 10 500 python bench.py --cpu-threads 8 > $out/bench_8thr.json 2> $out/bench_8thr.err; echo "bench 8 threads rc=$?"; tail -2 $out/bench_8thr.err
find $out -name "*.db" -delete 2>/dev/null
