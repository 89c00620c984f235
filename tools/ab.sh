#!/bin/bash
# A/B kernel variants on the GPU box: tools/ab.sh lib1.so lib2.so ...   (interleaved, same process settings)
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib (rep $rep)"
  SDEMPC_LIB=$lib timeout 120 python tools/prof_solve.py --mode rollout --batch 2048 --reps 3 2>&1 | grep -v amdgpu | tail -1
  SDEMPC_LIB=$lib timeout 120 python tools/prof_solve.py --mode grad --batch 2048 --reps 3 2>&1 | grep -v amdgpu | tail -1
  SDEMPC_LIB=$lib timeout 200 python tools/prof_solve.py --mode solve --batch 2048 --max-iter 20 --reps 2 2>&1 | grep -v amdgpu | tail -1
done; done
