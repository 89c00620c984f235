#!/bin/bash
# Interleaved A/B of builds on the GPU box, full C2 solves (or another config):
#   tools/ab_solve.sh "<prof_solve.py args>" lib1.so lib2.so ...     ("-" = the in-tree build)
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset SDEMPC_LIB; else export SDEMPC_LIB=$lib; fi
    echo "== $lib (round $rep)"
    timeout -k 10 300 python tools/prof_solve.py --mode solve $args 2>&1 | grep -v amdgpu | tail -2
  done
done
