#!/bin/bash
# same-box A/B of the throughput kernel: in-tree library vs build/libsdempc_ta.so (touch one step ahead in the adjoint)
cd $GRAFT_REPO_ROOT; out=gpurun_out/ab_ta; mkdir -p $out
for r in 1 2; do
  for v in intree ta; do
    if [ $v = intree ]; then unset SDEMPC_LIB; else export SDEMPC_LIB=build/libsdempc_ta.so; fi
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-math-mode --no-other-configs --verify 2 --latency-reps 0 > $out/${v}_$r.json 2> $out/${v}_$r.err
    python3 -c "import json,sys; r=json.loads(open('$out/${v}_$r.json').read().strip().split('\n')[-1]); print('$v $r', round(r['value'],1), r.get('verified_bit_exact'), r['roofline'].get('kernel_ms'))"
  done
done
