#!/bin/bash
# round 3: the default bench line of the final library with the driver's flags (kept as profiles/r3_c2_bench.json)
mkdir -p gpurun_out/r3o
t0=$(date +%s)
timeout -k 10 1000 python bench.py --steps 20 --warmup 5 > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err
echo "bench exit $? wall $(( $(date +%s) - t0 )) s" | tee gpurun_out/r3o/bench_wall.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3o/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_solve_latency_ms", "verified_bit_exact", "verified_instances", "library_build")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("kernel"))
print({k: (v["value"], v.get("verified_bit_exact")) for k, v in d["other_configs"].items()})
PY
