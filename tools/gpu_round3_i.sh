#!/bin/bash
# round 3: the new non-finite parity tests, then the default bench line of the final library (kept as profiles/r3_c2_bench.json)
mkdir -p gpurun_out/r3i
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "non_finite" > gpurun_out/r3i/pytest.log 2>&1
echo "pytest exit $?"; tail -3 gpurun_out/r3i/pytest.log
t0=$(date +%s)
timeout -k 10 900 python bench.py --steps 20 --warmup 2 > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
echo "bench exit $? wall $(( $(date +%s) - t0 )) s" | tee gpurun_out/r3i/bench_wall.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_solve_latency_ms", "verified_bit_exact", "verified_instances", "library_build")}, d["roofline"]["frac"], d["roofline"]["traffic"])
print({k: (v["value"], v.get("verified_bit_exact")) for k, v in d["other_configs"].items()})
PY
