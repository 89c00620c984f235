#!/bin/bash
# rocprofv3 evidence for BASELINE config C5 (H=200, P=1024) in f32 and with fp16 MLP operands: tools/profile_c5.sh <tag> [batch] [max_iter]
# kernel trace of bench.py on the C5 config + HBM byte counters + MFMA / VALU counters on tools/prof_solve.py (separate --pmc passes).
tag=${1:-r2c5}; B=${2:-768}; IT=${3:-20}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*" | sort -u > $out/mfma_counters.txt
for dt in f32 f16; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/trace_$dt --output-format csv -- python3 bench.py --config configs/c5_iris_traj_h200_p1024.yaml --mlp-dtype $dt \
      --batch $B --max-iter $IT --steps 2 --warmup 1 --no-cpu-baseline --latency-reps 0 --verify 0 > $out/bench_$dt.log 2>&1 || echo "trace $dt failed"
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc_${dt}_$i --output-format csv -- python3 tools/prof_solve.py --config configs/c5_iris_traj_h200_p1024.yaml --mlp-dtype $dt --mode solve --batch $B --max-iter $IT --reps 1 > $out/pmc_${dt}_$i.log 2>&1 || echo "pmc $dt pass $i failed"
  done
done
find $out -name "*.db" -delete 2>/dev/null
python3 - "$out" "$B" <<'PY'
import csv, glob, json, sys
out, B = sys.argv[1], int(sys.argv[2])
res = {}
for dt in ("f32", "f16"):
    acc = {}
    for f in glob.glob(f"{out}/pmc_{dt}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "solve" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); acc["kernel"] = r["Kernel_Name"]
    bench = None
    for line in open(f"{out}/bench_{dt}.log"):
        if line.startswith("{"): bench = json.loads(line)
    res[dt] = {"counters_per_launch": acc, "bench_under_rocprof": bench}
    if "FETCH_SIZE" in acc: res[dt]["hbm_bytes_per_solve"] = (2.0 * acc["FETCH_SIZE"] + acc["WRITE_SIZE"]) * 1024.0 / B
json.dump(res, open(f"{out}/c5_summary.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "bench_under_rocprof"} for k, v in res.items()}, indent=1)[:3000])
PY
