#!/bin/bash
# PMC A/B of the solve kernel under two option environments (run on the GPU box): tools/pmc_ab.sh <tag> <batch> "<envA>" "<envB>" [config]
tag=${1:-ab}; B=${2:-3072}; EA=${3:-SDEMPC_DUO=1}; EB=${4:-SDEMPC_DUO=0}; CFG=${5:-configs/c2_iris_traj_h50_p128.yaml}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
n=0
for E in "$EA" "$EB"; do
  n=$((n+1)); i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES"; do
    i=$((i+1))
    export $E
    timeout -k 10 200 rocprofv3 --pmc $c -d $out/v${n}_$i --output-format csv -- python3 tools/prof_solve.py --config $CFG --mode solve --batch $B --reps 1 > $out/v${n}_$i.log 2>&1 || echo "pass v$n $i failed"
    unset ${E%%=*}
  done
done
python3 - "$out" "$EA" "$EB" <<'PY'
import csv, glob, sys
out, names = sys.argv[1], sys.argv[2:]
for n, name in enumerate(names, 1):
    acc = {}
    for f in glob.glob(f"{out}/v{n}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "solve" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                acc["_kernel"] = r["Kernel_Name"][:90]
    print(name, acc.pop("_kernel", "?"))
    wc = acc.get("SQ_WAVE_CYCLES", 1.0)
    for k, v in sorted(acc.items()):
        print(f"   {k:28s} {v:14.5g}  /wave_cycles {v / wc:8.4f}")
PY
find $out -name "*.db" -delete 2>/dev/null
