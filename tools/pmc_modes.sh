#!/bin/bash
# PMC stall breakdown of the standalone rollout / grad kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_modes; rm -rf $out; mkdir -p $out
for mode in rollout grad; do
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $c -d $out/${mode}_$i --output-format csv -- python3 tools/prof_solve.py --mode $mode --batch 2048 --reps 1 > $out/${mode}_$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for mode in ("rollout", "grad"):
    acc = {}
    for f in glob.glob(f"gpurun_out/pmc_modes/{mode}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if mode in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = float(r["Counter_Value"])
    wc = acc.get("SQ_WAVE_CYCLES", 1)
    print(mode, {k: (f"{v:.4g}", f"{v/wc:.3f}") for k, v in sorted(acc.items())})
PY
