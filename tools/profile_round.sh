#!/bin/bash
# One round of profiling evidence on the GPU box: tools/profile_round.sh <tag> [batch] [mlp_dtype] [math_mode]
# kernel trace of bench.py (no single-instance latency launches: they run the same kernel and would mix into its average) + separate PMC passes (HBM bytes: FETCH_SIZE / WRITE_SIZE in their own passes, MI355X_MICROARCH.md) +
# counter calibration on the rollout / gradient kernels whose traffic is known. Summarise with tools/summarize_profile.py.
tag=${1:-r04}; B=${2:-12288}; MLP=${3:-f32x3}; MATH=${4:-fast}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
sha256sum sde4mbrl_px4_amd/csrc/libsdempc.so > $out/lib_sha.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --mlp-dtype $MLP --math-mode $MATH --no-cpu-baseline --no-other-math-mode --no-other-configs --verify 0 --latency-reps 0 > $out/bench_trace.log 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc_$i --output-format csv -- python3 tools/prof_solve.py --mode solve --batch $B --reps 1 --mlp-dtype $MLP --math-mode $MATH > $out/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
  echo "pmc pass $i done"
done
for mode in rollout grad; do
  j=0
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    j=$((j+1))
    timeout -k 10 200 rocprofv3 --pmc $c -d $out/cal_${mode}_$j --output-format csv -- python3 tools/prof_solve.py --mode $mode --batch $B --reps 1 --mlp-dtype $MLP --math-mode $MATH > $out/cal_${mode}_$j.log 2>&1 || echo "cal $mode $j failed"
  done
done
find $out -name "*.db" -delete 2>/dev/null
du -sh $out
