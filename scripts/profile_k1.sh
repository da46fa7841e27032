#!/bin/bash
# Profiles of the headline bench command (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats           -> per-kernel durations of `python3 bench.py`
#   2. rocprofv3 --pmc <one counter group each>   -> HBM bytes (FETCH_SIZE, WRITE_SIZE), VALU / LDS activity
# No tracing domain is combined with --pmc.  Outputs under gpurun_out/prof_k1/, summaries copied by the caller.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_k1
rm -rf $OUT && mkdir -p $OUT
STEPS=${STEPS:-128}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k1 -- python3 bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace done" >> $OUT/progress.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 16 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp" >> $OUT/progress.txt
  echo "pmc $i done" >> $OUT/progress.txt
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_summary.json ${K1_FILTER:-sc_screen2_kernel,sc_screen2_tail2_kernel} > /dev/null
python3 scripts/summarize_kernel_stats.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) > $OUT/kernel_stats_short.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 bench.py --steps $STEPS --warmup 2 --no-secondary > $OUT/bench.json 2> $OUT/bench.err
echo "all done" >> $OUT/progress.txt
