#!/bin/bash
# Kernel trace + MFMA counters of the configs[2] geometric-verification batch (25 candidates x 100 k points).
# Run on the GPU box through gpurun; outputs under gpurun_out/prof_icp/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_icp
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o icp -- python3 scripts/bench_icp25.py > $OUT/icp25_under_trace.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/icp_kernel_stats.csv
python3 scripts/summarize_kernel_stats.py $OUT/icp_kernel_stats.csv > $OUT/icp_kernel_stats_short.txt
i=0
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "FETCH_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  NC=6 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 scripts/bench_icp25.py > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp" >> $OUT/progress.txt
  echo "pmc $i done" >> $OUT/progress.txt
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_corr_reduce_mfma.json corr_reduce_mfma > /dev/null
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_nn_search.json nn_search > /dev/null
rm -rf $OUT/trace $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
python3 scripts/bench_icp25.py > $OUT/icp25_bench.json 2> $OUT/bench.err
echo "all done" >> $OUT/progress.txt
