#!/bin/bash
# Kernel trace + counters of the configs[2] geometric-verification batch (25 candidates x 100 k points).
# Run on the GPU box through gpurun; outputs under gpurun_out/prof_icp/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/prof_icp
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o icp -- python3 scripts/bench_icp25.py > $OUT/icp25_under_trace.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/icp_kernel_stats.csv
python3 scripts/summarize_kernel_stats.py $OUT/icp_kernel_stats.csv > $OUT/icp_kernel_stats_short.txt
rm -rf $OUT/trace
# the from-store query phase by phase (scripts/trace_icp_batch.py)
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2 -o t -- python3 scripts/trace_icp_batch.py run > $OUT/from_store_under_trace.log 2> $OUT/trace2.err
python3 scripts/trace_icp_batch.py $(find $OUT/trace2 -name "*kernel_trace.csv" | head -1) > $OUT/timeline.txt
rm -rf $OUT/trace2
echo "traces done" >> $OUT/progress.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  PROBE_P2P_ONLY=1 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 scripts/probe_icp_tiles.py > $OUT/p$i.log 2> $OUT/p$i.err || echo "pmc group $i failed: $grp" >> $OUT/progress.txt
  echo "pmc $i done" >> $OUT/progress.txt
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_icp_tile_search.json icp_tile_search > /dev/null
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 $OUT/p7
python3 scripts/bench_icp25.py > $OUT/icp25_bench.json 2> $OUT/bench.err
python3 scripts/bench_icp_prep.py > $OUT/icp_prep_and_loop.json 2>> $OUT/bench.err
echo "all done" >> $OUT/progress.txt
