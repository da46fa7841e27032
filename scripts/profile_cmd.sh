#!/bin/bash
# Kernel trace + counter passes of ONE command (run on the GPU box through gpurun).
#   usage: scripts/profile_cmd.sh <name> <kernel substring[,substring...]> <python script> [args...]
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel durations        (gpurun_out/prof_<name>/kernel_stats.csv)
#   2. rocprofv3 --pmc <one counter group each>  -> FETCH_SIZE / WRITE_SIZE (HBM bytes), L2 hits, SQ wait / VALU / LDS / MFMA
#                                                   condensed by profiles/summarize_pmc.py (pmc_summary.json)
# No tracing domain is combined with --pmc; the program after `--` is python3 itself.
NAME=$1; FILTER=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$NAME
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 "$@" > $OUT/under_trace.json 2> $OUT/trace.err || { echo "trace failed" >> $OUT/progress.txt; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 scripts/summarize_kernel_stats.py $OUT/kernel_stats.csv > $OUT/kernel_stats_short.txt
rm -rf $OUT/trace
echo "trace done" >> $OUT/progress.txt
if [ "${PMC:-1}" = "0" ]; then exit 0; fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 "$@" > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp" >> $OUT/progress.txt
  echo "pmc $i done" >> $OUT/progress.txt
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_summary.json $FILTER > /dev/null
for j in $(seq 1 $i); do rm -rf $OUT/p$j; done
python3 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "all done" >> $OUT/progress.txt
