#!/bin/bash
# Counters of icp_tile_search_kernel on the configs[2] batch (separate --pmc passes, no tracing domains beside them).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icp_tile
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  PROBE_P2P_ONLY=1 timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 scripts/probe_icp_tiles.py > $OUT/p$i.log 2> $OUT/p$i.err || echo "pmc group $i failed: $grp" >> $OUT/progress.txt
  echo "pmc $i done" >> $OUT/progress.txt
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_icp_tile_search.json icp_tile_search > /dev/null
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
