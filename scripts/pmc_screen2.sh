#!/bin/bash
# PMC passes of the second-form screening kernel (one counter group per run, no tracing domain); outputs gpurun_out/pmc2/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc2
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- ${PMC_CMD:-python3 bench.py --steps 16 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary} > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp"
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_summary.json ${1:-sc_screen2_kernel} > /dev/null
python3 - <<'P'
import json
d=json.load(open("gpurun_out/pmc2/pmc_summary.json"))
for k,v in d["counters"].items(): print(f"{k:34s} {v['mean_per_launch']:16.0f}  ({v['launches']} launches)")
P
rm -rf $OUT/p?
