cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc3
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" "SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_INSTS_FLAT SQ_INSTS_GDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  SCL_SCREEN_TAIL=0 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 16 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp"
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_summary.json "${PMC_KERNEL:-sc_align2_kernel}" > /dev/null
python3 - <<'P'
import json
d=json.load(open("gpurun_out/pmc3/pmc_summary.json"))
for k,v in d["counters"].items(): print(f"{k:34s} {v['mean_per_launch']:16.0f}  ({v['launches']} launches)")
P
rm -rf $OUT/p?
