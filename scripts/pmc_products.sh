cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# SQ / GRBM counters of the products kernel (sc_screen2_kernel) of the headline command: issue, wait and unit-busy cycles, and the
# clock the chip holds under it (GRBM_GUI_ACTIVE over the kernel's duration).  One counter group per pass, no tracing domain.
OUT=gpurun_out/pmc4
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 16 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/p$i.json 2> $OUT/p$i.err || echo "pmc group $i failed: $grp"
done
python3 profiles/summarize_pmc.py $OUT $OUT/pmc_summary.json "${PMC_KERNEL:-sc_screen2_kernel}" > /dev/null
python3 - <<'P'
import json
d=json.load(open("gpurun_out/pmc4/pmc_summary.json"))
for k,v in d["counters"].items(): print(f"{k:38s} {v['mean_per_launch']:16.0f}  ({v['launches']} launches)")
P
rm -rf $OUT/p?
