cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# The clock the chip holds under the screening kernels: GRBM_GUI_ACTIVE (graphics-clock cycles, summed over the XCDs) against the
# dispatch's duration from the same counter-collection record.
OUT=gpurun_out/pmc_clock
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o pmc -- python3 bench.py --steps 16 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/p1.json 2> $OUT/p1.err
python3 - <<'P'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/pmc_clock/**/*_counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("<")[0].split("(")[0][-40:]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        agg[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, v in agg.items():
    if "dur_ns" in v and len(v["dur_ns"]) > 3:
        d = sum(v["dur_ns"]) / len(v["dur_ns"]); g = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])
        w = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"]); b = sum(v["SQ_BUSY_CYCLES"]) / len(v["SQ_BUSY_CYCLES"])
        print(f"{k:42s} n={len(v['dur_ns']):4d} dur {d/1e3:8.1f} us  GRBM/8/dur = {g/8/d:6.3f} GHz   SQ_BUSY/dur {b/d:8.2f}  WAVE_CYCLES/dur {w/d:9.1f}")
P
rm -rf $OUT/p1
