#!/bin/bash
# A/B of engine builds on one box: scripts/ab_variants.sh name1 name2 ...   ("-" = the product library); two rounds, interleaved.
# Prints the headline value, the screening group's HIP-event time and the 80x180 secondary for each.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SCL_ENGINE_LIB; else export SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --only-secondary sc_distance_80x180 > gpurun_out/ab_$v.$round.json 2> gpurun_out/ab_$v.$round.err
    python3 - "$v" "$round" <<'P'
import json,sys
v,r=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(f"gpurun_out/ab_{v}.{r}.json").read().strip().splitlines()[-1])
    s=d.get("secondary",{}).get("sc_distance_80x180",{})
    print(f"{v:10s} round {r}: {d['value']/1e9:.3f} G pairs/s  group {d['kernel_ms']['sc_distance']*1e3:.1f} us  frac {d['roofline']['frac']:.3f}   80x180 {s.get('value',0)/1e6:.0f} M  group {s.get('kernel_ms',{}).get('screening_launch_group',0)*1e3:.1f} us")
except Exception as ex:
    print(v, r, "failed:", ex)
P
  done
done
