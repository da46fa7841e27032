#!/bin/bash
# A/B of engine builds on the exact distance matrix: scripts/ab_matrix.sh name1 name2 ...  ("-" = the product library); two rounds, interleaved
cd "$GRAFT_REPO_ROOT"
GRID=${GRID:-64x120}
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset SCL_ENGINE_LIB; else export SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_$v.so; fi
    timeout -k 10 200 python3 scripts/bench_matrix.py 64 $GRID > gpurun_out/abm_$v.$round.json 2> gpurun_out/abm_$v.$round.err
    python3 - "$v" "$round" "$GRID" <<'P'
import json,sys
v,r,g=sys.argv[1:4]
try:
    d=json.load(open(f"gpurun_out/abm_{v}.{r}.json"))[g]
    print(f"{v:12s} round {r}: {d['pairs_per_s']/1e6:.1f} M pairs/s  {d['ms_per_row']*1e3:.1f} us per row   group of rows on the device {d['group_us']:.0f} us = {d['pairs_per_s_on_the_device']/1e6:.1f} M pairs/s")
except Exception as ex:
    print(v, r, "failed:", ex)
P
  done
done
