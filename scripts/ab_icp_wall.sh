#!/bin/bash
# Untraced wall times of the configs[2] batch from the keyframe store for engine variants (scripts/build_variant.sh) and numbers of HIP
# hardware queues: scripts/ab_icp_wall.sh "4 8" prod p1 p2   ->   one line per (queues, variant): point to plane / point to point, ms
QS=$1; shift
for q in $QS; do
  for v in "$@"; do
    if [ "$v" = prod ]; then unset SCL_ENGINE_LIB; else export SCL_ENGINE_LIB=$PWD/scl_slam_amd/lib/variants/libscl_engine_$v.so; fi
    GPU_MAX_HW_QUEUES=$q LEAF=0.05 python3 scripts/trace_icp_batch.py run 2>/dev/null | awk -v q=$q -v v=$v '/estimator 1 rep 1/{a=$5} /estimator 0 rep 1/{b=$5} END{printf "queues %s %-6s p2plane %s ms  p2p %s ms\n", q, v, a, b}'
  done
done
