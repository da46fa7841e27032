#!/bin/bash
# points per scatter workgroup (kScPointsPerWorkgroup): device time of the batch scatter at several values, diagnostics build of engine.hip
#   scripts/build_variant.sh diag "-DSCL_DIAGNOSTICS" engine.hip && gpurun -- scripts/sweep_sc_slice.sh
for s in ${SLICES:-1024 2048 4096 8192 16384}; do
  echo "slice $s: $(SCL_SC_SLICE=$s SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_diag.so python3 scripts/bench_front.py ingest 2>/dev/null | python3 -c 'import json,sys; j=json.load(sys.stdin)["ingest_per_scan"]; print({k: v["device_us_per_batch"] for k, v in j.items()})')"
done
