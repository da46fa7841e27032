#!/bin/bash
# A/B of engine libraries on the configs[2] batch from the keyframe store, on ONE box: scripts/ab_icp.sh NAME [NAME ...]
# (NAME = prod | a variant of scripts/build_variant.sh), each traced by scripts/trace_icp.sh, twice, interleaved.  Prints, per run, the
# point-to-point phase's span and its three loop kernels; the timelines stay under gpurun_out/ab_icp/.
mkdir -p gpurun_out/ab_icp
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = prod ]; then unset SCL_ENGINE_LIB; else export SCL_ENGINE_LIB=$PWD/scl_slam_amd/lib/variants/libscl_engine_$v.so; fi
    bash scripts/trace_icp.sh > gpurun_out/ab_icp/run_${v}_$rep.log 2>&1 || exit 1
    cp gpurun_out/trace_icp/timeline.txt gpurun_out/ab_icp/timeline_${v}_$rep.txt
    echo "== $v rep $rep: $(grep 'estimator 0 rep 1' gpurun_out/ab_icp/run_${v}_$rep.log | sed 's/, points.*//') | $(grep 'estimator 1 rep 1' gpurun_out/ab_icp/run_${v}_$rep.log | sed 's/, iterations.*//')"
    python3 - gpurun_out/ab_icp/timeline_${v}_$rep.txt <<'PY'
import sys, re
txt = open(sys.argv[1]).read().split("--- phase ")
for ph in txt[-1:]:
    head = ph.splitlines()[0]
    ks = {m.group(1): (int(m.group(2)), float(m.group(3))) for m in re.finditer(r"^\s+(\S+)\s+calls\s+(\d+) total\s+([\d.]+) us", ph, re.M)}
    tl = [l for l in ph.splitlines() if l.startswith("      ") and ":" in l][-1].split()
    warm = [int(x.split(":")[1]) for x in tl[1:-1] if int(x.split(":")[1]) > 100]
    print("   p2p:", head.strip(), "| search", ks.get("icp_tile_search_kernel"), "finish", ks.get("icp_tile_finish_kernel"), "solve", ks.get("icp_solve_batch_kernel"),
          "| warm searches mean %.1f us over %d" % (sum(warm) / max(len(warm), 1), len(warm)))
PY
  done
done
