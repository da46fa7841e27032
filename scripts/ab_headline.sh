#!/bin/bash
# A/B of variant libraries on the headline stream (64x120, 10 k keyframes): scripts/ab_headline.sh NAME... (product library first)
cd "$GRAFT_REPO_ROOT"
python3 - "$@" <<'PY'
import json, os, subprocess, sys
names = ["product"] + sys.argv[1:]
for rep in range(2):
    for n in names:
        env = dict(os.environ)
        if n != "product": env["SCL_ENGINE_LIB"] = os.path.join("scl_slam_amd/lib/variants", f"libscl_engine_{n}.so")
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-secondary"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            print(f"{n:12s} {d['value']/1e9:7.3f} G pairs/s  ms/step {d['ms_per_step']*1e3:6.1f} us  group {d['kernel_ms']['sc_distance']*1e3:6.1f} us  frac {d['roofline']['frac']:.3f}", flush=True)
        except Exception as ex:
            print(n, "failed", ex, out.stderr[-400:], flush=True)
PY
