#!/bin/bash
# A/B of variant libraries on the 80x180 full-database stream: scripts/ab_wide.sh NAME... (product library first)
cd "$GRAFT_REPO_ROOT"
python3 - "$@" <<'PY'
import json, os, subprocess, sys
names = ["product"] + sys.argv[1:]
for rep in range(2):
    for n in names:
        env = dict(os.environ)
        if n != "product": env["SCL_ENGINE_LIB"] = os.path.join("scl_slam_amd/lib/variants", f"libscl_engine_{n}.so")
        out = subprocess.run([sys.executable, "scripts/bench_80x180.py"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            print(f"{n:12s} {d['value']/1e6:8.1f} M pairs/s  group {d['kernel_ms']['screening_launch_group']*1e3:7.1f} us  scans/launch {d['roofline']['scans_per_launch']:.1f}", flush=True)
        except Exception as ex:
            print(n, "failed", ex, out.stderr[-400:], flush=True)
PY
