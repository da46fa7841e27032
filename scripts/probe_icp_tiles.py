#!/usr/bin/env python3
"""Counters of the ICP loop's tile search on the configs[2] batch (needs a -DSCL_DIAGNOSTICS build of icp.hip:
scripts/build_variant.sh diag "-DSCL_DIAGNOSTICS=1" icp.hip (=2: phase stamps); SCL_ENGINE_LIB=scl_slam_amd/lib/variants/libscl_engine_diag.so)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
n_cand, n_pts = int(os.environ.get("NC", "25")), 100000
eng = ScanContextEngine(num_ring=64, num_sector=120)
tgts = [synth_structured_cloud(n_pts, seed=100 + c, extent=60.0) for c in range(n_cand)]
T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
rs = np.random.RandomState(3); src0 = tgts[0].copy()
p = tgts[0][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
src0[:, :3] = (p + 0.01 * rs.standard_normal(p.shape)).astype(np.float32)
lib = eng._lib
stats = (ctypes.c_ulonglong * 16)()
names = ["rounds", "rounds_overflowed", "lanes_in_memory", "table_entries", "points_staged", "row_visits", "points_compared", "tiles_flagged",
         "t_until_ball", "t_box", "t_table", "t_rowscan", "t_staging", "t_walk", "t_reduce", "-"]
for est in ((0,) if os.environ.get("PROBE_P2P_ONLY") else (0, 1)):   # PROBE_P2P_ONLY: scripts/profile_icp.sh
    pp = eng.icp_default_params(); pp.max_iterations = 30; pp.estimator = est; pp.normal_radius = 1.0
    eng.icp_align_batch(src0, tgts, pp)
    lib.scl_debug_icp_tile_stats(stats, 1)
    t0 = time.perf_counter()
    Tb, fb, cb, ib = eng.icp_align_batch(src0, tgts, pp)
    dt = time.perf_counter() - t0
    lib.scl_debug_icp_tile_stats(stats, 1)
    d = dict(zip(names, [int(v) for v in stats]))
    r = max(1, d["rounds"] - d["rounds_overflowed"])
    print(f"estimator {est}: {dt * 1e3:.2f} ms, iterations {ib.tolist()}")
    print("   ", d)
    wgs = max(1, sum(ib) + n_cand) * ((n_pts + 255) // 256)     # workgroups that did work: (iterations + fitness pass) x tiles, roughly
    print("    us per workgroup (thread 0): " + ", ".join(f"{k[2:]} {d[k] * 0.01 / wgs:.2f}" for k in names if k.startswith("t_")))
    print(f"    per served round: {d['table_entries'] / r:.0f} table entries, {d['points_staged'] / r:.0f} points; "
          f"per query-round: {d['row_visits'] / (r * 256):.2f} rows, {d['points_compared'] / (r * 256):.1f} points")
eng.close()
