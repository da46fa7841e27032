#!/usr/bin/env python3
"""BASELINE configs[4]-shaped streaming run on ONE GPU: dense Livox-like scans (~240 k points), 80x180 Scan
Context, database growing from 1 000 keyframes, per scan: voxel filter -> descriptor -> append -> reference-faithful
detection (top-k + SC distance) -> ICP verification of the detected loop (point-to-point, <= 30 iterations).
Reports end-to-end latency per scan (ms) and the number of scans that would miss a 10 Hz budget."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import synth_descriptors, synth_scan, synth_structured_cloud, rigid_transform  # noqa: E402

R, S = 80, 180
n0, n_scans = 1000, int(os.environ.get("SCL_STREAM_SCANS", "200"))
eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=10, initial_capacity=2048)
eng.save_bulk(synth_descriptors(n0, R, S, seed=1005))
scans = [synth_scan(240000, seed=100 + i) for i in range(8)]              # reused round robin (host RAM)
submap = synth_structured_cloud(100000, seed=7, extent=60.0)
Tinv = np.linalg.inv(rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05))
src_moved = submap[::2].copy()
src_moved[:, :3] = (submap[::2, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
p = eng.icp_default_params(); p.max_iterations = 30
lat, parts = [], {"(merged)": [], "voxel+descriptor+append": [], "detect": [], "icp": []}
for i in range(n_scans):
    scan = scans[i % len(scans)]
    t0 = time.perf_counter()
    t1 = t0
    eng.make_and_save_filtered(scan, 0.4, 0, n0 + i)                       # makeDescriptors, DM.h:996-1002 (descriptLeafSize, DM.h:185)
    t2 = time.perf_counter()
    lid, shift, dist = eng.detect_intra(n0 + i)
    t3 = time.perf_counter()
    eng.icp_align(src_moved, submap, p)                                    # ICP problem of the verification stage (drifted pose)
    t4 = time.perf_counter()
    lat.append((t4 - t0) * 1e3)
    for k, v in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        parts[k].append(v * 1e3)
lat = np.array(lat[5:])
out = {"scans": int(lat.size), "points_per_scan": 240000, "grid": f"{R}x{S}", "db_keyframes": [n0, n0 + n_scans],
       "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
       "stage_ms_p50": {k: float(np.percentile(v[5:], 50)) for k, v in parts.items()},
       "missed_10hz_budget": int((lat > 100.0).sum())}
print(json.dumps(out, indent=1))
