#!/usr/bin/env python3
"""bench.py's configs[2] measurement (25 candidates x 100 k points, both estimators, from the store and from host buffers) on
its own, for a kernel trace: run under rocprofv3 --kernel-trace --stats to see where the device time goes."""
import json, sys
sys.path.insert(0, ".")
import bench
from scl_slam_amd import ScanContextEngine

eng = ScanContextEngine(num_ring=64, num_sector=120)
r = bench.secondary_icp(eng)
for est in ("point_to_plane", "point_to_point"):
    for mode in ("from_store", "host_buffers"):
        x = r[est][mode]
        print(est, mode, f"{x['ms_per_query']:.2f} ms per query, {x['ms_per_candidate']:.3f} per candidate, {x['iterations_mean']:.1f} iterations")
