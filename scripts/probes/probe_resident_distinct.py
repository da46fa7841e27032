#!/usr/bin/env python3
"""scl_stream_from_store on distinct scans (bench.py's secondary.stream_from_resident_points, alone): us per scan, pairs/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
for n_scans in (512, 2048):
    r = bench.secondary_stream_from_resident_points(0, n_scans=n_scans)
    print(n_scans, "scans:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("value", "us_per_scan", "scans_per_s")})
