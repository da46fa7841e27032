#!/usr/bin/env python3
"""scl_stream_from_points from a pinned arena at several stream lengths: the per-scan time's asymptote against the link's floor."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors, synth_scan
R, S, n, npts = 64, 120, 10000, 120000
eng = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 4096)
eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
bases = [synth_scan(npts, seed=500 + i, stride_floats=4) for i in range(16)]
N = 1024
arena = eng.host_alloc((N, npts, 4))
for i in range(N):
    arena[i] = bench._distinct_scan(bases[i % 16], i)
eng.stream_from_points([arena[i] for i in range(32)])
h2d = eng.host_copy_rate(64 << 20, 8); h30 = eng.host_copy_rate(16 * npts * 16, 8)
print(f"H2D: 64 MB copies {h2d:.1f} GB/s, group-sized (30.7 MB) copies {h30:.1f} GB/s; floor {npts * 16 / h2d / 1e3:.1f} us per scan")
at = 32
for m in (64, 128, 256, 512):
    t0 = time.perf_counter(); eng.stream_from_points([arena[i] for i in range(at, at + m)]); dt = time.perf_counter() - t0
    at = (at + m) % (N - 512)
    print(f"{m} scans: {dt / m * 1e6:.2f} us per scan, {dt * 1e3:.2f} ms")
eng.close()
