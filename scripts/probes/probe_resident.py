#!/usr/bin/env python3
"""Where scl_stream_from_store's time goes: the whole call against the detection of the same keyframes on its own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors, synth_scan
R, S, n, npts = 64, 120, 10000, 120000
for n_scans in (256, 1024):
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 3 * n_scans + 128)
    eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
    clouds = [np.ascontiguousarray(synth_scan(npts, seed=700 + i, stride_floats=4)) for i in range(16)]
    for i in range(2 * n_scans):
        eng.keyframe_put(0, i, clouds[i % 16])
    eng.stream_from_store(0, 0, n_scans)
    n_before = eng.get_size()
    t0 = time.perf_counter(); eng.stream_from_store(0, n_scans, n_scans); t1 = time.perf_counter()
    q = np.arange(n_before, n_before + n_scans, dtype=np.int32)
    hi = (q - 100).astype(np.int32)
    eng.detect_full_stream(q, 0, hi, 16, 2)
    t2 = time.perf_counter(); eng.detect_full_stream(q, 0, hi, 16, 2); t3 = time.perf_counter()
    hi2 = np.full(n_scans, n_before - 100, dtype=np.int32)
    eng.detect_full_stream(q, 0, hi2, 16, 2)
    t4 = time.perf_counter(); eng.detect_full_stream(q, 0, hi2, 16, 2); t5 = time.perf_counter()
    print(f"{n_scans} scans: whole call {1e6 * (t1 - t0) / n_scans:.2f} us/scan; detection alone, ranges [0, key - 100): {1e6 * (t3 - t2) / n_scans:.2f}; one common range: {1e6 * (t5 - t4) / n_scans:.2f}")
    eng.close()
