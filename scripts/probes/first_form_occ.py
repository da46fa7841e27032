#!/usr/bin/env python3
"""Blocking calls of one to three scans (the first form of the screening products, 64x120, 10 k keyframes): wall clock per call.
   In a diagnostics build SCL_SCREEN_VARIANT=1 selects the kernel built for two waves per SIMD instead of three."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
R, S, N = 64, 120, 10000
eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 8)
eng.save_bulk(synth_descriptors(N, R, S, seed=1002))
for nq in (1, 2, 3, 4):
    ts = []
    for i in range(80):
        q = np.arange(N - 1 - (i % 40), N - 1 - (i % 40) - nq, -1, dtype=np.int32)
        t0 = time.perf_counter()
        if nq == 1: eng.detect_full_range(int(q[0]), 0, N - 100)
        else: eng.detect_full_stream(q, 0, N - 100, nq, 2)
        ts.append((time.perf_counter() - t0) * 1e6)
        time.sleep(0.0003)
    print(f"nq={nq}: p50 {np.percentile(ts[10:], 50):.1f} us  min {min(ts[10:]):.1f}")
eng.close()
