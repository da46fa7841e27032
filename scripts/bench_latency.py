#!/usr/bin/env python3
"""Latency of the reference-faithful detection (descriptor.h:1613-1674: ring-key top-k, k SC distances,
threshold) through the six-virtuals entry point, 10 k keyframes 64x120 (BASELINE configs[1] database),
and of the full-DB pass for comparison.  Secondary measurement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import synth_descriptors  # noqa: E402

R, S, N = 64, 120, 10000
eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 8)
eng.save_bulk(synth_descriptors(N, R, S, seed=1002))


def pct(fn, reps=400):
    for _ in range(20):
        fn(0)
    ts = []
    for i in range(reps):
        t0 = time.perf_counter(); fn(i); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return {"p50_us": float(np.percentile(ts, 50)), "p99_us": float(np.percentile(ts, 99)), "queries_per_s": float(1e6 / ts.mean())}


out = {"database": f"{N} keyframes {R}x{S}",
       "detect_intra_k3": pct(lambda i: eng.detect_intra(N - 1 - (i % 50))),
       "detect_inter_k3": pct(lambda i: eng.detect_inter(N - 1 - (i % 50))),
       "detect_full_blocking": pct(lambda i: eng.detect_full_range(N - 1 - (i % 50), 0, N - 100), reps=200)}
print(json.dumps(out, indent=1))
eng.close()
