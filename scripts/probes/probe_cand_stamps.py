#!/usr/bin/env python3
"""Phase stamps of sc_cand_exact_kernel (diagnostics build, SCL_INGEST_STAMPS=1): the reference-faithful detection at 10 k keyframes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
R, S, n = 64, 120, 10000
eng = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 64)
eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
lat = []
for i in range(300):
    t0 = time.perf_counter(); eng.detect_intra(n - 1 - (i % 90)); lat.append((time.perf_counter() - t0) * 1e6)
print("detect_intra p50 %.1f us" % np.percentile(lat[20:], 50), file=sys.stderr)
eng.close()
