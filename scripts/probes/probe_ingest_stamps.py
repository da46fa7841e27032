#!/usr/bin/env python3
"""Phase stamps of ingest_kernel (diagnostics build, SCL_INGEST_STAMPS=1: printed by scl_destroy): groups of 16 scans at 64x120."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_scan
R, S, npts = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 120, 120000)
eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=4096)
clouds = [np.ascontiguousarray(synth_scan(npts, seed=300 + i, stride_floats=4)) for i in range(16)]
for _ in range(20):
    eng.make_and_save_many(clouds, want_values=False)
eng.close()
