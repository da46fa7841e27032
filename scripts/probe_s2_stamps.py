#!/usr/bin/env python3
"""Diagnostic (variant build -DS2_STAMP): s_memtime stamps of one wave per iteration of the products' inner loop."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SCL_ENGINE_LIB"] = os.path.join(ROOT, "scl_slam_amd/lib/variants/libscl_engine_stamp.so")
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
from scl_slam_amd import _native
R, S, n = 64, 120, 10000
e = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n + 64)
e.save_bulk(synth_descriptors(n, R, S, seed=11, revisit_frac=0.0))
qs = (n - 100 + (np.arange(256) % 100)).astype(np.int32)
e.detect_full_stream(qs, 0, n - 100, 16, 2)
lib = e._lib
buf = (ctypes.c_ulonglong * 256)()
lib.scl_debug_s2_stamps(buf)
full = np.array(buf[:], dtype=np.uint64).reshape(4, 64).astype(np.int64)
a = full[:, :30]
for r in range(4):
    d = np.diff(a[r])
    print("wave", r, "iteration deltas (s_memtime ticks):", d.tolist(), "total", int(a[r, -1] - a[r, 0]))
    t0 = full[r, 32]
    ph = [("requested", full[r, 61] - t0), ("prefetch issued", full[r, 62] - t0), ("stored", full[r, 63] - t0), ("staged (barrier)", full[r, 33] - t0)]
    for k in range(6):
        if full[r, 34 + 3 * k]:
            ph.append((f"k{k}: start +{full[r, 34 + 3 * k] - t0}", f"loop {full[r, 35 + 3 * k] - full[r, 34 + 3 * k]}", f"epilogue {full[r, 36 + 3 * k] - full[r, 35 + 3 * k]}"))
    ph.append(("end", full[r, 60] - t0))
    print("   ", ph)
e.close()
