#!/usr/bin/env python3
"""Does a spatially sorted source cloud speed the ICP's neighbour search up?  (host-side Morton sort, same clouds)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud

def morton(c, bits=10):
    p = c[:, :3].astype(np.float64); mn = p.min(0); ext = (p.max(0) - mn).max()
    q = np.minimum(((p - mn) / ext * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    key = np.zeros(len(c), np.int64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(key, kind="stable")

eng = ScanContextEngine(num_ring=64, num_sector=120)
NC = 25
tgts = [synth_structured_cloud(100000, seed=100 + c, extent=60.0) for c in range(NC)]
T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
rs = np.random.RandomState(3)
src = tgts[0].copy(); p = tgts[0][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
src[:, :3] = (p + 0.01 * rs.standard_normal(p.shape)).astype(np.float32)
srt = src[morton(src)]
for est in (0, 1):
    pp = eng.icp_default_params(); pp.max_iterations = 30; pp.estimator = est
    for name, s in (("unsorted", src), ("morton", srt), ("unsorted", src), ("morton", srt)):
        eng.icp_align_batch(s, tgts[:2], pp)
        t0 = time.perf_counter(); Tb, fb, cb, ib = eng.icp_align_batch(s, tgts, pp); dt = time.perf_counter() - t0
        print(f"estimator {est} {name}: {dt*1e3:.1f} ms per query, {dt*1e3/NC:.2f} ms per candidate, iterations {ib.mean():.1f}, fitness[0] {fb[0]:.6f}")
eng.close()
