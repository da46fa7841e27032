#!/usr/bin/env python3
"""BASELINE configs[2]: geometric verification of the top-25 loop candidates of one scan
(~100 k points per cloud, 30 iterations max), point-to-point (the reference's estimator,
DM.h:1108) and point-to-plane (the BASELINE wording), clouds on the host vs resident in the
on-device keyframe store.  Secondary measurement, not the headline."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud  # noqa: E402
from test_oracle_icp_kat import moved_copy  # noqa: E402

NC = int(os.environ.get("NC", "25"))
eng = ScanContextEngine(num_ring=64, num_sector=120)
ident = np.eye(4, dtype=np.float32)
tgts, srcs = [], []
for c in range(NC):
    tgt = synth_structured_cloud(100000, seed=100 + c, extent=60.0)
    T = rigid_transform(0.004, -0.006, 0.02, 0.25 - 0.01 * c, -0.15, 0.05)
    tgts.append(tgt)
    srcs.append(moved_copy(tgt, T, keep_every=1, noise=0.01, seed=3 + c))
    eng.keyframe_put(0, 2 * c, tgt)
    eng.keyframe_put(0, 2 * c + 1, srcs[-1])
out = {"candidates": NC, "points_per_cloud": 100000}
for est, name in ((0, "point_to_point"), (1, "point_to_plane")):
    pp = eng.icp_default_params(); pp.max_iterations = 30; pp.estimator = est; pp.normal_radius = 1.0
    eng.icp_align(srcs[0], tgts[0], pp)
    t0 = time.perf_counter(); its = []
    for c in range(NC):                                   # the scan (srcs[0]) against each of its NC candidates
        T, f, conv, it = eng.icp_align(srcs[0], tgts[c], pp)
        its.append(it)
    host = time.perf_counter() - t0
    # the reference's use: ONE scan against its NC candidates, verified together
    eng.icp_align_batch(srcs[0], tgts, pp)                # warm-up with the whole batch (every lane workspace sized)
    t0 = time.perf_counter()
    Tb, fb, cb, ib = eng.icp_align_batch(srcs[0], tgts, pp)
    batch = time.perf_counter() - t0
    out[name] = {"one_by_one_ms_total": host * 1e3, "one_by_one_ms_per_candidate": host * 1e3 / NC,
                 "iterations_mean": float(np.mean(its)),
                 "batch_ms_total": batch * 1e3, "batch_ms_per_candidate": batch * 1e3 / NC,
                 "batch_iterations_mean": float(np.mean(ib))}
print(json.dumps(out, indent=1))
eng.close()
