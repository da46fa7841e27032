#!/usr/bin/env python3
"""Secondary measurements (not the headline): descriptor construction (K3) and ICP geometric
verification (K4-K6) at the sizes of BASELINE configs[1]/[2], GPU vs the CPU restatement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_binding as ob  # noqa: E402
import oracle_icp_binding as oi  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import rigid_transform, synth_scan, synth_structured_cloud  # noqa: E402
from test_oracle_icp_kat import moved_copy  # noqa: E402

out = {}
eng = ScanContextEngine(num_ring=64, num_sector=120)
cloud = synth_scan(120000, seed=3)
for _ in range(3):
    eng.make_descriptor(cloud)
eng.profile_reset(); eng.profile_enable(1)
t0 = time.perf_counter()
for _ in range(20):
    v = eng.make_descriptor(cloud)
wall = (time.perf_counter() - t0) / 20
p = eng.profile(); eng.profile_enable(0)
k3 = p["make_sc_ms"] / max(1, p["make_sc_launches"])
cfg = ob.make_config(R=64, S=120)
t0 = time.perf_counter(); vc = ob.make_scancontext(cfg, cloud); cpu = time.perf_counter() - t0
assert np.array_equal(v, vc)
out["make_sc_120k_pts"] = {"gpu_kernels_ms": k3, "gpu_call_incl_pcie_ms": wall * 1e3, "cpu_oracle_ms": cpu * 1e3,
                           "algorithmic_GBps": 120000 * 16 / (k3 * 1e-3) / 1e9}

tgt = synth_structured_cloud(100000, seed=11, extent=60.0)
T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
src = moved_copy(tgt, T, keep_every=1, noise=0.01, seed=3)
pp = eng.icp_default_params(); pp.max_iterations = 30
eng.icp_align(src, tgt, pp)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    Tg, fg, cg, ig = eng.icp_align(src, tgt, pp)
gpu = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); To, fo, co, io = oi.icp_align(src, tgt, oi.default_params(max_iterations=30)); cpu = time.perf_counter() - t0
assert ig == io and np.abs(Tg - To).max() < 1e-5
out["icp_100k_x_100k_30it"] = {"gpu_ms_per_problem_incl_pcie": gpu * 1e3, "iterations": ig, "cpu_oracle_grid_nn_1thread_ms": cpu * 1e3,
                               "speedup": cpu / gpu, "gpu_ms_for_25_candidates": 25 * gpu * 1e3}
print(json.dumps(out, indent=1))
