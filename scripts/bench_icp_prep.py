#!/usr/bin/env python3
"""configs[2] query (one scan against 25 candidates of 100 k points): what is preparation and what is the loop.  The same call
with max_iterations = 1 (preparation + cold search + solve + fitness pass) and 30, clouds in the keyframe store and as host buffers."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
n_cand, n_pts = int(os.environ.get("NC", "25")), 100000
eng = ScanContextEngine(num_ring=64, num_sector=120)
ident = np.eye(4, dtype=np.float32)
tgts = [synth_structured_cloud(n_pts, seed=100 + c, extent=60.0) for c in range(n_cand)]
T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
rs = np.random.RandomState(3); src0 = tgts[0].copy()
p = tgts[0][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
src0[:, :3] = (p + 0.01 * rs.standard_normal(p.shape)).astype(np.float32)
for c in range(n_cand): eng.keyframe_put(0, c, tgts[c])
eng.keyframe_put(0, n_cand, src0)
keys = np.arange(n_cand, dtype=np.int32); poses = np.tile(ident.reshape(1, 1, 16), (n_cand, 1, 1))
out = {}
for est, name in ((0, "point_to_point"), (1, "point_to_plane")):
    for iters in (1, 30):
        pp = eng.icp_default_params(); pp.max_iterations = iters; pp.estimator = est; pp.normal_radius = 1.0
        for mode in ("from_store", "host_buffers"):
            def run():
                if mode == "from_store":
                    return eng.loop_icp_batch_from_store(0, n_cand, ident, keys, 0, poses, 0.05, pp)[3]
                return eng.icp_align_batch(src0, tgts, pp)[3]
            run()
            ts = []
            for rep in range(5):
                t0 = time.perf_counter(); it = run(); ts.append(time.perf_counter() - t0)
            out[f"{name}.{mode}.max_iter_{iters}"] = {"ms_median": float(np.median(ts)) * 1e3, "ms_min": float(np.min(ts)) * 1e3, "iterations_mean": float(np.mean(it))}
print(json.dumps(out, indent=1))
eng.close()
