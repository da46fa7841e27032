import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_structured_cloud, rigid_transform
from test_oracle_icp_kat import moved_copy
e = ScanContextEngine()
tgt = synth_structured_cloud(100000, seed=11, extent=60.0)
T = rigid_transform(0.01, -0.02, 0.03, 0.4, -0.3, 0.1)
src = moved_copy(tgt, T, keep_every=4, noise=0.02, seed=3)
src[::7, :3] += 3.0                     # outliers
for iters in (1000, 2000):
    e.geometric_verification(src, tgt, iters, 0.25, 0.45, 1)
    t0 = time.perf_counter()
    for _ in range(5):
        Tm, ok, nc, ni = e.geometric_verification(src, tgt, iters, 0.25, 0.45, 1)
    print("geometric_verification %d pts vs %d pts, %d hypotheses: %.2f ms (success %s, %d corr, %d inliers)" % (src.shape[0], tgt.shape[0], iters, (time.perf_counter() - t0) / 5 * 1e3, ok, nc, ni))
e.close()
