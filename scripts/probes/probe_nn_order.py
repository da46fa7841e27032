import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_structured_cloud, rigid_transform
from test_oracle_icp_kat import moved_copy
e = ScanContextEngine()
tgt = synth_structured_cloud(100000, seed=11, extent=60.0)
T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
src = moved_copy(tgt, T, keep_every=1, noise=0.01, seed=3)
h = 0.625
key = (np.floor(src[:, 2] / h).astype(np.int64) * 4096 + np.floor(src[:, 1] / h).astype(np.int64)) * 4096 + np.floor(src[:, 0] / h).astype(np.int64)
src_sorted = src[np.argsort(key, kind="stable")]
rs = np.random.RandomState(0); src_rand = src[rs.permutation(src.shape[0])]
p = e.icp_default_params(); p.max_iterations = 30
for name, s in (("as generated", src), ("random order", src_rand), ("sorted by cell", src_sorted)):
    e.icp_align(s, tgt, p)
    t0 = time.perf_counter()
    for _ in range(5): Tm, f, c, it = e.icp_align(s, tgt, p)
    print("%-16s %.3f ms per alignment (%d iterations)" % (name, (time.perf_counter() - t0) / 5 * 1e3, it))
e.close()
