import sys, time, numpy as np
sys.path.insert(0, '.')
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_scan
e = ScanContextEngine(num_ring=80, num_sector=180)
c = synth_scan(240000, seed=5)
for _ in range(3): e.voxel_grid(c, 0.4)
t0 = time.perf_counter()
for _ in range(20): o = e.voxel_grid(c, 0.4)
print("voxel 240k pts: %.3f ms per call, out %d" % ((time.perf_counter() - t0) / 20 * 1e3, o.shape[0]))
e.close()
