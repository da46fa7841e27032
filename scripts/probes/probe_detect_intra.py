import os, sys, time
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
n = 10000
eng = ScanContextEngine(num_ring=64, num_sector=120, num_candidates=int(os.environ.get("K", "3")), num_exclude_recent=100, initial_capacity=n + 64)
eng.save_bulk(synth_descriptors(n, 64, 120, seed=1002, revisit_frac=0.01))
ts = []
for i in range(400):
    q = n - 100 + (i % 100)
    t0 = time.perf_counter(); r = eng.detect_intra(q); ts.append(time.perf_counter() - t0)
ts = np.array(ts[50:]) * 1e6
print(f"detect_intra k={os.environ.get('K','3')}: p50 {np.percentile(ts,50):.2f} us  p10 {np.percentile(ts,10):.2f}  p99 {np.percentile(ts,99):.2f}  last {r}")
eng.close()
