"""Is the SC-distance kernel sensitive to where the candidate data comes from (HBM vs L2/MALL)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
R, S, n = 64, 120, 10000
eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n + 64)
eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
def run(cand, label):
    for _ in range(3): eng.sc_distance_batch(n - 1, cand=cand)
    eng.profile_reset(); eng.profile_enable(2)
    for _ in range(20): eng.sc_distance_batch(n - 1, cand=cand)
    p = eng.profile(); eng.profile_enable(0)
    print(f"{label:40s} K1 {p['sc_distance_ms'] / p['sc_distance_launches'] * 1e3:8.1f} us")
run(np.arange(9900, dtype=np.int32), "9900 distinct slots (HBM stream)")
run((np.arange(9900) % 256).astype(np.int32), "9900 pairs over 256 slots (L2/MALL)")
run((np.arange(9900) % 8).astype(np.int32), "9900 pairs over 8 slots (L2)")
