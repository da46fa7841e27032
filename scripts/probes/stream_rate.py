#!/usr/bin/env python3
"""Full-DB stream form, 10 000 keyframes x 64 x 120, with and without the profile's event pairs: scans per second as the
host sees it.  usage: stream_rate.py [n_scans]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from scl_slam_amd import ScanContextEngine

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
R, S, N = 64, 120, 10000
rng = np.random.default_rng(1)
eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 256)
eng.save_bulk(rng.uniform(0, 8, size=(N + 128, R * S)).astype(np.float32))
q = (N + (np.arange(n_scans) % 128)).astype(np.int32)
for prof in (0, 3, 0):
    eng.profile_reset(); eng.profile_enable(prof)
    eng.detect_full_stream(q[:64], 0, N, 4, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.detect_full_stream(q, 0, N, 4, 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"profile={prof}: {dt / n_scans * 1e6:.2f} us per scan, {N * n_scans / dt / 1e6:.1f} M pairs/s")
