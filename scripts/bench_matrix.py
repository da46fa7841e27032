#!/usr/bin/env python3
"""The exact distance matrix (scl_sc_distance_matrix) on BASELINE configs[1]'s database: rows x 9 900 keyframes, 64x120 -- and on
configs[4]'s grid (80x180).  usage: bench_matrix.py [rows] [64x120|80x180]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = sys.argv[2] if len(sys.argv) > 2 else None
out = {}
for R, S, seed in ((64, 120, 1002), (80, 180, 1005)):
    if only and only != f"{R}x{S}":
        continue
    n = 10000
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n + 64)
    eng.save_bulk(synth_descriptors(n, R, S, seed=seed, revisit_frac=0.01))
    qs = (n - 100 + (np.arange(rows) % 100)).astype(np.int32)
    eng.sc_distance_matrix(qs[:16], 0, n - 100)
    ts = []
    eng.profile_reset(); eng.profile_enable(2)             # HIP events around every group of rows (screening launches + the exact kernel)
    for _ in range(3):
        t0 = time.perf_counter()
        d, s = eng.sc_distance_matrix(qs, 0, n - 100)
        ts.append(time.perf_counter() - t0)
    eng.profile_enable(0)
    prof = eng.profile()
    dt = sorted(ts)[1]
    pairs = rows * (n - 100)
    bpp = 4 * R * S + 8 * S
    out[f"{R}x{S}"] = {"rows": rows, "pairs_per_s": pairs / dt, "ms_per_row": dt / rows * 1e3, "frac_of_hbm_at_survey_bytes_per_pair": pairs * bpp / dt / 8e12,
                      "finite": int(np.isfinite(d).sum()),
                      "group_us": prof["sc_distance_ms"] / max(1, prof["sc_distance_launches"]) * 1e3,
                      "pairs_per_s_on_the_device": prof["sc_distance_pairs"] / max(1e-9, prof["sc_distance_ms"] * 1e-3)}
    eng.close()
print(json.dumps(out, indent=1))
