#!/usr/bin/env python3
"""Randomised parity soak (GPU vs the CPU checker): many seeds, the three tuned grids (64x120, 20x60, 80x180), random exclusion
ranges and planted rotations; every sampled (query, keyframe) pair must agree bit for bit and every full-DB
winner must be the checker's winner over the sampled set's superset property (winner distance <= every
sampled distance, and equal to the checker's value for that pair); the stream form (launches of up to 16 scans with ragged
ranges: the second form of the screening products) must return what the one-scan calls return.  Usage: soak_parity.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_binding as ob  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import synth_descriptors  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
seed, pairs, wins, streams, mats = 0, 0, 0, 0, 0
while time.time() < t_end:
    seed += 1
    rs = np.random.RandomState(seed)
    R, S = [(64, 120), (20, 60), (80, 180), (64, 120)][seed % 4]
    n = int(rs.randint(300, 1500)) if S != 180 else int(rs.randint(300, 800))
    contrast = float(rs.choice([1.0, 0.1, 0.01]))                      # low contrast -> alignment near ties
    descs = synth_descriptors(n, R, S, seed=1000 + seed, revisit_frac=0.05)
    if contrast != 1.0:
        m = descs.mean()
        descs = ((descs - m) * contrast + m).astype(np.float32) * (descs > 0)
    for _ in range(5):                                                  # planted rotations of old keyframes
        q, j = int(rs.randint(n // 2, n)), int(rs.randint(0, n // 2))
        descs[q] = np.roll(descs[j], int(rs.randint(0, S)), axis=1)
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=20, initial_capacity=2048)
    eng.save_bulk(descs)
    cfg = ob.make_config(R=R, S=S)
    for _ in range(6):
        q = int(rs.randint(n // 2, n))
        hi = int(rs.randint(1, q))
        dist, sh = eng.sc_distance_batch(q, n=hi)
        nn, shift, d = eng.detect_full_range(q, 0, hi)
        j = int(np.argmin(dist))
        assert (nn, shift) == (j, int(sh[j])) and d == dist[j], (seed, q, hi)
        for c in [j] + [int(x) for x in rs.randint(0, hi, 40)]:
            dc, sc = ob.distance(cfg, descs[q], descs[c], fast=True)
            assert sc == sh[c] and np.float64(dc).view(np.uint64) == dist[c:c + 1].view(np.uint64)[0], (seed, q, c, dc, dist[c], sc, sh[c])
            pairs += 1
        wins += 1
    # the stream form: 5 to 40 scans, every one with its own range (some empty, some one keyframe long)
    m = int(rs.randint(5, 41))
    qs = rs.randint(n // 2, n, size=m).astype(np.int32)
    los = np.array([int(rs.randint(0, max(1, q // 2))) for q in qs], np.int32)
    his = np.array([int(rs.randint(lo, q)) if rs.random_sample() > 0.1 else lo + int(rs.randint(0, 2)) for lo, q in zip(los, qs)], np.int32)
    nn_s, sh_s, d_s = eng.detect_full_stream(qs, los, his, 16, 2)
    for i in range(m):
        nn1, sh1, d1 = eng.detect_full_range(int(qs[i]), int(los[i]), int(his[i]))
        assert (int(nn_s[i]), int(sh_s[i])) == (nn1, sh1) and np.float64(d_s[i]).view(np.uint64) == np.float64(d1).view(np.uint64), (seed, i, qs[i], los[i], his[i])
        streams += 1
    # the exact distance matrix (screened grids: alignment + screening + shift masks, then the open shifts in fp64): 1 to 19 rows over a
    # random sub-range against the 13-shift kernel of the sampled pairs' path above (another program), entry for entry
    if S != 60:
        rows = rs.randint(0, n, size=int(rs.randint(1, 20))).astype(np.int32)
        lo_m = int(rs.randint(0, n - 1)); hi_m = int(rs.randint(lo_m + 1, n + 1))
        dm, sm = eng.sc_distance_matrix(rows, lo_m, hi_m)
        for r, q in enumerate(rows):
            d1, s1 = eng.sc_distance_batch(int(q), cand=np.arange(lo_m, hi_m, dtype=np.int32))
            assert np.array_equal(dm[r].view(np.uint64), d1.view(np.uint64)) and np.array_equal(sm[r], s1), (seed, int(q), lo_m, hi_m)
            mats += hi_m - lo_m
    eng.close()
    print(f"seed {seed}: {R}x{S} n={n} contrast={contrast}: ok ({pairs} pairs, {wins} winners, {streams} streamed scans, {mats} matrix entries so far)", flush=True)
print(f"soak done: {seed} databases, {pairs} pairs bit-identical, {wins} full-DB winners consistent, {streams} streamed scans equal to their one-scan calls, "
      f"{mats} distance-matrix entries equal to the 13-shift kernel's")
