#!/usr/bin/env python3
"""Randomised parity soak (GPU vs the CPU checker): many seeds, both wave-kernel grids, random exclusion
ranges and planted rotations; every sampled (query, keyframe) pair must agree bit for bit and every full-DB
winner must be the checker's winner over the sampled set's superset property (winner distance <= every
sampled distance, and equal to the checker's value for that pair).  Usage: soak_parity.py [seconds]"""
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
seed, pairs, wins = 0, 0, 0
while time.time() < t_end:
    seed += 1
    rs = np.random.RandomState(seed)
    R, S = [(64, 120), (20, 60)][seed % 2]
    n = int(rs.randint(300, 1500))
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
    eng.close()
    print(f"seed {seed}: {R}x{S} n={n} contrast={contrast}: ok ({pairs} pairs, {wins} winners so far)", flush=True)
print(f"soak done: {seed} databases, {pairs} pairs bit-identical, {wins} full-DB winners consistent")
