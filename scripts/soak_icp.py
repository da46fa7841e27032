#!/usr/bin/env python3
"""Randomised soak of the geometric verification (GPU vs the CPU checker), the counterpart of soak_parity.py for icp.hip:
  * exact nearest neighbours, cold and after a random move (index and distance bits) -- random sizes, extents, sources partly far
    outside the target, duplicated target points, clouds on a plane / a line;
  * whole alignments, both estimators: same convergence flag and iteration count, transform within 1e-5, fitness within 1e-4 rel.;
  * a batch of candidates against the same alignments one at a time: bit for bit (the batch takes the LDS tiles from 300 k queries
    on, the lone alignment searches in memory).
Usage: soak_icp.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_icp_binding as oi  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud  # noqa: E402
from test_oracle_icp_kat import moved_copy  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
eng = ScanContextEngine(num_ring=64, num_sector=120)
seed = nn_q = aligns = batches = 0
TOL = 1e-5
while time.time() < t_end:
    seed += 1
    rs = np.random.RandomState(seed)
    ext = float(rs.choice([8.0, 25.0, 60.0, 120.0]))
    n_tgt = int(rs.randint(200, 70000)); n_src = int(rs.randint(1, 70000))
    tgt = synth_structured_cloud(n_tgt, seed=2 * seed, extent=ext)
    src = synth_structured_cloud(n_src, seed=2 * seed + 1, extent=ext)
    kind = seed % 5
    if kind == 1: src[: max(1, n_src // 20), :3] += 5.0 * ext                 # a part of the scan far outside the target
    if kind == 2: tgt[n_tgt // 2:] = tgt[: n_tgt - n_tgt // 2]                 # every target point twice: ties by index
    if kind == 3: tgt[:, 2] = 0.0                                              # a flat target
    if kind == 4 and n_tgt > 400: tgt = tgt[:400].copy(); tgt[:, 1:3] = 0.0; n_tgt = 400   # a few points on a line
    # -- neighbours, cold and warm
    gi, gd = eng.nn_correspondences(src, tgt)
    oi_, od = oi.nn(src, tgt, use_grid=True)
    assert np.array_equal(gi, oi_) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), ("cold nn", seed)
    mv = float(rs.choice([1e-4, 1e-2, 0.3, 3.0]))
    T = rigid_transform(*(rs.uniform(-1, 1, 3) * 0.01 * min(1.0, mv)), *(rs.uniform(-1, 1, 3) * mv)).astype(np.float32)
    gi, gd = eng.nn_correspondences_moved(src, tgt, T)
    oi_, od = oi.nn(oi.transform(src, T), tgt, use_grid=True)
    assert np.array_equal(gi, oi_) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), ("warm nn", seed)
    nn_q += 2 * n_src
    # -- an alignment against the checker (a moved, noisy copy of a structured cloud)
    if seed % 3 == 0:
        n = int(rs.randint(3000, 40000))
        base = synth_structured_cloud(n, seed=7 * seed, extent=40.0)
        Tm = rigid_transform(*(rs.uniform(-1, 1, 3) * 0.03), *(rs.uniform(-1, 1, 3) * 0.4))
        s2 = moved_copy(base, Tm, keep_every=int(rs.randint(1, 4)), noise=float(rs.choice([0.0, 0.005, 0.02])), seed=seed)
        for est in (0, 1):
            pg = eng.icp_default_params(); pg.max_iterations = 30; pg.estimator = est; pg.normal_radius = 1.5
            po = oi.default_params(); po.max_iterations = 30; po.estimator = est; po.normal_radius = 1.5
            Tg, fg, cg, ig = eng.icp_align(s2, base, pg)
            To, fo, co, io = oi.icp_align(s2, base, po)
            assert cg == co and ig == io, ("icp flags", seed, est, cg, co, ig, io)
            assert np.abs(Tg - To).max() < TOL and abs(fg - fo) <= 1e-4 * max(1e-6, abs(fo)) + 1e-12, ("icp values", seed, est)
            aligns += 1
    # -- a batch (tiles) against one at a time (memory): bit for bit
    if seed % 7 == 0:
        nc = int(rs.randint(6, 10)); npt = int(rs.randint(45000, 70000))
        cands = [synth_structured_cloud(npt, seed=11 * seed + c, extent=60.0) for c in range(nc)]
        Tm = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
        s3 = moved_copy(cands[0], Tm, keep_every=1, noise=0.01, seed=seed)
        for est in (0, 1):
            pg = eng.icp_default_params(); pg.max_iterations = 30; pg.estimator = est; pg.normal_radius = 1.0
            Tb, fb, cb, ib = eng.icp_align_batch(s3, cands, pg)
            for c in (0, nc - 1):
                T1, f1, c1, i1 = eng.icp_align(s3, cands[c], pg)
                assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == f1 and bool(cb[c]) == c1 and ib[c] == i1, ("batch", seed, est, c)
        batches += 1
    if seed % 10 == 0:
        print(f"seed {seed}: ok ({nn_q} neighbour queries, {aligns} alignments against the checker, {batches} batches against one-by-one so far)", flush=True)
print(f"soak done: {seed} seeds, {nn_q} neighbour queries bit-identical, {aligns} alignments equal to the checker's (flags, iterations, 1e-5), {batches} batches bit-identical to one-by-one")
eng.close()
