#!/usr/bin/env python3
"""Randomised parity soak of the front half (raw points -> descriptor -> database slot -> detection): random grids, batch sizes (0, 1,
ragged, several groups of 16), record strides (12 / 16 / 32 bytes), clouds with boundary-hugging and special points mixed in, pinned
and pageable buffers.  scl_make_and_save_many against the CPU checker scan by scan (wire values, ring key: uint32 views) and
scl_stream_from_points against scl_make_and_save + scl_detect_full_range on a second engine (winner, shift, fp64 distance: bit
views).  Usage: soak_front.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_binding as ob  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import synth_descriptors, synth_scan  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
seed = scans = points = streams = 0
SPECIAL = np.array([0.0, -0.0, 1e-42, -1e-42, 1e-20, 1e20, np.inf, -np.inf, np.nan, 80.0, -80.0, 40.0, 1.25, 3e-5], dtype=np.float32)
while time.time() < t_end:
    seed += 1
    rs = np.random.RandomState(seed)
    R, S = [(64, 120), (20, 60), (80, 180), (64, 120), (20, 60)][seed % 5]
    stride = int(rs.choice([3, 4, 8]))
    nb = int(rs.choice([0, 1, 2, 5, 16, 17, 33, 40]))
    clouds = []
    for i in range(nb):
        n = int(rs.choice([0, 1, 7, 255, 256, 257, 4095, 4096, 4097, 20000, 60000, 130000]))
        c = synth_scan(n, seed=seed * 100 + i, stride_floats=stride) if n else np.zeros((0, stride), np.float32)
        if n > 50:
            k = n // 10
            # a tenth of the points ON bin boundaries (ring r * 80 / R, sector s * 360 / S), a few float steps either side, and specials
            idx = rs.choice(n, size=k, replace=False)
            ang = np.deg2rad(rs.randint(0, S + 1, size=k) * (360.0 / S) + rs.randint(-3, 4, size=k) * 1e-6)
            rad = rs.randint(1, R + 1, size=k) * (80.0 / R) * (1.0 + rs.randint(-3, 4, size=k) * 2.0 ** -23)
            c[idx, 0] = (rad * np.cos(ang)).astype(np.float32); c[idx, 1] = (rad * np.sin(ang)).astype(np.float32)
            sp = rs.choice(n, size=min(40, n), replace=False)
            c[sp, 0] = SPECIAL[rs.randint(0, len(SPECIAL), size=sp.size)]; c[sp, 1] = SPECIAL[rs.randint(0, len(SPECIAL), size=sp.size)]
            c[sp[:5], 2] = SPECIAL[rs.randint(0, len(SPECIAL), size=min(5, sp.size))]
        clouds.append(np.ascontiguousarray(c))
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=8)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    src = clouds
    if seed % 3 == 0 and nb:                                   # pinned buffers
        src = []
        for c in clouds:
            a = eng.host_alloc(c.shape if c.size else (1, stride)); a[:c.shape[0]] = c; src.append(a[:c.shape[0]])
    vals = eng.make_and_save_many(src)
    for i, c in enumerate(clouds):
        v = db.make_and_save(c, 0, i)
        assert np.array_equal(vals[i].view(np.uint32), v.view(np.uint32)), (seed, i, "values")
        assert np.array_equal(eng.get_ringkey(i).view(np.uint32), db.ringkey(i).view(np.uint32)), (seed, i, "ring key")
        scans += 1; points += c.shape[0]
    eng.close(); db.close()
    if seed % 4 == 1 and S != 60:                              # the whole pipeline against scan-by-scan calls
        n0, excl = int(rs.randint(0, 400)), int(rs.choice([3, 30, 100]))
        e1 = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=excl, initial_capacity=16)
        e2 = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=excl)
        if n0:
            base = synth_descriptors(n0, R, S, seed=seed); e1.save_bulk(base); e2.save_bulk(base)
        cl = [synth_scan(int(rs.randint(2000, 9000)), seed=seed * 1000 + i, stride_floats=4) for i in range(int(rs.randint(1, 40)))]
        for i in range(excl + 2, len(cl), 3):
            cl[i] = cl[i - excl - 2]
        nn, sh, dd = e1.stream_from_points(cl)
        for i, c in enumerate(cl):
            e2.make_and_save(c, 0, n0 + i)
            g = e2.detect_full_range(n0 + i, 0, max(0, n0 + i - excl))
            assert (int(nn[i]), int(sh[i])) == (g[0], g[1]) and np.float64(dd[i]).view(np.uint64) == np.float64(g[2]).view(np.uint64), (seed, i, nn[i], sh[i], dd[i], g)
        streams += len(cl)
        e1.close(); e2.close()
print(f"soak_front: {seed} configurations, {scans} scans ({points} points) bit-identical to the checker through scl_make_and_save_many, "
      f"{streams} scans through scl_stream_from_points equal to the scan-by-scan calls")
