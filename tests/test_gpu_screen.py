"""The screening pass of the full-DB mode (scl_slam_amd/csrc/sc_screen.hip) on the hardware: its reduced-precision
distances must stay inside the stated bound around the exact fp64 distances (the CPU checker's), the survivor set must
contain every keyframe that can hold the minimum, and the full-DB winner must be the checker's bit for bit -- also on
inputs built to stress the bound (near ties, zero sectors, tiny / huge / non-finite values)."""
import numpy as np
import pytest

import oracle_binding as ob
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu
R, S = 64, 120


def test_screening_on_the_80x180_grid():
    """BASELINE configs[4]'s grid: W = 19 shifts (two M tiles), 23 k-steps per ring group with the last one partly empty,
    exact three-shifts-per-lane alignment, exact pass by the one-sector-per-lane kernel"""
    R2, S2, n = 80, 180, 1300
    descs = synth_descriptors(n, R2, S2, seed=1005, revisit_frac=0.03)
    rs = np.random.RandomState(2)
    descs[40] = 0.0; descs[41][:, ::3] = 0.0; descs[42] = descs[n - 1] * np.float32(1e25); descs[43][5, 7] = np.nan
    for i, mag in enumerate([0.0, 1e-6, 1e-4, 1e-3, 3e-3]):
        d = np.roll(descs[n - 1], int(rs.randint(0, S2)), axis=1)
        descs[60 + 9 * i] = np.clip(d + mag * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R2, num_sector=S2, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R2, S=S2))
    eng.save_bulk(descs); db.save_bulk(descs)
    worst = 0.0
    for q, lo, hi in ((n - 1, 0, n - 100), (n - 2, 3, n - 101), (700, 0, 600), (n - 3, 0, 17), (40, 0, 300), (42, 0, 200)):
        e, m = _check(eng, db, q, lo, hi)
        worst = max(worst, e)
    print(f"screening 80x180: worst |d~ - d| = {worst:.3e}")
    assert worst < 3e-4
    # the stream / batched forms go through the same path one query at a time
    qs = np.arange(n - 1, n - 9, -1, dtype=np.int32)
    nn, sh, dd = eng.detect_full_stream(qs, 0, qs - 100, 4, 2)
    for i, q in enumerate(qs):
        o = db.detect_full(int(q))
        assert (nn[i], sh[i]) == (o[1], o[2]) and dd[i].view(np.uint64) == np.float64(o[3]).view(np.uint64)
    eng.close()


def _check(eng, db, query, lo, hi):
    approx, surv, eps = eng.screen_distances(query, lo, hi)
    d_ref, s_ref = db.distance_batch(query, cand=np.arange(lo, hi, dtype=np.int32))
    finite = np.isfinite(approx)
    ok = d_ref < 1e7                                          # 1e7 = the reference's "no finite distance"
    assert np.all(approx[~ok & ~np.isneginf(approx)] == np.inf)
    err = np.abs(approx[finite & ok].astype(np.float64) - d_ref[finite & ok])
    assert err.size == 0 or err.max() <= eps, (err.max(), eps)
    if ok.any():
        best = int(np.flatnonzero(ok)[np.argmin(d_ref[ok])]) + lo        # first minimum = lowest slot
        assert best in surv
        assert np.all(np.diff(surv) > 0)
        # everything that is not a survivor is provably worse than the best upper bound
        mask = np.ones(hi - lo, bool); mask[surv - lo] = False
        assert np.all(d_ref[mask] > d_ref[best - lo])
    nn, sh, d = eng.detect_full_range(query, lo, hi)
    if ok.any():
        assert nn == best and sh == s_ref[best - lo] and np.float64(d).view(np.uint64) == d_ref[best - lo].view(np.uint64)
    else:
        assert nn == -1
    return err.max() if err.size else 0.0, len(surv)


@pytest.mark.parametrize("R2,S2,n", [(64, 120, 900), (80, 180, 700)])
def test_every_column_of_a_screening_launch_meets_the_bound(R2, S2, n):
    """The products' second form scores up to sixteen scans per launch, one per column of its matrix products (80 x 180: five ring
    slices of 16, two sectors per k-step, the second pass on the keyframe fragment of three iterations ago, columns 8 .. 15 in the
    second block of the scans' LDS image).  Every scan of launches of 2 .. 16 scans -- each rotated differently, so that every
    column has its own first shift against every keyframe -- is compared with the checker's fp64 distance: inside the bound, and
    the all-zero / NaN / huge keyframes flagged for the exact pass."""
    descs = synth_descriptors(n, R2, S2, seed=2024, revisit_frac=0.05)
    rs = np.random.RandomState(5)
    descs[30] = 0.0; descs[31][:, ::3] = 0.0; descs[32] = descs[n - 1] * np.float32(1e25); descs[33][5, 7] = np.nan
    for i in range(16):                                        # the scans: rotated, perturbed copies of keyframes all over the database
        src = descs[int(rs.randint(40, n - 40))]
        d = np.roll(src, int(rs.randint(0, S2)), axis=1)
        descs[n - 16 + i] = np.clip(d + np.float32(10.0 ** -(i % 4 + 2)) * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R2, num_sector=S2, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R2, S=S2))
    eng.save_bulk(descs); db.save_bulk(descs)
    worst = 0.0
    for nq, lo, hi in ((16, 0, n - 16), (9, 3, n - 50), (5, 17, 300), (2, 0, 131), (12, 200, 217), (16, 0, 5)):
        qs = np.arange(n - nq, n, dtype=np.int32)
        approx, eps = eng.screen_distances_many(qs, lo, hi)
        assert approx.shape == (nq, hi - lo)
        for i, q in enumerate(qs):
            d_ref, _ = db.distance_batch(int(q), cand=np.arange(lo, hi, dtype=np.int32))
            a = approx[i]
            finite = np.isfinite(a)
            ok = d_ref < 1e7
            assert np.all(a[~ok & ~np.isneginf(a)] == np.inf), (nq, i)
            err = np.abs(a[finite & ok].astype(np.float64) - d_ref[finite & ok])
            assert err.size == 0 or err.max() <= eps, (nq, i, float(err.max()), eps)
            assert finite[ok].mean() > 0.9 or ok.sum() < 8, (nq, i)     # the bound is met by values, not by flagging everything
            worst = max(worst, float(err.max()) if err.size else 0.0)
    print(f"{R2}x{S2}: worst |d~ - d| over all columns = {worst:.3e}")
    assert worst < 4e-4
    eng.close()


def test_screening_bound_and_survivors_on_the_bench_database():
    n = 3000
    descs = synth_descriptors(n, R, S, seed=1002, revisit_frac=0.02)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    worst, most = 0.0, 0
    for q, lo, hi in ((n - 1, 0, n - 100), (n - 2, 7, n - 101), (1500, 0, 1400), (n - 3, 0, 16), (n - 4, 0, 17), (n - 5, 33, 34)):
        e, m = _check(eng, db, q, lo, hi)
        worst, most = max(worst, e), max(most, m)
    print(f"screening: worst |d~ - d| = {worst:.3e}, most survivors = {most}")
    assert worst < 3e-4                                        # observed error is far inside the a-priori bound
    eng.close()


def test_ranged_and_full_passes_interleaved():
    """The call pattern of round 2's abort: the diagnostic entry point over a SUB-RANGE of the database between
    full-range passes of the stream form (16 scans per launch: second form of the products, last-workgroup reductions, buffer
    halves reused), ranges growing and shrinking, on one engine.  Every pass is compared with the checker."""
    n = 2100
    descs = synth_descriptors(n, R, S, seed=1011, revisit_frac=0.03)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    qs = np.arange(n - 1, n - 41, -1, dtype=np.int32)
    want = [db.detect_full(int(q)) for q in qs]

    def full():
        nn, sh, dd = eng.detect_full_stream(qs, 0, qs - 100, 16, 2)
        for i in range(len(qs)):
            assert (nn[i], sh[i]) == (want[i][1], want[i][2]) and dd[i].view(np.uint64) == np.float64(want[i][3]).view(np.uint64)

    for q, lo, hi in ((n - 1, 0, n - 100), (n - 2, 1900, 1990), (700, 0, 33), (n - 3, 5, 6), (n - 4, 0, n - 104), (40, 1000, 1017)):
        full()
        _check(eng, db, q, lo, hi)
    full()
    eng.close()


def test_survivor_statistics_and_a_database_where_many_keyframes_survive():
    """scl_survivor_stats counts what the exact pass scores.  The headline rate depends on it (VERDICT r2 #1c): on the bench
    database a scan leaves a handful of survivors; with 5 % of the database within the screening margin of the winner the
    pass must still return the reference's winner bit for bit -- it just scores that many keyframes exactly."""
    n = 2500
    descs = synth_descriptors(n, R, S, seed=1002, revisit_frac=0.02)
    rs = np.random.RandomState(17)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    qs = np.arange(n - 1, n - 33, -1, dtype=np.int32)
    eng.survivor_stats(reset=True)
    nn, sh, dd = eng.detect_full_stream(qs, 0, qs - 100, 16, 2)
    q, tot, mx = eng.survivor_stats(reset=True)
    assert q == len(qs) and 1 <= mx <= 64 and tot >= q, (q, tot, mx)
    eng.close()
    # adversarial: 5 % of the keyframes are noisy rolled copies of ONE scan, all within 2 eps of each other
    adv = descs.copy()
    base = adv[n - 1].copy()
    planted = rs.choice(n - 200, size=(n - 200) // 20, replace=False)
    for j in planted:
        d = np.roll(base, int(rs.randint(0, S)), axis=1)
        adv[j] = np.clip(d + np.float32(rs.uniform(1e-4, 2e-3)) * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db2 = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(adv); db2.save_bulk(adv)
    qs2 = np.full(16, n - 1, dtype=np.int32)
    nn, sh, dd = eng.detect_full_stream(qs2, 0, qs2 - 100, 16, 2)
    q, tot, mx = eng.survivor_stats()
    o = db2.detect_full(n - 1)
    assert np.all(nn == o[1]) and np.all(sh == o[2]) and np.all(dd.view(np.uint64) == np.float64(o[3]).view(np.uint64))
    assert q == 16 and mx >= len(planted) // 2, (q, tot, mx, len(planted))          # most of the planted copies had to be scored exactly
    print(f"survivors: bench database <= 64 per scan; adversarial {mx} of {n - 101} eligible ({len(planted)} planted)")
    eng.close(); db.close(); db2.close()


@pytest.mark.parametrize("R2,S2,n", [(64, 120, 2500), (80, 180, 1500)])
def test_the_stream_changes_its_exact_pass_with_the_survivors_it_sees(R2, S2, n):
    """The stream form scores a chunk's survivors with one workgroup per scan (sc_small_exact_kernel) or, when the chunks collected
    last left dozens per scan, with the survivors' kernel (80 x 180: select + masked kernel + arg-min); the choice follows the
    results with two chunks' delay.  A call whose first scans leave a hundred survivors each and whose later scans leave a handful goes
    through both passes and both switches: every winner must be the checker's, bit for bit, whichever kernel scored it."""
    descs = synth_descriptors(n, R2, S2, seed=1002, revisit_frac=0.02)
    rs = np.random.RandomState(23)
    base = descs[n - 1].copy()
    planted = rs.choice(n - 200, size=(n - 200) // 20, replace=False)
    for j in planted:
        d = np.roll(base, int(rs.randint(0, S2)), axis=1)
        descs[j] = np.clip(d + np.float32(rs.uniform(1e-4, 2e-3)) * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R2, num_sector=S2, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R2, S=S2))
    eng.save_bulk(descs); db.save_bulk(descs)
    light = [q for q in range(n - 2, n - 40, -1) if q not in set(planted.tolist())][:6]
    qs = np.concatenate([np.full(400, n - 1), np.resize(np.array(light), 700), np.full(300, n - 1), np.resize(np.array(light), 200)]).astype(np.int32)
    want = {int(q): db.detect_full(int(q)) for q in set(qs.tolist())}
    eng.survivor_stats(reset=True)
    nn, sh, dd = eng.detect_full_stream(qs, 0, qs - 100, 16, 2)
    for i, q in enumerate(qs):
        o = want[int(q)]
        assert (nn[i], sh[i]) == (o[1], o[2]) and dd[i].view(np.uint64) == np.float64(o[3]).view(np.uint64), (i, int(q), nn[i], o[1])
    cnt, tot, mx = eng.survivor_stats()
    # (on the 80 x 180 database the planted copies are within the margin of every scan's minimum: all of its chunks are heavy)
    assert cnt == len(qs) and mx >= len(planted) // 2 and (tot < len(qs) * mx or S2 == 180), (cnt, tot, mx)
    eng.close(); db.close()


def test_screening_with_adversarial_descriptors():
    rs = np.random.RandomState(11)
    n = 700
    descs = synth_descriptors(n, R, S, seed=5, revisit_frac=0.0)
    q = n - 1
    base = descs[q].copy()
    # near ties around the minimum: copies of the query with perturbations from 1e-7 to 1e-2, rolled
    for i, mag in enumerate([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 0.0, 1e-6]):
        d = np.roll(base, int(rs.randint(0, S)), axis=1)
        descs[10 + 7 * i] = np.clip(d + mag * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    descs[100] = 0.0                                           # all-zero keyframe: NaN distance in the reference
    descs[101][:, ::2] = 0.0                                   # half the sectors empty
    descs[102] = base * np.float32(1e-30)                      # tiny values: norms ~1e-29 (still inside the fp32 reciprocal range)
    descs[103] = base * np.float32(1e25)                       # huge values: norm outside [2^-60, 2^60] -> exact only
    descs[104] = base * np.float32(1e-38)                      # subnormal products
    descs[105][3, 5] = np.inf
    descs[106][0, 0] = np.nan
    descs[107] = np.where(rs.random_sample(base.shape) < 0.98, 0.0, base)    # almost empty
    descs[108][:] = 0.0; descs[108][17, 40] = 3.5              # a single non-zero cell
    descs[109] = -base                                         # negative heights: cosine -1
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    _check(eng, db, q, 0, n - 100)
    for qq in (100, 101, 103, 105, 106, 108, 109, 17):         # degenerate queries as well
        _check(eng, db, qq, 0, n - 1 if qq != n - 1 else n - 100)
    # every candidate flagged for the exact path
    approx, surv, _ = eng.screen_distances(103, 0, 300)
    assert np.all(np.isneginf(approx)) and len(surv) == 300
    eng.close()


@pytest.mark.parametrize("R,S", [(64, 120), (80, 180)])
def test_alignment_kernel_on_ties_and_near_ties(R, S):
    """The first shift comes from the fp32 matrix-core correlation of sc_align_kernel only when it leads every other shift
    by more than the filter's margin; ties (the reference keeps the lowest shift), near ties, flat and periodic sector
    keys, extreme magnitudes and non-finite values must fall through to the reference's own fp64 evaluation -- any wrong
    first shift moves the 13-shift window and shows as a distance outside the bound."""
    n = 700
    rs = np.random.RandomState(7)
    descs = synth_descriptors(n, R, S, seed=1011, revisit_frac=0.02)
    base = descs[n - 1].copy()
    k = 20
    descs[k + 0] = np.tile(base[:, :1], (1, S))                               # flat sector key: every shift ties
    descs[k + 1] = np.tile(base[:, :S // 2], (1, 2))                          # period S/2: two exact ties
    descs[k + 2] = np.tile(base[:, :2], (1, S // 2))                          # period 2
    descs[k + 3] = np.tile(base[:, :S // 2], (1, 2)); descs[k + 3][0, S // 2 + 1] += np.float32(1e-6)     # ... broken in the last bits
    descs[k + 4] = np.tile(base[:, :S // 2], (1, 2)); descs[k + 4][3, 7] *= np.float32(1.0 + 1e-7)
    descs[k + 5] = np.roll(base, 17, axis=1) * np.float32(1e18)               # huge: the filter's norm guard
    descs[k + 6] = np.roll(base, 33, axis=1) * np.float32(1e-18)              # tiny
    descs[k + 7] = np.roll(base, 5, axis=1); descs[k + 7][2, 9] = np.nan
    descs[k + 8] = np.roll(base, 5, axis=1); descs[k + 8][2, 9] = np.inf
    descs[k + 9] = 0.0
    for i in range(10):                                                       # near copies of the query at every kind of shift
        d = np.roll(base, int(rs.randint(0, S)), axis=1)
        descs[k + 10 + i] = np.clip(d + np.float32(10.0 ** -(i % 5 + 3)) * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    eng.alignment_stats(reset=True)
    _check(eng, db, n - 1, 0, n - 100)
    pairs, fallbacks = eng.alignment_stats(reset=True)
    # the flat / periodic / huge / non-finite / all-zero rows must have gone to the exact evaluation, ordinary ones must not
    assert pairs >= n - 100 and 6 <= fallbacks <= 40, (pairs, fallbacks)
    for q in (k + 0, k + 1, k + 3, k + 5, k + 7, k + 9, n - 2):               # the stress rows as queries as well
        _check(eng, db, q, 0, n - 100 if q >= n - 100 else k + 20)
    pairs, fallbacks = eng.alignment_stats()
    assert fallbacks >= 3 * (k + 20)                                          # a flat / periodic / all-zero query ties against everything
    qs = np.array([n - 1, k + 1, k + 3, k + 0, n - 2, k + 5, k + 7, n - 3], dtype=np.int32)
    nn, sh, dd = eng.detect_full_stream(qs, 0, np.full(len(qs), 300, np.int32), 4, 2)
    for i, q in enumerate(qs):
        o = db.detect_full_range(int(q), 0, 300) if hasattr(db, "detect_full_range") else None
        if o is None:
            d_ref, s_ref = db.distance_batch(int(q), cand=np.arange(0, 300, dtype=np.int32))
            ok = d_ref < 1e7
            if not ok.any():
                assert nn[i] == -1
                continue
            b = int(np.flatnonzero(ok)[np.argmin(d_ref[ok])])
            assert (nn[i], sh[i]) == (b, s_ref[b]) and dd[i].view(np.uint64) == d_ref[b].view(np.uint64), (q, nn[i], b)
    eng.close()


@pytest.mark.parametrize("R,S,n", [(64, 120, 777), (80, 180, 401)])
def test_random_ranges_against_the_checker(R, S, n):
    """Ranges of every size around the kernels' granularity (groups of 16 keyframes, four waves, 64-entry walks of the
    selection), starting anywhere, queries anywhere: the winner of the screened pass is the checker's, bit for bit."""
    descs = synth_descriptors(n, R, S, seed=4242, revisit_frac=0.08)
    rs = np.random.RandomState(99)
    descs[5] = 0.0; descs[6][:, ::2] = 0.0; descs[7][1, 1] = np.nan
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    sizes = [1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    cases = []
    for sz in sizes:
        for _ in range(2):
            lo = int(rs.randint(0, n - sz))
            cases.append((int(rs.randint(0, n)), lo, lo + sz))
    for _ in range(20):
        lo = int(rs.randint(0, n - 1)); hi = int(rs.randint(lo + 1, n + 1))
        cases.append((int(rs.randint(0, n)), lo, hi))
    qs = np.array([c[0] for c in cases], np.int32); los = np.array([c[1] for c in cases], np.int32); his = np.array([c[2] for c in cases], np.int32)
    nn, sh, dd = eng.detect_full_stream(qs, los, his, 4, 2)
    for i, (q, lo, hi) in enumerate(cases):
        d_ref, s_ref = db.distance_batch(q, cand=np.arange(lo, hi, dtype=np.int32))
        ok = d_ref < 1e7
        one = eng.detect_full_range(q, lo, hi)
        if not ok.any():
            assert nn[i] == -1 and one[0] == -1, (q, lo, hi)
            continue
        b = int(np.flatnonzero(ok)[np.argmin(d_ref[ok])])
        assert (nn[i], sh[i]) == (lo + b, s_ref[b]) and dd[i].view(np.uint64) == d_ref[b].view(np.uint64), (q, lo, hi, nn[i], lo + b)
        assert one[0] == lo + b and one[1] == s_ref[b] and np.float64(one[2]).view(np.uint64) == d_ref[b].view(np.uint64)
    eng.close()


@pytest.mark.parametrize("env", [{"SCL_SCREEN_FORM": "1"}, {"SCL_SCREEN_V2_MIN": "1"}])
def test_both_forms_of_the_products_meet_the_bound_for_every_batch_size(env):
    """The form of the screening products is chosen per batch (second form from four scans on, two on 80x180) and the switches
    are read once per process: the bound / survivor / winner checks of this file -- single-scan probes and streams -- once with the
    first form on every batch and once with the second form on every batch, in a child interpreter"""
    import os, subprocess, sys
    here = os.path.abspath(__file__)
    out = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-m", "gpu", "-x", "-k",
                          "bench_database or adversarial or random_ranges or 80x180"], env=dict(os.environ, **env),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
