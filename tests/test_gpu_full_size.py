"""BASELINE configs[1] at full size (10 k keyframes, 64x120) on the GPU: size-independent properties plus
spot checks against the CPU checker (the checker needs ~0.15 ms per pair, so it scores a sample only)."""
import numpy as np
import pytest

import oracle_binding as ob
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu
R, S, N = 64, 120, 10000


@pytest.fixture(scope="module")
def world():
    descs = synth_descriptors(N, R, S, seed=2024)
    rs = np.random.RandomState(3)
    planted = {}
    for q in range(N - 40, N, 4):                                   # queries that revisit an old place, rotated
        j, sh = int(rs.randint(0, N - 200)), int(rs.randint(0, S))
        descs[q] = np.roll(descs[j], sh, axis=1)
        planted[q] = (j, sh)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 8)
    e.save_bulk(descs)
    yield e, descs, planted
    e.close()


def test_planted_revisits_are_found_with_their_rotation(world):
    e, descs, planted = world
    for q, (j, sh) in planted.items():
        nn, shift, d = e.detect_full_range(q, 0, N - 100)
        assert nn == j and shift == sh and d < 1e-12


def test_fused_pass_equals_per_pair_path_and_checker(world):
    e, descs, planted = world
    cfg = ob.make_config(R=R, S=S)
    rs = np.random.RandomState(5)
    for q in (N - 1, N - 2, N - 40):
        hi = N - 100
        nn, shift, d = e.detect_full_range(q, 0, hi)               # one fused launch: arg-min inside the kernel
        dist, sh = e.sc_distance_batch(q, n=hi)                    # per-pair outputs of the same kernel
        j = int(np.argmin(dist))                                   # first minimum = lowest index on ties
        assert (nn, shift) == (j, int(sh[j])) and np.float64(d).view(np.uint64) == dist[j:j + 1].view(np.uint64)[0]
        assert np.all(np.isfinite(dist)) and dist.min() >= 0.0 and dist.max() <= 2.0
        for c in [j] + [int(x) for x in rs.randint(0, hi, 24)]:     # checker on the winner and a random sample
            dc, sc = ob.distance(cfg, descs[q], descs[c], fast=True)
            assert sc == sh[c] and np.float64(dc).view(np.uint64) == dist[c:c + 1].view(np.uint64)[0]


def test_batched_launch_equals_single_launches(world):
    e, _, _ = world
    qs = [N - 1, N - 2, N - 3, N - 40]
    single = [e.detect_full_range(q, 0, N - 100) for q in qs]
    tickets = e.detect_full_submit_many(qs, 0, N - 100)
    assert [e.detect_full_collect(t) for t in tickets] == single


def test_ring_shift_of_the_query_shifts_the_answer(world):
    """circshift(query, s) against an unrotated database: the candidate must turn s sectors further (D.h:1559 shifts
    the candidate), so the best ring shift moves by +s (mod S); same distance"""
    e, descs, _ = world
    q = descs[N - 7]
    e.stage_query(q)
    nn0, sh0, d0 = e.detect_full_range(-1, 0, N - 100)
    for s in (1, 17, 60, 119):
        e.stage_query(np.roll(q, s, axis=1))
        nn, sh, d = e.detect_full_range(-1, 0, N - 100)
        assert nn == nn0 and sh == (sh0 + s) % S and abs(d - d0) < 1e-12


def test_distance_matrix_equals_the_13_shift_kernel_on_the_whole_database(world):
    """scl_sc_distance_matrix at BASELINE configs[1]'s size: alignment + screening + shift masks from the recorded rounding-error norms,
    then sc_matrix_kernel at the open shifts -- against scl_sc_distance_batch, the exact kernel that evaluates all 13 shifts of
    every pair (another program: one wave per pair, the candidate rotating under the staged scan).  Every entry of 23 rows x 9 900
    keyframes must agree bit for bit (23 = a group of 16, one of 7: odd, so a workgroup with one scan; ranges that are not a whole
    number of workgroups), and the checker confirms a sample."""
    e, descs, planted = world
    cfg = ob.make_config(R=R, S=S)
    rs = np.random.RandomState(9)
    qs = np.array([N - 1 - i for i in range(20)] + [17, 4242, N - 40], dtype=np.int32)
    for lo, hi in ((0, N - 100), (1234, 1234 + 4097)):
        dist, sh = e.sc_distance_matrix(qs, lo, hi)
        assert dist.shape == (len(qs), hi - lo)
        for r, q in enumerate(qs):
            d1, s1 = e.sc_distance_batch(int(q), cand=np.arange(lo, hi, dtype=np.int32))
            assert np.array_equal(dist[r].view(np.uint64), d1.view(np.uint64)) and np.array_equal(sh[r], s1), (q, lo, hi)
        for r in (0, 21):
            for c in rs.randint(lo, hi, 12):
                dc, sc = ob.distance(cfg, descs[qs[r]], descs[int(c)], fast=True)
                assert sc == sh[r][c - lo] and np.float64(dc).view(np.uint64) == dist[r][c - lo:c - lo + 1].view(np.uint64)[0]


def test_points_pipeline_at_full_size_agrees_three_ways(world):
    """BASELINE configs[1]'s per-incoming-scan path at its stated size -- 120 k-point scans into the 10 k-keyframe database -- through the
    three forms: scl_stream_from_points (host clouds, groups of 16), scl_stream_from_store (the same clouds resident in HBM) and the
    reference's own call pattern, scl_make_and_save + scl_detect_full_range per keyframe.  Size-independent properties: the three
    agree on every descriptor, winner, shift and fp64 distance bit for bit; a scan that is an earlier scan turned about z by whole
    sectors is found at that scan with that shift and distance ~0; descriptors of a sample equal the checker's."""
    from scl_slam_amd.synth import synth_scan
    _, descs, _ = world
    n_scans, npts, excl = 36, 120000, 100
    engines = [ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=excl, initial_capacity=N + 64) for _ in range(3)]
    for e in engines:
        e.save_bulk(descs)
    clouds = [np.ascontiguousarray(synth_scan(npts, seed=9100 + i, stride_floats=4)) for i in range(n_scans)]
    # scan 20 = scan 3 turned about z by 11 sectors (36 scans behind an exclusion window of 100: a scan never finds another scan of
    # the call, so the rotation is checked on the descriptors, below)
    a = np.deg2rad(11 * 360.0 / S)
    src = clouds[3].copy()
    x, y = src[:, 0].astype(np.float64), src[:, 1].astype(np.float64)
    src[:, 0] = (x * np.cos(a) - y * np.sin(a)).astype(np.float32); src[:, 1] = (x * np.sin(a) + y * np.cos(a)).astype(np.float32)
    clouds[20] = src
    nn, sh, dd, vals = engines[0].stream_from_points(clouds, want_values=True)
    for i, c in enumerate(clouds):
        engines[1].keyframe_put(0, i, c)
    nn1, sh1, dd1, vals1 = engines[1].stream_from_store(0, 0, n_scans, want_values=True)
    assert np.array_equal(vals.view(np.uint32), vals1.view(np.uint32))
    assert np.array_equal(nn, nn1) and np.array_equal(sh, sh1) and np.array_equal(dd.view(np.uint64), dd1.view(np.uint64))
    for i, c in enumerate(clouds):
        v = engines[2].make_and_save(c, 0, N + i)
        g = engines[2].detect_full_range(N + i, 0, N + i - excl)
        assert np.array_equal(v.view(np.uint32), vals[i].view(np.uint32)), i
        assert (int(nn[i]), int(sh[i])) == (g[0], g[1]) and np.float64(dd[i]).view(np.uint64) == np.float64(g[2]).view(np.uint64), i
    assert (nn >= 0).all() and (nn < N + n_scans - excl).all()
    # the rotation property (D.h:1376-1395 on a cloud turned about z): most cells of scan 20's descriptor are scan 3's, 11 sectors on
    d3, d20 = vals[3].reshape(R, S), vals[20].reshape(R, S)
    same = (np.roll(d3, 11, axis=1) == d20).mean()
    assert same > 0.9, same                                             # (points on bin edges move to a neighbouring cell under the float rotation)
    cfg = ob.make_config(R=R, S=S)
    for i in (0, 20, 35):
        assert np.array_equal(vals[i].view(np.uint32), ob.make_scancontext(cfg, clouds[i]).view(np.uint32)), i
    for e in engines:
        e.close()
