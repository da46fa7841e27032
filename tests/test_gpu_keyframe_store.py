"""On-device keyframe store (robots[id].keyFrameArray, DM.h:86) through the C ABI.
Bar: submaps built from stored clouds are bit-identical to the CPU restatement of
loopFindNearKeyframes (DM.h:1163-1186) and to the host-cloud entry point; the fused
submap + ICP call returns exactly what scl_icp_align returns on those submaps."""
import numpy as np
import pytest

import oracle_icp_binding as oi
from scl_slam_amd import ScanContextEngine, SclError
from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
from test_oracle_icp_kat import moved_copy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    """12 keyframes of one robot along a short trajectory, poses as 4x4 matrices"""
    e = ScanContextEngine()
    rs = np.random.RandomState(11)
    clouds, poses = [], []
    for k in range(12):
        clouds.append(synth_structured_cloud(3000 + 150 * k, seed=70 + k))
        poses.append(e.pose_to_matrix(0.8 * k, 0.1 * k, 0.02 * k, *(rs.uniform(-0.03, 0.03, 3))))
        e.keyframe_put(0, k, clouds[-1])
    yield e, clouds, poses
    e.close()


def _window(poses, key, sn):
    ident = np.eye(4, dtype=np.float32)
    return [poses[k] if 0 <= k < len(poses) else ident for k in range(key - sn, key + sn + 1)]


def _oracle_submap(clouds, poses, key, sn, leaf):
    ks = [k for k in range(key - sn, key + sn + 1) if 0 <= k < len(clouds)]
    merged = np.concatenate([oi.transform(clouds[k], poses[k]) for k in ks])
    return oi.voxel_grid(merged, leaf)


def test_store_round_trip(world):
    e, clouds, _ = world
    assert e.keyframe_count(0) == 12 and e.keyframe_count(3) == 0
    for k in (0, 5, 11):
        assert np.array_equal(e.keyframe_get(0, k).view(np.uint32), clouds[k].view(np.uint32))


@pytest.mark.parametrize("key,sn", [(6, 3), (0, 3), (11, 2), (4, 0), (5, 25)])
def test_submap_from_store_bit_exact(world, key, sn):
    e, clouds, poses = world
    cap = sum(c.shape[0] for c in clouds)
    g = e.submap_from_store(0, key, sn, _window(poses, key, sn), 0.4, cap)
    o = _oracle_submap(clouds, poses, key, sn, 0.4)
    assert g.shape == o.shape and np.array_equal(g.view(np.uint32), o.view(np.uint32))
    ks = [k for k in range(key - sn, key + sn + 1) if 0 <= k < 12]
    h = e.assemble_submap([clouds[k] for k in ks], [poses[k] for k in ks], 0.4)
    assert np.array_equal(g.view(np.uint32), h.view(np.uint32))


def test_replace_keyframe_and_errors(world):
    e, clouds, poses = world
    bigger = synth_structured_cloud(9000, seed=5)
    e.keyframe_put(1, 0, clouds[0])
    e.keyframe_put(1, 0, bigger)                                   # larger cloud: re-allocated
    assert np.array_equal(e.keyframe_get(1, 0).view(np.uint32), bigger.view(np.uint32))
    e.keyframe_put(1, 0, clouds[1])                                # smaller: in place
    assert np.array_equal(e.keyframe_get(1, 0).view(np.uint32), clouds[1].view(np.uint32))
    e.keyframe_put(1, 2, clouds[2])                                # index 1 of robot 1 never stored
    with pytest.raises(SclError):
        e.submap_from_store(1, 1, 1, _window(poses, 1, 1), 0.4, 50000)
    with pytest.raises(SclError):
        e.keyframe_put(1, 3, np.zeros((10, 4), np.float32))        # other record stride than the store's
    # a robot with nothing stored: empty submap, as loopFindNearKeyframes on an empty keyFrameArray
    assert e.submap_from_store(7, 0, 2, _window(poses, 0, 2), 0.4, 10).shape[0] == 0


def test_loop_icp_from_store_equals_icp_on_submaps():
    e = ScanContextEngine()
    try:
        # a revisit: keyframe 9 sees the place of keyframes 2..4 from a slightly wrong pose estimate
        base = synth_structured_cloud(24000, seed=3)
        ident = np.eye(4, dtype=np.float32)
        poses = [ident.copy() for _ in range(10)]
        for k in range(9):
            e.keyframe_put(0, k, base[k::3][:6000].copy())
        drift = rigid_transform(0.01, -0.015, 0.04, 0.25, -0.2, 0.05)
        cur = moved_copy(base, drift, keep_every=4, noise=0.005)
        e.keyframe_put(0, 9, cur)
        sn, leaf = 2, 0.3
        T, fit, conv, it, ns, nt = e.loop_icp_from_store(0, 9, poses[9], 3, sn, _window(poses, 3, sn), leaf)
        src = e.submap_from_store(0, 9, 0, [poses[9]], leaf, cur.shape[0])
        tgt = e.submap_from_store(0, 3, sn, _window(poses, 3, sn), leaf, 40000)
        assert (ns, nt) == (src.shape[0], tgt.shape[0]) and ns >= 300 and nt >= 1000
        T2, fit2, conv2, it2 = e.icp_align(src, tgt)
        assert conv and conv2 and it == it2
        assert np.array_equal(T.view(np.uint32), T2.view(np.uint32)) and fit == fit2
        To, fito, convo, ito = oi.icp_align(src, tgt)
        assert convo and np.abs(T - To).max() < 1e-5 and abs(fit - fito) < 1e-5
        # size gate of DM.h:1108: too few points -> no alignment attempted
        T3, fit3, conv3, it3, ns3, nt3 = e.loop_icp_from_store(0, 9, poses[9], 3, sn, _window(poses, 3, sn), leaf,
                                                             min_src_points=10 ** 6)
        assert not conv3 and it3 == 0 and np.array_equal(T3, ident)
    finally:
        e.close()


def test_service_verification_from_store_equals_stepwise():
    """geometricVerificationService with the submap from the store == voxel filter + submap + scl_geometric_verification"""
    e = ScanContextEngine()
    try:
        base = synth_structured_cloud(30000, seed=13)
        ident = np.eye(4, dtype=np.float32)
        for k in range(7):
            e.keyframe_put(0, k, base[k::3][:8000].copy())
        T = rigid_transform(0.02, -0.01, 0.03, 0.3, -0.2, 0.05)
        received = moved_copy(base, T, keep_every=3, noise=0.01)
        received[::9, :3] += 2.5                                      # outliers for the RANSAC stage
        sn = 2
        poses = [ident] * (2 * sn + 1)
        Tg, ok, ns, nt, nc, ni = e.geometric_verification_from_store(received, 0.2, 0, 3, sn, poses, 0.3, 1000, 0.25, 0.45, 7)
        src = e.voxel_grid(received, 0.2)
        tgt = e.submap_from_store(0, 3, sn, poses, 0.3, 60000)
        Ts, oks, ncs, nis = e.geometric_verification(src, tgt, 1000, 0.25, 0.45, 7)
        assert (ns, nt) == (src.shape[0], tgt.shape[0]) and (ok, nc, ni) == (oks, ncs, nis)
        assert np.array_equal(Tg.view(np.uint32), Ts.view(np.uint32)) and ni > 100
        # size gate (DM.h:1204)
        T2, ok2, *_ = e.geometric_verification_from_store(received, 0.2, 0, 3, sn, poses, 0.3, min_src_points=10 ** 7)
        assert not ok2 and np.array_equal(T2, ident)
    finally:
        e.close()


def test_loop_icp_batch_from_store_equals_one_by_one():
    """the fused ICP loops of a scan's candidates (BASELINE configs[2]) return what scl_loop_icp_from_store returns per candidate"""
    e = ScanContextEngine()
    try:
        base = synth_structured_cloud(24000, seed=3)
        ident = np.eye(4, dtype=np.float32)
        for k in range(12):
            e.keyframe_put(0, k, (base if k < 9 else synth_structured_cloud(24000, seed=40 + k))[k % 3::3][:6000].copy())
        drift = rigid_transform(0.01, -0.015, 0.04, 0.25, -0.2, 0.05)
        e.keyframe_put(0, 12, moved_copy(base, drift, keep_every=4, noise=0.005))
        sn, leaf = 1, 0.3
        keys = [3, 6, 10, 1, 11]                                 # matching places and unrelated ones
        poses = np.stack([np.stack(_window([ident] * 13, k, sn)) for k in keys])
        pp = e.icp_default_params(); pp.max_iterations = 30
        Tb, fb, cb, ib, ns, ntb = e.loop_icp_batch_from_store(0, 12, ident, keys, sn, poses, leaf, pp)
        for c, k in enumerate(keys):
            T1, f1, c1, i1, ns1, nt1 = e.loop_icp_from_store(0, 12, ident, k, sn, _window([ident] * 13, k, sn), leaf, pp)
            assert (ns, ntb[c]) == (ns1, nt1) and cb[c] == c1 and ib[c] == i1
            assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == np.float32(f1)
        assert cb.all()
        # point to plane: the candidates' normals by the batched launch on the side stream == one target at a time
        pq = e.icp_default_params(); pq.max_iterations = 30; pq.estimator = 1; pq.normal_radius = 1.5
        Tb, fb, cb, ib, ns, ntb = e.loop_icp_batch_from_store(0, 12, ident, keys, sn, poses, leaf, pq)
        for c, k in enumerate(keys):
            T1, f1, c1, i1, ns1, nt1 = e.loop_icp_from_store(0, 12, ident, k, sn, _window([ident] * 13, k, sn), leaf, pq)
            assert (ns, ntb[c]) == (ns1, nt1) and cb[c] == c1 and ib[c] == i1
            assert np.array_equal(Tb[c].view(np.uint32), T1.view(np.uint32)) and fb[c] == np.float32(f1)
        # size gate: nothing attempted
        T0, f0, c0, i0, _, _ = e.loop_icp_batch_from_store(0, 12, ident, keys[:2], sn, poses[:2], leaf, pp, min_tgt_points=10 ** 7)
        assert not c0.any() and not i0.any() and np.array_equal(T0[1], ident)
    finally:
        e.close()
