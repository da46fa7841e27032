"""The plugin layer of LiDAR-Iris (lidar_iris_descriptor's six virtuals, reference include/descriptor.h:1026-1271).
CPU: the restatement (oracle/iris_plugin_oracle.py) on hand-checkable scenarios.  GPU: scl_iris_* through the C ABI,
identical to it -- loop ids, shifts, float distances by bit pattern."""
import math

import numpy as np
import pytest

import oracle_binding as ob
import oracle_iris_binding as oi
from oracle.iris_plugin_oracle import IrisPluginOracle
from scl_slam_amd.synth import synth_scan

ROWS, COLS = 80, 360


def _moved(cloud, yaw_deg, dx, dy, seed):
    """a revisit: the same place seen again under another heading, a little off the first track, with range noise"""
    rs = np.random.RandomState(seed)
    th = math.radians(yaw_deg)
    out = cloud.copy()
    x, y = cloud[:, 0] - dx, cloud[:, 1] - dy
    out[:, 0] = math.cos(th) * x - math.sin(th) * y + 0.01 * rs.standard_normal(len(x))
    out[:, 1] = math.sin(th) * x + math.cos(th) * y + 0.01 * rs.standard_normal(len(x))
    return out


def _trajectory(n, n_points=12000, first_seed=300):
    return [synth_scan(n_points, seed=first_seed + k, max_range=85.0) for k in range(n)]


def _same(a, b):
    return (a[0], a[1]) == (b[0], b[1]) and (np.float32(a[2]).view(np.uint32) == np.float32(b[2]).view(np.uint32))


def test_plugin_restatement_finds_the_planted_revisit():
    kw = dict(num_exclude_recent=6, num_candidates=4)
    po = IrisPluginOracle(oi, ob, **kw)
    scans = _trajectory(14)
    scans[12] = _moved(scans[2], 41.0, 0.3, -0.2, 1)                      # keyframe 12 revisits keyframe 2
    wires = [po.make_and_save(s, 0, k) for k, s in enumerate(scans)]
    assert wires[0].shape == (ROWS * COLS + ROWS,) and po.get_size() == 14 and po.get_size(0) == 14 and po.get_index(5) == (0, 5)
    # D.h:1092: too few keyframes behind the exclusion window -> no search
    assert po.detect_intra(10) == (-1, 0.0, 10000000.0)
    loop, bias, dis = po.detect_intra(12)
    # the revisit's columns sit 41 further: circShift(T1, -41) lines it up with keyframe 2 -- compare() reports the window's own
    # shift (bias1, which can be negative) or (bias2 + 180) % 360 (D.h:986-997): the same turn modulo 360
    assert loop == 2 and dis < 0.32 and min((bias - (360 - 41)) % 360, (360 - 41 - bias) % 360) <= 1
    dis_windows = dis
    loop13, _, dis13 = po.detect_intra(13)                                 # an unrelated place: candidates compared, none accepted
    assert loop13 == -1 and 0.32 <= dis13 < 1.0
    # a yaw-only revisit in image space (columns rolled, same row key): libnabo's self-match rule (d2 <= FLT_EPSILON) drops it
    # from the candidates; with self matches allowed it is found, a quarter turn away, at distance 0
    for eps, expect in ((np.finfo(np.float32).eps, None), (0.0, 2)):
        p2 = IrisPluginOracle(oi, ob, knn_exclude_eps=float(eps), **kw)
        for k, s in enumerate(scans):
            if k == 12:
                img2, key2 = p2.features[0][2][0], p2.rowkeys[0][2]
                p2.save(np.roll(img2, 90, axis=1), key2, 0, 12)
            else:
                p2.make_and_save(s, 0, k)
        loop, bias, dis = p2.detect_intra(12)
        if expect is None:
            assert loop != 2
        else:
            assert loop == 2 and bias % 360 == 270.0 and dis == 0.0          # circShift(T of 12, 270) == T of 2 (bias1 = -90 or (bias2 + 180) % 360 = 270)
    # every column shift instead of compare()'s windows (shift_search = 1): the first minimum over [0, 360)
    p3 = IrisPluginOracle(oi, ob, shift_search=1, **kw)
    for k, s in enumerate(scans):
        p3.make_and_save(s, 0, k)
    loop, bias, dis3 = p3.detect_intra(12)
    assert loop == 2 and abs(bias - (360 - 41)) <= 1 and dis3 <= dis_windows + 1e-7      # a superset of the windows: never larger


def test_plugin_restatement_wire_and_inter_robot():
    kw = dict(num_exclude_recent=6, num_candidates=3, robot_num=3, this_id=0)
    scans = _trajectory(6, first_seed=400)
    remote = _trajectory(5, first_seed=500)
    remote[3] = _moved(scans[1], -23.0, 0.2, 0.1, 3)                      # robot 1's keyframe 3 sees robot 0's place 1
    po = IrisPluginOracle(oi, ob, wire_decode=1, **kw)
    sender = IrisPluginOracle(oi, ob, robot_num=3, this_id=1)
    for k, s in enumerate(scans):
        po.make_and_save(s, 0, k)
    assert po.detect_inter(1) == (-1, 0.0, 10000000.0)                     # D.h:1198: nothing received yet
    for k, s in enumerate(remote):
        po.save_from_wire(sender.make_and_save(s, 1, k), 1, k)
    assert po.get_size() == 11 and po.get_size(0) == 6 and po.get_size(1) == 5 and po.get_size(2) == 0 and po.get_index(9) == (1, 3)
    # with the emitted layout decoded as emitted, a received keyframe equals the sender's
    assert np.array_equal(po.features[1][3][0], sender.features[1][3][0]) and np.array_equal(po.features[1][3][1], sender.features[1][3][1])
    loop, bias, dis = po.detect_inter(9)                                   # received keyframe -> searched among this robot's
    assert loop == 1 and dis < 0.32 and min((bias - 23) % 360, (23 - bias) % 360) <= 1
    loop, bias, dis = po.detect_inter(1)                                   # own keyframe -> searched among the other robots'
    assert loop == 9 and dis < 0.32 and min((bias - (360 - 23)) % 360, (360 - 23 - bias) % 360) <= 1
    # the reference's own decoder (D.h:1035) shears the image: row r starts r + 1 columns late; the last row ends in the row key
    pr = IrisPluginOracle(oi, ob, wire_decode=0, **kw)
    w = sender.make_and_save(remote[0], 1, 99)
    pr.save_from_wire(w, 1, 0)
    img = w[:ROWS * COLS].reshape(ROWS, COLS).astype(np.uint8)
    got = pr.features[1][0][0]
    assert np.array_equal(got[0, :-1], img[0, 1:]) and got[0, -1] == img[1, 0]
    assert np.array_equal(got[5, :COLS - 6], img[5, 6:]) and np.array_equal(got[5, COLS - 6:], img[6, :6])
    assert np.array_equal(pr.rowkeys[1][0], w[ROWS * COLS:])


@pytest.mark.gpu
def test_plugin_on_the_gpu_equals_the_restatement():
    from scl_slam_amd.iris import IrisEngine
    for decode in (0, 1):
        kw = dict(num_exclude_recent=6, num_candidates=4, robot_num=3, this_id=0, wire_decode=decode)
        eng = IrisEngine(**kw)
        po = IrisPluginOracle(oi, ob, **kw)
        sender = IrisPluginOracle(oi, ob, robot_num=3, this_id=1)
        scans = _trajectory(16, n_points=20000)
        scans[12] = _moved(scans[2], 41.0, 0.3, -0.2, 1)
        scans[14] = _moved(scans[5], -120.0, -0.25, 0.15, 4)
        remote = _trajectory(7, n_points=20000, first_seed=500)
        remote[3] = _moved(scans[1], -23.0, 0.2, 0.1, 3)
        remote[6] = _moved(scans[9], 77.0, 0.1, 0.3, 5)
        # interleaved arrival: own keyframes built from clouds, the other robots' from the wire
        order = [(0, k) for k in range(8)] + [(1, k) for k in range(4)] + [(0, k) for k in range(8, 16)] + [(1, k) for k in range(4, 7)] + [(2, 0)]
        for robot, k in order:
            if robot == 0:
                w_g = eng.make_and_save(scans[k], 0, k)
                w_o = po.make_and_save(scans[k], 0, k)
                assert np.array_equal(w_g.view(np.uint32), w_o.view(np.uint32))
            else:
                w = sender.make_and_save(remote[k] if robot == 1 else scans[3], robot, k)
                if robot == 2:
                    w = w.copy(); w[-ROWS:] = [-300.7, -1.5, -0.4, 0.4, 255.9, 256.2, 1e12, -1e12, np.nan, np.inf] * (ROWS // 10)
                eng.save_from_wire(w, robot, k); po.save_from_wire(w, robot, k)
        n = len(order)
        assert eng.get_size() == po.get_size() == n
        for r in range(3):
            assert eng.get_size(r) == po.get_size(r)
        for key in range(n):
            assert eng.get_index(key) == po.get_index(key)
            robot, idx = po.get_index(key)
            local = po.local2global[robot].index(key)
            assert eng.local_to_global(robot, local) == key
            img_g, key_g = eng.get_image(key)
            assert np.array_equal(img_g, po.features[robot][local][0]) and np.array_equal(key_g.view(np.uint32), po.rowkeys[robot][local].view(np.uint32))
            T_g, M_g = eng.get_feature(key)
            assert np.array_equal(T_g, po.features[robot][local][1]) and np.array_equal(M_g, po.features[robot][local][2])
        hits = 0
        for cur in range(16):
            g, o = eng.detect_intra(cur), po.detect_intra(cur)
            assert _same(g, o), (decode, cur, g, o)
            hits += g[0] >= 0
        assert hits == 2 and eng.detect_intra(12)[0] == 2 and eng.detect_intra(14)[0] == 5
        found = {}
        for cur in range(n):
            g, o = eng.detect_inter(cur), po.detect_inter(cur)
            assert _same(g, o), (decode, cur, g, o)
            if g[0] >= 0:
                found[cur] = g[0]
        if decode == 1:                                                    # received images intact: the planted inter-robot revisits are found both ways
            k_r3, k_r6 = po.local2global[1][3], po.local2global[1][6]
            assert found.get(k_r3) == po.local2global[0][1] and found.get(k_r6) == po.local2global[0][9]
            assert found.get(po.local2global[0][1]) == k_r3
        with pytest.raises(RuntimeError):
            eng.detect_intra(16)
        with pytest.raises(RuntimeError):
            eng.save_from_wire(np.zeros(ROWS * COLS + ROWS, np.float32), 3, 0)
        eng.close()
