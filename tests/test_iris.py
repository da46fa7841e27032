"""LiDAR-Iris building blocks (SURVEY.md 8(f)-4 part 2; reference include/descriptor.h:462-1302).
CPU: known answers of the restatement.  GPU: image, row key, templates and Hamming matching bit-identical to it."""
import math

import numpy as np
import pytest

import oracle_iris_binding as oi
from scl_slam_amd.synth import synth_scan


def _cloud(pts):
    c = np.zeros((len(pts), 8), np.float32)
    if len(pts):
        c[:, :3] = np.asarray(pts, np.float32)
    return c


def test_iris_image_known_answers():
    cfg = oi.config()
    # (x, y, z) -> distance bin floor(hypot), yaw bin floor(atan2(y, x) deg + 180 + 0.5), elevation bit floor((atan2(z, dis) deg + 24.9) / 4)
    img, key = oi.make_image(cfg, _cloud([[10.0, 0.0, 0.0]]))
    assert img[10, 180] == 1 << 6 and np.count_nonzero(img) == 1 and not key.any()       # elevation 0 deg -> (0 + 24.9) / 4 = 6.2 -> bit 6; z = 0 is not > 0
    img, key = oi.make_image(cfg, _cloud([[0.0, 5.5, 1.0]]))
    assert img[5, 270] == 1 << 7 and key[5] == np.float32(1.0) / np.float32(360)          # 90 deg + 180; atan2(1, 5.5) = 10.3 deg -> 8.8 -> clamp 7
    img, _ = oi.make_image(cfg, _cloud([[-3.0, -0.001, -1.2]]))
    assert img[3, 0] != 0                                                                 # yaw just above -180 deg -> bin 0
    img, _ = oi.make_image(cfg, _cloud([[200.0, 0.0, -50.0]]))
    assert img[79, 180] == 1 << 2                                                         # distance clamps to rows - 1; atan2(-50, 200) = -14.04 deg -> (10.86) / 4 -> bit 2
    img, _ = oi.make_image(oi.config(nscan=32), _cloud([[10.0, 0.0, 0.0]]))
    assert not img.any()                                                                  # only 16- and 64-beam branches exist (D.h:538, 560)
    # bits accumulate, heights take the maximum
    img, key = oi.make_image(cfg, _cloud([[10.0, 0.0, 0.0], [10.2, 0.01, 3.0], [10.4, 0.0, 2.0]]))
    assert img[10, 180] == (1 << 6) | (1 << 7) and key[10] == np.float32(3.0) / np.float32(360)


def test_iris_templates_and_hamming_properties():
    cfg = oi.config(rows=8, cols=36, nscale=2, min_wavelength=6)
    rs = np.random.RandomState(3)
    a = (rs.random_sample((8, 36)) < 0.4).astype(np.uint8) * rs.randint(1, 255, size=(8, 36)).astype(np.uint8)
    Ta, Ma = oi.encode(cfg, a)
    assert Ta.shape == (32, 36) and set(np.unique(Ta)) <= {0, 255} and set(np.unique(Ma)) <= {0, 255}
    resp = oi.responses(cfg, a)
    # the filters have no DC and no negative frequencies: rows of an all-zero image answer 0 (masked), and the response is analytic
    z = np.zeros_like(a)
    Tz, Mz = oi.encode(cfg, z)
    assert not Tz.any() and Mz.all()
    assert abs(resp[..., 0].sum(axis=2)).max() < 1e-9 * max(1.0, abs(resp).max())        # no DC
    # identical templates: distance 0 at shift 0; a column-rolled copy is found at its shift
    d, b = oi.hamming(cfg, Ta, Ma, Ta, Ma, 0)
    assert d == 0.0 and b == 0
    for sh in (1, 5, 17, 35):
        Tb, Mb = oi.encode(cfg, np.roll(a, sh, axis=1))
        d, b = oi.hamming_all(cfg, Ta, Ma, Tb, Mb)
        assert d == 0.0 and b == sh                                                       # circShift(T1, 0, sh) == T2 (D.h:938)
        d, b = oi.hamming(cfg, Ta, Ma, Tb, Mb, sh + 1)
        assert d == 0.0 and b == sh
        d, b = oi.hamming(cfg, Ta, Ma, Tb, Mb, sh + 3)
        assert b != sh and d > 0.0                                                        # outside the +-2 window
    # everything masked -> NaN, bias -1 (D.h:934-951)
    d, b = oi.hamming(cfg, Tz, Mz, Ta, Ma, 0)
    assert math.isnan(d) and b == -1


@pytest.mark.gpu
def test_iris_on_the_gpu_equals_the_restatement():
    from scl_slam_amd.iris import IrisEngine
    eng = IrisEngine(robot_num=3)
    cfg = oi.config()
    rs = np.random.RandomState(5)
    clouds = [synth_scan(60000, seed=70 + k, max_range=85.0) for k in range(5)]
    clouds[1][:, 2] += 1.0                                        # more positive heights for the row key
    # a yaw-rotated revisit of scan 0
    th = np.deg2rad(37.0); c0 = clouds[0].copy()
    c0[:, 0], c0[:, 1] = (math.cos(th) * clouds[0][:, 0] - math.sin(th) * clouds[0][:, 1]), (math.sin(th) * clouds[0][:, 0] + math.cos(th) * clouds[0][:, 1])
    clouds.append(c0)
    edge = _cloud([[0, 0, 0], [0, 0, 5], [1e-30, 0, 1], [-5, 0, 2], [-5, -0.0, 2], [3, 4, np.nan], [np.inf, 1, 1], [79.999, 0, 0.1], [80.0, 0, 0.1], [1e6, -1e6, 3]])
    clouds.append(edge)
    feats = []
    for k, cl in enumerate(clouds):
        img_o, key_o = oi.make_image(cfg, cl)
        img_g, key_g = eng.make_image(cl)
        assert np.array_equal(img_g, img_o), k
        assert np.array_equal(key_g.view(np.uint32), key_o.view(np.uint32)), k
        vals = eng.make_and_save(cl, 1, 10 + k)
        assert np.array_equal(vals[:80 * 360], img_o.reshape(-1).astype(np.float32)) and np.array_equal(vals[80 * 360:].view(np.uint32), key_o.view(np.uint32))
        T_o, M_o = oi.encode(cfg, img_o)
        T_g, M_g = eng.get_feature(k)
        assert np.array_equal(T_g, T_o) and np.array_equal(M_g, M_o), k
        feats.append((T_o, M_o))
    assert eng.get_size() == len(clouds) and eng.get_index(3) == (1, 13)
    img_back, key_back = eng.get_image(2)
    assert np.array_equal(img_back, oi.make_image(cfg, clouds[2])[0])
    # an image stored from the wire gets the same templates
    eng.save_image(*oi.make_image(cfg, clouds[4]), robot=2, index=99)
    assert np.array_equal(eng.get_feature(len(clouds))[0], feats[4][0])
    # Hamming matching: windows around given estimates, and every shift
    n = len(clouds)
    for k1 in (0, 5, 6):
        cand = np.array([c for c in range(n) if c != k1], np.int32)
        scales = rs.randint(-400, 400, size=cand.size).astype(np.int32)
        scales[0] = 37 if k1 == 0 else scales[0]
        d_g, b_g = eng.hamming_batch(k1, cand, scales)
        for i, c in enumerate(cand):
            d_o, b_o = oi.hamming(cfg, *feats[k1], *feats[c], int(scales[i]))
            assert (np.float32(d_g[i]).view(np.uint32) == np.float32(d_o).view(np.uint32) or (math.isnan(d_g[i]) and math.isnan(d_o))) and b_g[i] == b_o
        d_g, b_g = eng.hamming_all_shifts(k1, cand)
        for i, c in enumerate(cand):
            d_o, b_o = oi.hamming_all(cfg, *feats[k1], *feats[c])
            assert (np.float32(d_g[i]).view(np.uint32) == np.float32(d_o).view(np.uint32) or (math.isnan(d_g[i]) and math.isnan(d_o))) and b_g[i] == b_o
    # the rotated revisit is the nearest keyframe of scan 0, at the rotation's column shift
    d_all, b_all = eng.hamming_all_shifts(0, np.array([1, 2, 3, 4, 5], np.int32))
    assert int(np.argmin(d_all)) == 4 and abs(int(b_all[4]) - 37) <= 1 and d_all[4] < 0.5 * np.delete(d_all, 4).min()
    eng.close()


def test_restated_atan2f_equals_libm_on_a_sample():
    """D.h:547-549 call std::atan2(float, float) = the platform's atan2f; the checker (and, bit for bit, the device: the image tests
    below compare the two) restates glibc's.  `make -C oracle atan2f-check` compares it with libm on 4e9 pairs (0 differences); here:
    every pair of special values and 60 000 random pairs, without a GPU."""
    import ctypes
    import oracle_binding as ob
    L = ob.load()
    L.iriso_atan2f.restype = ctypes.c_float; L.iriso_atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    libm = ctypes.CDLL("libm.so.6"); libm.atan2f.restype = ctypes.c_float; libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    special = np.array([0, 0x80000000, 1, 0x80000001, 0x007fffff, 0x00800000, 0x3f800000, 0xbf800000, 0x7f800000, 0xff800000, 0x7fc00000,
                        0x7f7fffff, 0xff7fffff, 0x3f000000, 0x40490fdb, 0x1e3ce508], dtype=np.uint32).view(np.float32)
    rs = np.random.RandomState(11)
    rnd = rs.randint(0, 2 ** 32, size=(20000, 2), dtype=np.uint64).astype(np.uint32).view(np.float32)
    coords = (rs.uniform(-100, 100, size=(40000, 2))).astype(np.float32)
    pairs = [(float(a), float(b)) for a in special for b in special] + [tuple(map(float, p)) for p in rnd] + [tuple(map(float, p)) for p in coords]
    for y, x in pairs:
        a, b = L.iriso_atan2f(y, x), libm.atan2f(y, x)
        assert (a != a and b != b) or np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (y, x, a, b)
