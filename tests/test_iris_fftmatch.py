"""LiDAR-Iris: the shift estimate in front of the matching -- logPolarFFTTemplateMatch, descriptor.h:793-925 -- and compare(),
descriptor.h:964-1024.  OpenCV is absent from the image, so the estimate is OpenCV's published algorithms restated
(oracle/iris_oracle.c: iriso_fft_match; PARITY UNPINNED against the reference's binaries).  CPU: known answers of the
restatement.  GPU: scl_iris_fft_match / scl_iris_compare equal it bit for bit (float bit patterns, integer shifts)."""
import numpy as np
import pytest

import oracle_iris_binding as oi
from scl_slam_amd.synth import synth_scan

ROWS, COLS = 80, 360


def _iris_like(seed, rows=ROWS, cols=COLS):
    rs = np.random.RandomState(seed)
    img = np.zeros((rows, cols), np.uint8)
    for _ in range(40):                                        # blobs of set elevation bits, like walls seen over a few bins
        r, c = rs.randint(0, rows), rs.randint(0, cols)
        h, w = rs.randint(1, 6), rs.randint(2, 25)
        img[r:r + h, np.arange(c, c + w) % cols] |= np.uint8(rs.randint(1, 256))
    return img


def test_estimate_of_a_turned_copy_is_the_turn():
    """im1 = im0 with its columns rolled by s: the translation stage finds exactly -s (the peak is a delta: the centroid is exact),
    the log-polar stage rotation 0 / scale 1 -- so compare()'s first window is centred on the shift that lines the templates up."""
    base = _iris_like(1)
    for s in (0, 1, 5, -7, 37, 100, 179, -180):
        cx, ok, dbg = oi.fft_match(ROWS, COLS, base, np.roll(base, s, axis=1))
        est = int(np.float32(cx) - np.float32(COLS // 2))
        # (a half turn: +180 and -180 are the same roll; the peak sits at column 0 of the swapped array: +180)
        assert ok == 1 and (est == -s or (abs(s) == 180 and abs(est) == 180)), (s, cx, dbg)
        assert abs(dbg[2]) < 1e-5 and abs(dbg[3] - 1.0) < 1e-6 and abs(dbg[5]) < 1e-6
    # a small geometry goes through the same code (the tests of the plugin layer use the full size)
    small = _iris_like(2, 16, 72)
    cx, ok, _ = oi.fft_match(16, 72, small, np.roll(small, 9, axis=1))
    assert ok == 1 and int(np.float32(cx) - np.float32(36)) == -9
    assert oi.fft_match(15, 72, small[:15], small[:15])[1] == -1           # odd sizes: not restated


def test_compare_windows_and_match_num():
    """compare() (D.h:964-1024) on a pair whose second image is the first turned by s: distance 0 at bias == -s (first pass) or
    (bias2 + 180) % 360 == -s mod 360 (second pass, the candidate turned by 180 columns); match_num picks the passes."""
    cfg = oi.config()
    a = _iris_like(3)
    Ta, Ma = oi.encode(cfg, a)
    for s in (0, 12, -33, 170):
        b = np.roll(a, s, axis=1)
        Tb, Mb = oi.encode(cfg, b)
        # img1 = b (the query, whose templates are shifted), img2 = a
        d0, b0, sh0 = oi.compare(cfg, 0, b, Tb, Mb, a, Ta, Ma)
        d1, b1, sh1 = oi.compare(cfg, 1, b, Tb, Mb, a, Ta, Ma)
        d2, b2, sh2 = oi.compare(cfg, 2, b, Tb, Mb, a, Ta, Ma)
        assert d0 == 0.0 and b0 == -s and sh0[0] == -s and sh0[1] == -2 ** 31
        assert d1 == 0.0 and b1 % 360 == (-s) % 360 and sh1[0] == -2 ** 31
        assert d2 == 0.0 and b2 % 360 == (-s) % 360 and sh2[0] == -s
        # D.h:986: the first pass wins only with a strictly smaller distance -- on a tie the second pass's (bias2 + 180) % 360
        assert b2 == (sh2[1] + 180 + (b1 - 180 - sh1[1])) % 360 or b2 == b1
    # an estimate two columns off still finds the exact alignment (the window is +-2), three columns off does not
    Tb, Mb = oi.encode(cfg, np.roll(a, 20, axis=1))
    assert oi.hamming(cfg, Tb, Mb, Ta, Ma, -18) == (0.0, -20) and oi.hamming(cfg, Tb, Mb, Ta, Ma, -17)[0] > 0.0


@pytest.mark.gpu
def test_fft_match_and_compare_on_the_gpu_equal_the_restatement():
    from scl_slam_amd.iris import IrisEngine
    cfg = oi.config()
    imgs = [_iris_like(10 + k) for k in range(3)]
    imgs.append(np.roll(imgs[0], 77, axis=1))
    imgs += [oi.make_image(cfg, synth_scan(20000, seed=40 + k, max_range=85.0))[0] for k in range(3)]       # images of real-shaped scans
    noisy = imgs[5].copy(); noisy[::7, ::5] ^= 0x10
    imgs.append(np.roll(noisy, -41, axis=1))
    for match_num in (2, 0, 1):
        eng = IrisEngine(match_num=match_num)
        for k, im in enumerate(imgs):
            eng.save_image(im, np.zeros(ROWS, np.float32) + k, 0, k)
        n = len(imgs)
        # compare() is checked on the engine's own templates (the templates themselves are tests/test_iris.py's subject; on these
        # drawn images a few imaginary responses cancel exactly and their sign bit is decided by rounding)
        feats = [eng.get_feature(k) for k in range(n)]
        if match_num == 2:
            for k0, roll, k1 in ((0, 0, 3), (3, 180, 0), (1, 0, 2), (5, 0, 7), (7, 180, 5), (4, 0, 6), (2, 180, 2)):
                cx_g, ok_g = eng.fft_match(k0, roll, k1)
                cx_o, ok_o, dbg = oi.fft_match(ROWS, COLS, np.roll(imgs[k0], roll, axis=1), imgs[k1])
                assert np.float32(cx_g).view(np.uint32) == np.float32(cx_o).view(np.uint32) and bool(ok_o == 1) == ok_g, (k0, roll, k1, cx_g, cx_o, dbg)
        for key1 in (3, 7, 4):
            cand = [k for k in range(n) if k != key1]
            d_g, b_g = eng.compare(key1, cand)
            for i, k2 in enumerate(cand):
                d_o, b_o, _ = oi.compare(cfg, match_num, imgs[key1], feats[key1][0], feats[key1][1], imgs[k2], feats[k2][0], feats[k2][1])
                assert b_g[i] == b_o and (np.float32(d_g[i]).view(np.uint32) == np.float32(d_o).view(np.uint32) or (np.isnan(d_g[i]) and np.isnan(d_o))), (match_num, key1, k2, d_g[i], d_o, b_g[i], b_o)
        eng.close()
