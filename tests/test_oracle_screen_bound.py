"""The finishing kernel's per-shift bound on the screened distance (scl_slam_amd/csrc/sc_screen.hip, sc_screen2_finish_compute;
the recorded error norms come from make_sc.hip's ingest_kernel), restated on the CPU.

The screening pass evaluates the shifted cosine distances of distanceBtnScanContext (D.h:1545-1566) on fp16 unit columns with fp32
accumulation.  Round 4: the margin around a screened distance no longer assumes fp16's worst-case rounding for every element
(kScreenEps = 1.5e-3) but takes the ACTUAL rounding-error norms of the two descriptors' unit columns, recorded at ingest:

    |d~_t - d_t|  <=  (E_q + E_k) * 1.002 / n_eff(t) + screen2_acc_eps,      E = sum over columns of |h_c - x_c / norm_c|_2

(never more than kScreenEps).  This file checks that bound on random, sparse, wide-range and near-subnormal descriptors under three
models of the fp32 accumulation (sequential round-to-nearest, sequential TRUNCATING additions, one rounding per 32-product step),
and that the mask rule built on it -- a shift stays open iff the lower end of its interval is not above the smallest upper end --
never drops the shift that holds the exact minimum.  No GPU: numpy float16 rounds to nearest even like v_cvt_f16_f32."""
import numpy as np
import pytest

R, S, W = 64, 120, 13
K_SCREEN_EPS = 1.5e-3
K_ACC_EPS = ((S // 4) * 32 + 16) * 2.0 ** -23 * 1.002 + 2.0e-6      # screen2_acc_eps<120>: chains of 960 terms, truncating additions


def unit_fp16(desc):
    """ingest_kernel: column norms in fp64 (ring order), reciprocal narrowed to fp32, x * iv in fp32, rounded to fp16;
    E as the kernel records it (an entry below fp16's normal range counts with the larger of its error and its value)."""
    x = desc.astype(np.float32)
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(axis=0))
    iv = np.where(nrm > 0, (1.0 / np.where(nrm > 0, nrm, 1.0)).astype(np.float32), np.float32(0))
    h = (x * iv[None, :]).astype(np.float32).astype(np.float16)
    u = np.where(nrm[None, :] > 0, x.astype(np.float64) / np.where(nrm > 0, nrm, 1.0)[None, :], 0.0)
    hd = h.astype(np.float64)
    d = np.abs(hd - u)
    d = np.where(np.abs(hd) < 6.103515625e-05, np.maximum(d, np.abs(u)), d)
    d = np.where(nrm[None, :] > 0, d, 0.0)
    E = np.float32(np.nextafter(np.float32(np.sqrt((d ** 2).sum(axis=0)).sum() * (1 + 1e-6) + 1e-12), np.float32(np.inf)))
    return h, nrm, E


def exact_distances(q, k, nq, nk, first):
    """D.h:1513-1536 per shift (fp64, extended precision for the sums: the bound is against the real value, the reference's own
    rounding is 1e-14)."""
    out, neff = [], []
    ql, kl = q.astype(np.longdouble), k.astype(np.longdouble)
    for t in range(W):
        s = (first + t) % S
        ks, nks = np.roll(kl, s, axis=1), np.roll(nk, s)
        ok = (nq > 0) & (nks > 0)
        cos = (ql * ks).sum(axis=0)[ok] / (nq[ok].astype(np.longdouble) * nks[ok].astype(np.longdouble))
        out.append(float(1 - cos.sum() / ok.sum()) if ok.sum() else np.inf)
        neff.append(int(ok.sum()))
    return np.array(out), np.array(neff)


def trunc32(x64):
    """fp64 -> fp32 rounding toward zero"""
    f = x64.astype(np.float32)
    over = np.abs(f.astype(np.float64)) > np.abs(x64)
    return np.where(over, np.nextafter(f, np.float32(0)), f).astype(np.float32)


def screened_distances(hq, hk, mq, mk, first, model):
    """sim[t] = sum over sectors and rings of the fp16 products, fp32 accumulation in chains like the second form's (four
    accumulators by sector mod 4, two ring halves, then the partials joined), d~ = 1 - sim / n_eff in fp32"""
    out = []
    q32, k32 = hq.astype(np.float32), hk.astype(np.float32)
    for t in range(W):
        s = (first + t) % S
        ks = np.roll(k32, s, axis=1)
        prod = q32.astype(np.float64) * ks.astype(np.float64)              # exact (11 x 11 bits)
        parts = []
        for half in range(2):
            for a in range(4):
                acc = np.float32(0)
                for c in range(a, S, 4):
                    p = prod[32 * half:32 * half + 32, c]
                    if model == "step":
                        acc = np.float32(np.float64(acc) + p.sum())
                    else:
                        for v in p:
                            acc = np.float32(np.float64(acc) + v) if model == "nearest" else trunc32(np.array(np.float64(acc) + v))[()]
                parts.append(acc)
        sim = np.float32(0)
        for p in parts:
            sim = np.float32(sim + p)
        ne = int(((mq > 0) & (np.roll(mk, s) > 0)).sum())
        out.append(np.float32(1) - sim / np.float32(ne) if ne else np.float32(np.inf))
    return np.array(out, np.float32)


def descriptors():
    rs = np.random.RandomState(17)
    out = []
    def smooth():
        r, c = np.meshgrid(np.arange(R), np.arange(S), indexing="ij")
        d = sum(rs.uniform(0.3, 2) * np.sin(rs.uniform(0, 0.3) * r + rs.uniform(0, 0.2) * c + rs.uniform(0, 6)) for _ in range(8))
        d = np.clip(d + 3, 0, 12).astype(np.float32)
        z = rs.randint(S); d[:, z:z + rs.randint(1, 30)] = 0
        return d
    for _ in range(4):
        a = smooth(); out.append((a, np.roll(a, rs.randint(S), axis=1) + rs.normal(0, 0.05, (R, S)).astype(np.float32) * (a > 0)))
    for _ in range(3):
        out.append((smooth(), smooth()))
    a = rs.uniform(0, 12, (R, S)).astype(np.float32); b = rs.uniform(0, 12, (R, S)).astype(np.float32)
    out.append((a * (rs.rand(R, S) < 0.2), b * (rs.rand(R, S) < 0.2)))                       # sparse
    out.append((a * 10.0 ** rs.randint(-4, 3, (R, S)).astype(np.float32), b))             # wide range inside the columns: elements near and below fp16's normal range
    c = np.full((R, S), 1e-3, np.float32); c[0, :] = 12.0
    out.append((c, a))                                                                      # every other element of a unit column is ~1e-5: subnormal in fp16
    out.append((a - 6.0, b - 6.0))                                                          # signed cells (negative heights are kept by the reference)
    return out


@pytest.mark.parametrize("model", ["nearest", "truncating", "step"])
def test_screened_distance_stays_inside_the_recorded_error_bound_and_the_mask_keeps_the_minimum(model):
    worst_ratio, worst_err, open_new, open_old, pairs = 0.0, 0.0, 0, 0, 0
    for q, k in descriptors():
        hq, nq, Eq = unit_fp16(q)
        hk, nk, Ek = unit_fp16(k)
        for first in (0, 37):
            d, neff = exact_distances(q, k, nq, nk, first)
            dt = screened_distances(hq, hk, nq, nk, first, model)
            e_pair = np.float32(Eq + Ek) * np.float32(1.002)
            eps = np.minimum(e_pair / np.maximum(neff, 1).astype(np.float32) + np.float32(K_ACC_EPS), np.float32(K_SCREEN_EPS))
            fin = neff > 0
            err = np.abs(dt[fin].astype(np.float64) - d[fin])
            assert np.all(err <= eps[fin]), (model, err.max(), eps[fin].min())
            worst_ratio = max(worst_ratio, float((err / eps[fin]).max())); worst_err = max(worst_err, float(err.max()))
            lo, hi = dt - eps, dt + eps
            mask = fin & (lo <= hi[fin].min())
            tmin = int(np.argmin(np.where(fin, d, np.inf)))
            assert mask[tmin], "the shift of the exact minimum was dropped"
            assert np.all(mask[fin & (d == d[tmin])]), "a shift that ties for the minimum was dropped"
            open_new += int(mask.sum()); open_old += int((fin & (dt <= dt[fin].min() + 2 * K_SCREEN_EPS)).sum()); pairs += 1
    assert worst_ratio < 1.0
    # the point of the change: fewer shifts stay open than with the worst-case margin
    assert open_new <= open_old
    print(f"{model}: worst |d~ - d| = {worst_err:.2e}, worst error / bound = {worst_ratio:.2f}, open shifts per pair {open_new / pairs:.2f} (worst-case margin: {open_old / pairs:.2f})")
