"""The alignment kernel's filter (scl_slam_amd/csrc/sc_screen.hip, sc_align_role), restated on the CPU: the circular
correlation of two sector keys in fp32, in the order the matrix-core evaluation takes (blocks of 16 sectors, four steps of
K = 4 per block) under three models of how a step accumulates (fused multiply-adds one after the other, products rounded on
their own, one rounding per step).  For every model the error against the exact correlation must stay below HALF the
margin the kernel demands between the best and the second-best shift (4 * 4.07e-6 |vq| |vk|) -- then a shift the filter
accepts is the arg-max of the exact correlation, i.e. the reference's arg-min (fastAlignUsingVkey, D.h:1491-1511), which the
last assertion checks against the CPU checker.  Everything the filter does not accept goes to the exact fp64 evaluation."""
import numpy as np
import pytest

import oracle_binding as ob

S = 120
KB = (S + 15) // 16
EPS = 4.07e-6


def corr_exact(q, k):
    ql, kl = q.astype(np.longdouble), k.astype(np.longdouble)
    idx = (np.arange(S)[None, :] + np.arange(S)[:, None]) % S          # [s][u] -> (u + s) mod S
    return (ql[idx] * kl[None, :]).sum(axis=1)


def corr_fp32(q, k, model):
    qf, kf = q.astype(np.float32), k.astype(np.float32)
    kp = np.zeros(16 * KB, np.float32); kp[:S] = kf                    # K padded with zeros, as the kernel's B image
    acc = np.zeros(S, np.float32)
    s = np.arange(S)
    for b in range(KB):
        for e in range(4):
            us = 16 * b + 4 * np.arange(4) + e                         # the four sectors of one MFMA step
            a = qf[(us[None, :] + s[:, None]) % S]                     # [s][4]
            bb = kp[us]
            if model == "fma_chain":
                for j in range(4):
                    acc = (a[:, j].astype(np.float64) * np.float64(bb[j]) + acc.astype(np.float64)).astype(np.float32)
            elif model == "rounded_products":
                for j in range(4):
                    acc = (acc + (a[:, j] * bb[j]).astype(np.float32)).astype(np.float32)
            else:                                                      # one rounding per step
                acc = ((a.astype(np.float64) * bb.astype(np.float64)[None, :]).sum(axis=1) + acc.astype(np.float64)).astype(np.float32)
    return acc, qf, kf


def cases():
    rs = np.random.RandomState(11)
    out = []
    for _ in range(60):
        q = rs.uniform(0, 8, S); out.append((q, np.roll(q, rs.randint(S)) + rs.normal(0, 10.0 ** rs.randint(-6, 0), S)))
    for _ in range(20):                                                # sparse keys, large dynamic range
        q = rs.uniform(0, 8, S) * (rs.rand(S) < 0.3); k = rs.uniform(0, 8, S) * (rs.rand(S) < 0.3) * 10.0 ** rs.randint(-3, 4, S)
        out.append((q, k))
    base = rs.uniform(0, 8, S)
    out.append((base, np.tile(base[:60], 2)))                          # periodic: exact ties
    out.append((np.full(S, 3.0), base))                                # flat
    out.append((base, np.roll(base, 7) * (1 + 1e-7)))                  # near tie of magnitudes
    out.append((base * 1e6, np.roll(base, 50) * 1e-6))
    for _ in range(20):                                                # alternating signs of the error: near-constant keys
        out.append((4.0 + rs.normal(0, 1e-3, S), 4.0 + rs.normal(0, 1e-3, S)))
    return out


@pytest.mark.parametrize("model", ["fma_chain", "rounded_products", "one_rounding_per_step"])
def test_filter_error_is_inside_half_the_margin_and_accepted_shifts_are_the_references(model):
    L = ob.load()
    worst, accepted, total = 0.0, 0, 0
    for q, k in cases():
        c, qf, kf = corr_fp32(q, k, model)
        ce = corr_exact(q, k)
        nq, nk = float(np.sqrt((qf.astype(np.float64) ** 2).sum())), float(np.sqrt((kf.astype(np.float64) ** 2).sum()))
        if nq == 0 or nk == 0:
            continue
        err = float(np.max(np.abs(c.astype(np.longdouble) - ce))) / (nq * nk)
        worst = max(worst, err)
        assert err < 2 * EPS, (model, err)
        eps = np.float32(EPS) * np.float32(nq) * np.float32(nk) + np.float32(1e-12) * np.float32(nq * nq + nk * nk)
        order = np.argsort(-c, kind="stable")
        v1, v2 = c[order[0]], c[order[1]]
        total += 1
        if v2 < v1 - np.float32(4.0) * eps:                            # the kernel's acceptance rule
            accepted += 1
            ref = L.sco_fast_align(S, ob._p(np.ascontiguousarray(q, np.float64), ob.c_double), ob._p(np.ascontiguousarray(k, np.float64), ob.c_double))
            assert int(order[0]) == ref, (model, int(order[0]), ref)
    print(f"{model}: worst error {worst:.3e} of |vq||vk| (half margin {2 * EPS:.3e}), {accepted} of {total} pairs accepted by the filter")
    assert accepted > total // 2


def test_fp16_stage_error_and_accepted_shifts():
    """Stage 1 of the alignment filter: both keys as unit vectors rounded to fp16, products exact, fp32 accumulation (here one
    rounding per product: the worst order).  The error of the normalised correlation must stay below half the lead the kernel
    demands (kAlign16Margin = 3e-3), and a shift it accepts -- norms inside the kernel's range check -- must be the checker's."""
    L = ob.load()
    margin = np.float32(3.0e-3)
    worst, accepted, total = 0.0, 0, 0
    for q, k in cases():
        nq, nk = np.linalg.norm(q), np.linalg.norm(k)
        if not (nq > 0 and nk > 0):
            continue
        qh = (q / nq).astype(np.float32).astype(np.float16).astype(np.float32)
        kh = (k / nk).astype(np.float32).astype(np.float16).astype(np.float32)
        s = np.arange(S)
        acc = np.zeros(S, np.float32)
        for u in range(S):
            acc = (acc + (qh[(u + s) % S] * kh[u]).astype(np.float32)).astype(np.float32)
        ce = corr_exact(q, k) / (np.longdouble(nq) * np.longdouble(nk))
        err = float(np.max(np.abs(acc.astype(np.longdouble) - ce)))
        worst = max(worst, err)
        assert err < 1.5e-3, err
        total += 1
        in_range = 1e-30 <= nq <= 4e6 and 1e-30 <= nk <= 4e6 and nq <= 1e4 * nk and nk <= 1e4 * nq
        order = np.argsort(-acc, kind="stable")
        if in_range and acc[order[1]] < acc[order[0]] - margin:
            accepted += 1
            ref = L.sco_fast_align(S, ob._p(np.ascontiguousarray(q, np.float64), ob.c_double), ob._p(np.ascontiguousarray(k, np.float64), ob.c_double))
            assert int(order[0]) == ref, (int(order[0]), ref)
    print(f"fp16 stage: worst error {worst:.3e} (half margin 1.5e-3), {accepted} of {total} pairs accepted")
    assert accepted > total // 3


def test_fp16_stage_bound_from_the_keys_own_rounding_errors():
    """Round 3: the lead the first stage demands is twice |q - qh| (1 + |k - kh|) + |k - kh| + 7.6e-6 with the ACTUAL rounding-error
    norms of the two unit keys (written at ingest, make_sc.hip) instead of the worst case of fp16 rounding.  The error of every
    shift's value must stay inside that bound under the worst accumulation order, the bound must not exceed the constant it
    replaces by more than the accumulation term for ordinary keys, and a shift accepted with the smaller lead must be the
    checker's."""
    L = ob.load()
    worst_ratio, accepted, accepted_old, total = 0.0, 0, 0, 0
    for q, k in cases():
        nq, nk = np.linalg.norm(q), np.linalg.norm(k)
        if not (nq > 0 and nk > 0):
            continue
        uq, uk = q / nq, k / nk
        qh = uq.astype(np.float32).astype(np.float16).astype(np.float64)
        kh = uk.astype(np.float32).astype(np.float16).astype(np.float64)

        def err_norm(u, h):
            d = np.abs(u - h)
            sub = np.abs(h) < 2.0 ** -14                               # below fp16's normal range: may be taken for zero
            d = np.where(sub, np.maximum(d, np.abs(u)), d)
            return float(np.sqrt((d * d).sum())) * (1 + 1e-6) + 1e-10
        eq, ek = err_norm(uq, qh), err_norm(uk, kh)
        bound = eq + ek + eq * ek + 7.6e-6
        s = np.arange(S)
        for flush in (False, True):                                    # a matrix core that keeps subnormal inputs / takes them for zero
            a = np.where(np.abs(qh) < 2.0 ** -14, 0.0, qh) if flush else qh
            b = np.where(np.abs(kh) < 2.0 ** -14, 0.0, kh) if flush else kh
            acc = np.zeros(S, np.float32)
            for u in range(S):
                acc = (acc + (a[(u + s) % S].astype(np.float32) * np.float32(b[u])).astype(np.float32)).astype(np.float32)
            ce = corr_exact(q, k) / (np.longdouble(nq) * np.longdouble(nk))
            err = float(np.max(np.abs(acc.astype(np.longdouble) - ce)))
            assert err <= bound, (err, bound, eq, ek)
            worst_ratio = max(worst_ratio, err / bound)
        total += 1
        in_range = 1e-30 <= nq <= 4e6 and 1e-30 <= nk <= 4e6 and nq <= 1e4 * nk and nk <= 1e4 * nq
        order = np.argsort(-acc, kind="stable")
        lead = min(3.0e-3, 2.0 * (eq + ek + eq * ek) * 1.0001 + 1.6e-5)
        if in_range and acc[order[1]] < acc[order[0]] - np.float32(3.0e-3):
            accepted_old += 1
        if in_range and acc[order[1]] < acc[order[0]] - np.float32(lead):
            accepted += 1
            ref = L.sco_fast_align(S, ob._p(np.ascontiguousarray(q, np.float64), ob.c_double), ob._p(np.ascontiguousarray(k, np.float64), ob.c_double))
            assert int(order[0]) == ref, (int(order[0]), ref, lead)
    print(f"fp16 stage, per-pair bound: worst error / bound {worst_ratio:.3f}; accepted {accepted} (constant lead: {accepted_old}) of {total}")
    assert accepted >= accepted_old and worst_ratio <= 1.0
