"""End-to-end detection parity (detectIntra/InterLoopClosureID + full-DB mode) vs the checker."""
import numpy as np
import pytest

import oracle_binding as ob
from scl_slam_amd import ScanContextEngine, ScanContextDescriptor, SclError
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu


def test_c1_plumbing_200_keyframes():
    """BASELINE configs[0]: 200 VLP-16 keyframes, 20x60, single query at curPtr = 199."""
    R, S = 20, 60
    descs, truth = synth_descriptors(200, R, S, seed=1001, revisit_frac=0.03, return_truth=True)
    sc = ScanContextDescriptor()                       # reference defaults (DM.h:404 passes nothing)
    db = ob.OracleDB(ob.make_config())
    for i in range(200):
        sc.saveDescriptorAndKey(descs[i], 0, i)
        db.save_wire(descs[i], 0, i)
    assert sc.getSize() == 200 and sc.getIndex(57) == (0, 57)
    lid, shift = sc.detectIntraLoopClosureID(199)
    o_lid, o_shift, o_dist, _ = db.detect_intra(199)
    assert (lid, shift) == (o_lid, o_shift)
    assert sc.detectIntraLoopClosureID(103) == (-1, 0.0)     # early-out, D.h:1620
    sc.engine.close()


@pytest.mark.parametrize("R,S,k,n", [(20, 60, 3, 600), (64, 120, 3, 500), (64, 120, 25, 400)])
def test_detect_intra_inter_full_match_oracle(R, S, k, n):
    descs, truth = synth_descriptors(n, R, S, seed=50 + k, revisit_frac=0.06, return_truth=True)
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=k, initial_capacity=128)
    db = ob.OracleDB(ob.make_config(R=R, S=S, k=k))
    eng.save_bulk(descs); db.save_bulk(descs)
    curs = sorted(set([c for c, _, _ in truth] + [n - 1, 150, 104, 103]))
    found = 0
    for cur in curs:
        lid, shift, dist = eng.detect_intra(cur)
        o_lid, o_shift, o_dist, _ = db.detect_intra(cur)
        assert (lid, shift) == (o_lid, o_shift) and dist == o_dist
        found += lid >= 0
    assert found >= 1
    for cur in curs[-12:]:                              # the inter path keeps tree-period state: same call sequence
        lid, yaw, dist = eng.detect_inter(cur)
        o_lid, o_yaw, o_dist = db.detect_inter(cur)
        assert (lid, np.float32(yaw)) == (o_lid, np.float32(o_yaw)) and dist == o_dist
    for cur in curs[-6:]:
        lid, nn, sh, dist = eng.detect_full(cur)
        o_lid, o_nn, o_sh, o_dist = db.detect_full(cur)
        assert (lid, nn, sh) == (o_lid, o_nn, o_sh) and dist == o_dist
    eng.close()


def test_inter_with_fewer_keys_than_candidates():
    # tree covers [0, N-100): with N = 101 it holds one key; unfilled candidates read slot 0 (D.h:1710)
    R, S = 20, 60
    descs = synth_descriptors(102, R, S, seed=3)
    eng = ScanContextEngine(num_ring=R, num_sector=S)
    db = ob.OracleDB(ob.make_config())
    eng.save_bulk(descs[:101]); db.save_bulk(descs[:101])
    assert eng.detect_inter(100)[0] == db.detect_inter(100)[0]
    g = eng.detect_inter(100); o = db.detect_inter(100)
    assert g[2] == o[2]
    eng.close()


def test_errors_are_status_codes():
    eng = ScanContextEngine()
    with pytest.raises(SclError):
        eng.detect_intra(5)                 # empty DB: out of range
    with pytest.raises(SclError):
        eng.sc_distance_batch(-1, n=1)      # no staged query
    with pytest.raises(ValueError):
        eng.save_from_wire(np.zeros(7, np.float32))
    eng.close()


@pytest.mark.parametrize("R,S,k", [(64, 120, 3), (20, 60, 5), (64, 120, 25), (80, 180, 3)])
def test_full_mode_also_delivers_the_ringkey_topk(R, S, k):
    """full-DB pass = ring-key top-k + SC distance over everything (fused on the 20x60 / 64x120 grids)."""
    n = 420
    descs = synth_descriptors(n, R, S, seed=7 + k)
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=k, initial_capacity=512)
    db = ob.OracleDB(ob.make_config(R=R, S=S, k=k))
    eng.save_bulk(descs); db.save_bulk(descs)
    for (q, lo, hi) in [(n - 1, 0, n - 100), (n - 7, 13, 250), (5, 0, 2)]:
        nn, sh, d = eng.detect_full_range(q, lo, hi)
        idx, d2 = eng.last_topk(k)
        o_idx, o_d2, o_found = ob.knn(db.ringkeys(hi)[lo:], db.ringkey(q), k)
        exp = [int(x) + lo if x >= 0 else -1 for x in o_idx]
        assert list(idx) == exp
        assert np.array_equal(d2[:o_found].view(np.uint32), o_d2[:o_found].view(np.uint32))
        dist, shift = db.distance_batch(q, cand=np.arange(lo, hi, dtype=np.int32))
        j = int(np.lexsort((np.arange(hi - lo), dist))[0])
        assert (nn, sh) == (lo + j, int(shift[j])) and d == dist[j]
    eng.close()


def test_submit_collect_pipeline_matches_blocking_calls():
    R, S, n = 64, 120, 330
    descs = synth_descriptors(n, R, S, seed=77)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=512)
    eng.save_bulk(descs)
    qs = [n - 1, n - 2, n - 3, n - 4, n - 5, 200]
    ref = [eng.detect_full_range(q, 0, n - 100) for q in qs]
    tickets = [eng.detect_full_submit(q, 0, n - 100) for q in qs]       # six passes in flight
    got = [eng.detect_full_collect(t) for t in reversed(tickets)][::-1]  # collected out of order
    assert got == ref
    empty = eng.detect_full_submit(n - 1, 5, 5)
    assert eng.detect_full_collect(empty) == (-1, 0, 10000000.0)
    with pytest.raises(SclError):
        eng.detect_full_collect(empty)                                    # ticket already consumed
    for _ in range(8):
        eng.detect_full_submit(0, 0, 10)
    with pytest.raises(SclError):
        eng.detect_full_submit(0, 0, 10)                                  # ring of 8 is full
    eng.close()


def test_duplicate_ring_key_in_history_under_both_knn_rules():
    """A yaw-only revisit has the query's ring key bit for bit (the key is rotation invariant).  libnabo, the tree of
    the live intra-robot path (D.h:1631-1642, optionFlags = 0), skips neighbours at squared distance <= FLT_EPSILON;
    nanoflann (D.h:1710-1716) returns them.  knn_exclude_eps selects the rule; the C++ adapter defaults to libnabo's."""
    R, S, n = 20, 60, 400
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.0)
    descs[n - 1] = np.roll(descs[57], 23, axis=1)               # pure rotation of keyframe 57: identical ring key
    eps = float(np.finfo(np.float32).eps)
    for rule in (0.0, eps):
        eng = ScanContextEngine(num_ring=R, num_sector=S, knn_exclude_eps=rule)
        db = ob.OracleDB(ob.make_config(R=R, S=S, knn_exclude_eps=rule))
        eng.save_bulk(descs); db.save_bulk(descs)
        assert np.array_equal(eng.get_ringkey(n - 1).view(np.uint32), eng.get_ringkey(57).view(np.uint32))
        got, want = eng.detect_intra(n - 1), db.detect_intra(n - 1)
        assert got[:2] == want[:2] and got[2] == want[2]
        idx, d2, found = eng.ringkey_topk(n - 1, 0, n - 100, 3)
        if rule == 0.0:
            assert got[0] == 57 and got[1] == 23.0 and idx[0] == 57 and d2[0] == 0.0      # nanoflann: the twin is candidate no. 1
        else:
            assert 57 not in idx and got[0] != 57                                          # libnabo: the twin is never proposed
        # the inter-robot path uses nanoflann's rule regardless (D.h:1710-1716)
        assert eng.detect_inter(n - 1)[0] == db.detect_inter(n - 1)[0] == 57
        eng.close()
