"""K2 parity: brute-force ring-key top-k vs the reference's nanoflann (golden) and the checker."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from golden.gen_knn_golden import golden_keys, golden_queries
from scl_slam_amd import ScanContextEngine

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn_golden.json")


def engine_with_ringkeys(keys, S=8):
    """A descriptor whose every row is constant has ring key == that constant (mean of equal
    floats is exact when S is a power of two), so any key table can be planted."""
    N, R = keys.shape
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, initial_capacity=max(64, N + 8))
    descs = np.repeat(keys[:, :, None], S, axis=2).astype(np.float32)
    eng.save_bulk(descs)
    return eng


@pytest.mark.parametrize("name", sorted(json.load(open(GOLD))["cases"].keys()))
def test_topk_equals_reference_nanoflann_golden(name):
    case = json.load(open(GOLD))["cases"][name]
    keys = golden_keys(case["N"], case["R"], case["seed"], case["kind"])
    queries = golden_queries(keys, case["seed"], case["nq"])
    eng = engine_with_ringkeys(keys)
    for i in range(min(4, case["N"])):
        assert np.array_equal(eng.get_ringkey(i), keys[i])
    k = case["k"]
    for q, gold in zip(queries, case["results"]):
        eng.stage_query(np.repeat(q[:, None], 8, axis=1))
        idx, d2, found = eng.ringkey_topk(-1, 0, case["N"], k)
        assert found == gold["found"]
        bits = [int(b) for b in d2[:found].view(np.uint32)]
        if case["kind"] == "dups":
            assert sorted(bits) == sorted(gold["d2_bits"])
            o_idx, o_d2, o_found = ob.knn(keys, q, k)         # tie rule: ascending index
            assert list(idx[:found]) == list(o_idx[:o_found])
        else:
            assert [int(x) for x in idx[:found]] == gold["idx"]
            assert bits == gold["d2_bits"]
        assert all(int(x) == -1 for x in idx[found:])
    eng.close()


@pytest.mark.parametrize("N,R,k", [(5000, 64, 25), (1025, 20, 3), (40000, 64, 10), (300, 22, 64)])
def test_topk_matches_checker_on_ranges(N, R, k):
    keys = golden_keys(N, R, 4242 + N, "walk")
    eng = engine_with_ringkeys(keys)
    rs = np.random.RandomState(N)
    for (lo, hi) in [(0, N), (0, N - 100), (17, N // 2), (N - 3, N)]:
        qi = int(rs.randint(0, N))
        idx, d2, found = eng.ringkey_topk(qi, lo, hi, k)
        o_idx, o_d2, o_found = ob.knn(keys[lo:hi], keys[qi], k)
        assert found == o_found
        assert list(idx[:found]) == [int(x) + lo for x in o_idx[:o_found]]
        assert np.array_equal(d2[:found].view(np.uint32), o_d2[:o_found].view(np.uint32))
    eng.close()


def test_topk_self_match_exclusion_switch():
    keys = golden_keys(500, 20, 99, "walk")
    N, R = keys.shape
    eng = ScanContextEngine(num_ring=R, num_sector=8, knn_exclude_eps=float(np.finfo(np.float32).eps),
                            initial_capacity=512)
    eng.save_bulk(np.repeat(keys[:, :, None], 8, axis=2))
    idx, d2, found = eng.ringkey_topk(123, 0, N, 3)            # libnabo semantics: the query itself is skipped
    o_idx, o_d2, o_found = ob.knn(keys, keys[123], 3, exclude_eps=float(np.finfo(np.float32).eps))
    assert 123 not in idx and list(idx) == list(o_idx)
    eng.close()


def test_alternating_k_never_returns_the_previous_calls_block():
    """The k candidates leave as one block in pinned memory with a polled sequence word BEHIND the block, at an offset that depends on
    k: a call with k = 20 leaves idx[] / d2[] data where a k = 3 (byte 64) or k = 1 (byte 32) call polls.  The word is cleared before
    every launch, so a stale keyframe index that happens to equal the next sequence number cannot end the poll early."""
    from scl_slam_amd.synth import synth_descriptors
    R, S, n = 20, 60, 700
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.05)
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, initial_capacity=1024)
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    eng.save_bulk(descs); db.save_bulk(descs)
    keys = db.ringkeys()
    for rep in range(120):
        for k in (20, 3, 9, 1, 16, 3):
            q = 300 + (rep * 7 + k) % 390
            idx, d2, dist, shift, found = eng.topk_with_distance(q, 0, q - 100, k)
            o_idx, o_d2, o_found = ob.knn(keys[:q - 100], keys[q], k)
            assert found == o_found and list(idx[:found]) == [int(x) for x in o_idx[:o_found]], (rep, k, q)
            o_dist, o_shift = db.distance_batch(q, cand=np.asarray(idx[:found], dtype=np.int32))
            assert np.array_equal(dist[:found].view(np.uint64), o_dist.view(np.uint64)) and np.array_equal(shift[:found], o_shift), (rep, k, q)
    eng.close()
