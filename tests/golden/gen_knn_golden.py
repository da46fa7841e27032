"""Generates tests/golden/knn_golden.json with the REFERENCE's own vendored nanoflann.

Run in the build container only (needs oracle/_ref/libnanoflann_ref.so, which
oracle/Makefile compiles from /root/reference/include where it lies):

    make -C oracle && python tests/golden/gen_knn_golden.py

Inputs are regenerated from seeds by ``golden_keys`` (same function the tests import);
only the outputs of the reference (index lists and the float bit patterns of the squared
distances) are stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

# name, N, R, k, seed, n_queries, kind
CASES = [
    ("c1_vlp16_99x20_k3", 99, 20, 3, 1001, 8, "walk"),          # BASELINE C1: first 99 keys at curPtr = 199
    ("c2_hdl64_9900x64_k3", 9900, 64, 3, 1002, 8, "walk"),
    ("c2_hdl64_9900x64_k25", 9900, 64, 25, 1002, 4, "walk"),
    ("c5_livox_3000x80_k10", 3000, 80, 10, 1005, 4, "walk"),
    ("tail_300x22_k5", 300, 22, 5, 7, 4, "walk"),                 # R % 4 != 0: nanoflann's tail loop
    ("tiny_2x20_k3", 2, 20, 3, 9, 2, "walk"),                     # fewer points than k
    ("dups_64x20_k6", 64, 20, 6, 13, 3, "dups"),                  # deliberate equal distances
]


def golden_keys(N, R, seed, kind):
    rs = np.random.RandomState(seed)
    if kind == "dups":
        base = rs.uniform(0, 4, size=(8, R)).astype(np.float32)
        return np.ascontiguousarray(base[rs.randint(0, 8, size=N)])
    keys = np.empty((N, R), dtype=np.float32)
    cur = rs.uniform(0.5, 4.0, size=R)
    for i in range(N):
        cur = np.clip(cur + 0.08 * rs.standard_normal(R), 0.0, 12.0)
        keys[i] = cur.astype(np.float32)
    return keys


def golden_queries(keys, seed, nq):
    rs = np.random.RandomState(seed + 77)
    pick = rs.randint(0, keys.shape[0], size=nq)
    q = keys[pick].astype(np.float64) + 0.05 * rs.standard_normal((nq, keys.shape[1]))
    return np.ascontiguousarray(q, dtype=np.float32)


def main():
    L = ob.load_ref_nanoflann()
    if L is None:
        raise SystemExit("oracle/_ref/libnanoflann_ref.so missing: run `make -C oracle` where /root/reference exists")
    out = {"generator": "tests/golden/gen_knn_golden.py", "source": "reference include/nanoflann.hpp v1.3.2 "
           "driven as descriptor.h:1699,1710-1716", "cases": {}}
    for name, N, R, k, seed, nq, kind in CASES:
        keys = golden_keys(N, R, seed, kind)
        queries = golden_queries(keys, seed, nq)
        res = []
        for q in queries:
            idx, d2, found = ob.ref_knn(L, keys, q, k)
            res.append({"found": int(found), "idx": [int(x) for x in idx[:found]],
                        "d2_bits": [int(x) for x in d2[:found].view(np.uint32)]})
        out["cases"][name] = {"N": N, "R": R, "k": k, "seed": seed, "nq": nq, "kind": kind, "results": res}
    with open(os.path.join(HERE, "knn_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote knn_golden.json with", len(CASES), "cases")


if __name__ == "__main__":
    main()
