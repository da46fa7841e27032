#!/usr/bin/env python3
"""Stream form of the sharded front (one process, G shards; on a one-GPU box all shards share the card): G x 5 000 keyframes 64x120,
512 scans through scl_detect_full_stream of the front, against one unsharded engine of the same database.
usage: bench_front_stream.py [G]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R, S, n = 64, 120, 10000
descs = synth_descriptors(n, R, S, seed=11, revisit_frac=0.0)
out = {"shards": G, "keyframes": n}
for name, kw in (("front", dict(devices=[0] * G, exchange=1)), ("single", {})):
    e = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 64, **kw)
    e.save_bulk(descs)
    qs = (n - 100 + (np.arange(512) % 100)).astype(np.int32)
    e.detect_full_stream(qs[:32], 0, n - 100, 16, 2)
    dts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = e.detect_full_stream(qs, 0, n - 100, 16, 2)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    out[name + "_us_per_scan_all"] = [round(x / len(qs) * 1e6, 2) for x in dts]
    out[name] = {"us_per_scan": dt / len(qs) * 1e6, "pairs_per_s": (n - 100) * len(qs) / dt}
    out[name + "_res"] = [res[0].tolist(), res[1].tolist(), res[2].view(np.uint64).tolist()]
    e.close()
assert out.pop("front_res") == out.pop("single_res")          # every scan: same winner, shift and distance bits
print(json.dumps(out))
