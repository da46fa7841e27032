#!/usr/bin/env python3
"""What a short timed block costs (the driver runs bench.py with --steps 20): 20 scans through scl_detect_full_stream directly
and through bench.py's FullScanStream path.  usage: short_block.py [n_scans]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.sharded import FullScanStream
from scl_slam_amd.synth import synth_descriptors

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R, S, N = 64, 120, 10000
descs = synth_descriptors(N, R, S, seed=1002, revisit_frac=0.02)
eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 64)
eng.save_bulk(descs)
n_elig = N - 100
q = (n_elig + (np.arange(k) % 100)).astype(np.int32)
lo = np.zeros(k, np.int32); hi = np.full(k, n_elig, np.int32)
def t(fn, rep=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rep):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6
print(f"scl_detect_full_stream, {k} scans: {t(lambda: eng.detect_full_stream(q, lo, hi, 4, 2)):.1f} us")
def via_stream():
    st = FullScanStream(eng, 0, 1, device="cpu", depth=2, merge_every=16, scans_per_launch=4, native_chunk=256)
    st.submit_many(q, 0, n_elig)
    return st.drain()
print(f"FullScanStream.submit_many + drain, {k} scans: {t(via_stream):.1f} us")
print(f"empty torch.cuda.synchronize pair: {t(lambda: None):.1f} us")
