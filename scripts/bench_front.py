#!/usr/bin/env python3
"""The front half of the per-incoming-scan path on its own (what scripts/profile_front.sh profiles): raw points -> descriptor ->
database slot in batches of 16 (K3), the one-launch ring-key scan (K2), and the whole pipeline from pinned clouds
(scl_stream_from_points).  Prints one JSON object; the same functions fill bench.py's secondaries."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import bench  # noqa: E402
from scl_slam_amd import ScanContextEngine  # noqa: E402
from scl_slam_amd.synth import synth_descriptors  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {}
if what in ("all", "ingest"):
    out["ingest_per_scan"] = bench.secondary_ingest_per_scan(0)
if what in ("all", "topk"):
    n = bench.N_KEYFRAMES_1GPU
    eng = ScanContextEngine(num_ring=bench.R, num_sector=bench.S, num_exclude_recent=bench.N_EXCLUDE, initial_capacity=n + 64)
    eng.save_bulk(synth_descriptors(n, bench.R, bench.S, seed=1002))
    out["ringkey_topk"] = bench.secondary_ringkey_topk(eng, n - bench.N_EXCLUDE, bench.N_EXCLUDE)
    out["detect_intra_us_p50"] = bench._p50_us(lambda: eng.detect_intra(n - 1), 200, 20)
    eng.close()
if what in ("all", "stream"):
    out["stream_from_points"] = bench.secondary_stream_from_points(0)
if what in ("all", "resident"):
    out["stream_from_resident_points"] = bench.secondary_stream_from_resident_points(0)
print(json.dumps(out))
