#!/usr/bin/env python3
"""One blocking scan over the whole database (scl_detect_full_range): the kernels of a call from a rocprofv3 kernel trace.
   rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 scripts/trace_blocking.py run
   python3 scripts/trace_blocking.py DIR/.../t_kernel_trace.csv"""
import csv, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "run":
    import numpy as np
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors
    R, S, N = 64, 120, 10000
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=N + 8)
    eng.save_bulk(synth_descriptors(N, R, S, seed=1002))
    ts = []
    for i in range(60):
        t0 = time.perf_counter(); eng.detect_full_range(N - 1 - (i % 50), 0, N - 100); ts.append((time.perf_counter() - t0) * 1e6)
        time.sleep(0.0005)
    print("blocking call us: p50 %.1f  min %.1f" % (float(np.percentile(ts[10:], 50)), min(ts[10:])))
    time.sleep(0.01)
    ts = []
    for i in range(60):                                   # ... and the reference-faithful call (top-3 + 3 distances + threshold)
        t0 = time.perf_counter(); eng.detect_intra(N - 1 - (i % 50)); ts.append((time.perf_counter() - t0) * 1e6)
        time.sleep(0.0005)
    print("detect_intra call us: p50 %.1f  min %.1f" % (float(np.percentile(ts[10:], 50)), min(ts[10:])))
    eng.close()
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or r.get("kernel_name"))) for r in rows)
ks = [k for k in ks if "ingest" not in k[2]]
blocks, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - max(e for _, e, _ in cur) > 100_000:
        blocks.append(cur); cur = [k]
    else:
        cur.append(k)
blocks.append(cur)
import re
full = [b for b in blocks if any("small_exact" in k[2] or "survivors" in k[2] for k in b)][-3:]          # blocking full-database scans
intra = [b for b in blocks if any("cand_exact" in k[2] or "topk_pack" in k[2] for k in b)][-3:]            # reference-faithful detection calls
for b in full + intra:
    t0 = b[0][0]; prev_end = t0
    print(f"--- call of {len(b)} kernels, {(max(e for _, e, _ in b) - t0) / 1e3:.1f} us from first start to last end")
    for s, e, n in b:
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void scl::", "").replace("scl::", ""))[:56]
        print(f"  {n:56s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}")
        prev_end = max(prev_end, e)
