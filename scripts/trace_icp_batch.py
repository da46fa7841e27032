#!/usr/bin/env python3
"""configs[2] batch (one scan against 25 candidates of 100 k points, from the keyframe store) under rocprofv3 --kernel-trace:
  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 scripts/trace_icp_batch.py run
  analyse:  python3 scripts/trace_icp_batch.py DIR/.../t_kernel_trace.csv
The run separates its phases by 50 ms of idle device; the analysis splits the trace there and prints, per phase, the span
from first start to last end and the kernels by total duration (count, total, share of the span if serialised)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "run":
    import numpy as np
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
    n_cand, n_pts = 25, 100000
    eng = ScanContextEngine(num_ring=64, num_sector=120)
    ident = np.eye(4, dtype=np.float32)
    for c in range(n_cand):
        tgt = synth_structured_cloud(n_pts, seed=100 + c, extent=60.0)
        eng.keyframe_put(0, c, tgt)
        if c == 0:
            T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
            rs = np.random.RandomState(3); src0 = tgt.copy()
            p = tgt[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            src0[:, :3] = (p + 0.01 * rs.standard_normal(p.shape)).astype(np.float32)
    eng.keyframe_put(0, n_cand, src0)
    keys = np.arange(n_cand, dtype=np.int32); poses = np.tile(ident.reshape(1, 1, 16), (n_cand, 1, 1))
    for est in (1, 0):
        pp = eng.icp_default_params(); pp.max_iterations = 30; pp.estimator = est; pp.normal_radius = 1.0
        for rep in range(2):
            time.sleep(0.05)
            t0 = time.perf_counter()
            out = eng.loop_icp_batch_from_store(0, n_cand, ident, keys, 0, poses, float(os.environ.get("LEAF", "0.02")), pp)
            print(f"estimator {est} rep {rep}: {1e3 * (time.perf_counter() - t0):.2f} ms wall, iterations mean {float(np.mean(out[3])):.2f}, points {out[4]} vs {float(np.mean(out[5])):.0f}", flush=True)
    eng.close()
else:
    import csv
    from collections import defaultdict
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or r.get("kernel_name"))) for r in rows)
    blocks, cur = [], [ks[0]]
    for k in ks[1:]:
        if k[0] - max(e for _, e, _ in cur) > 20_000_000: blocks.append(cur); cur = [k]
        else: cur.append(k)
    blocks.append(cur)
    for bi, b in enumerate(blocks[-4:]):
        span = (max(e for _, e, _ in b) - b[0][0]) / 1e3
        busy, last = 0, b[0][0]
        for s, e, _ in b:                       # union of the kernel intervals
            if e > last: busy += e - max(s, last); last = e
        print(f"--- phase {bi}: {len(b)} kernels, span {span:.0f} us, device busy {busy / 1e3:.0f} us")
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in b:
            short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split("::")[-1][:44] or n[:44]
            agg[short][0] += 1; agg[short][1] += e - s
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"   {n:44s} calls {c:4d} total {t / 1e3:8.1f} us  avg {t / c / 1e3:7.1f}")
        tile = [(s - b[0][0], e - s) for s, e, n in b if "icp_tile_search_kernel" in n.split("(")[0] + n.split("(anonymous namespace)::")[1 if "(anonymous namespace)::" in n else 0].split("(")[0]]
        if tile:
            print("   icp_tile_search_kernel, start (us after the phase's first kernel) : duration (us):")
            print("      " + "  ".join(f"{s / 1e3:.0f}:{d / 1e3:.0f}" for s, d in tile))
    if os.environ.get("DUMP"):
        b = blocks[-1]
        t0 = b[0][0]
        for s, e, n in b[:int(os.environ["DUMP"])]:
            short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split("::")[-1][:40] or n[:40]
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {short}")
