#!/usr/bin/env python3
"""Timeline of the driver's short call (`bench.py --steps 20 --warmup 5`) from a rocprofv3 kernel trace (csv): every kernel of
the last timed block with start / end relative to the block's first kernel and the gap to its predecessor.
usage: trace_short_call.py kernel_trace.csv [kernels_per_block]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or r.get("kernel_name"))[:48]) for r in rows)
ks = [k for k in ks if k[2].startswith(("sc_", "void scl::", "scl::")) or "sc_" in k[2]]
# blocks = runs of kernels separated by > 200 us of idle device
blocks, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - max(e for _, e, _ in cur) > 200_000:
        blocks.append(cur); cur = [k]
    else:
        cur.append(k)
blocks.append(cur)
for b in blocks[-2:]:
    t0 = b[0][0]; prev_end = t0
    print(f"--- block of {len(b)} kernels, {(max(e for _, e, _ in b) - t0) / 1e3:.1f} us from first start to last end")
    for s, e, n in b:
        print(f"  {n:48s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}")
        prev_end = max(prev_end, e)
