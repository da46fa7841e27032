#!/usr/bin/env python3
"""A window of a rocprofv3 kernel trace, one line per kernel: start (us from the window's begin), duration, queue, name.
usage: trace_window.py <dir or kernel_trace.csv> [window_us=1500] [fraction of the trace's kernels in front of the window=0.5]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p): p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 1500.0
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = list(csv.DictReader(open(p)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Kernel_Name") or r.get("kernel_name")) for r in rows)
t0 = ks[int((len(ks) - 1) * frac)][0]                  # (the window starts at the kernel that far into the trace)
qs = {}
for s, e, q, n in ks:
    if s < t0 or s > t0 + win * 1e3: continue
    qi = qs.setdefault(q, len(qs))
    nm = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('scl::', '')
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f}  q{qi}  {' ' * (28 * qi)}{nm[:44]}")
