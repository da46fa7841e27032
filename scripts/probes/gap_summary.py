#!/usr/bin/env python3
"""median gaps between consecutive kernels of a rocprofv3 kernel trace (csv dir or file): name_a -> name_b"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p): p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(p)))
gaps = {}
for a, b in zip(rows, rows[1:]):
    gaps.setdefault((a[2], b[2]), []).append((b[0] - a[1]) / 1e3)
for k, v in gaps.items():
    v.sort()
    d = sorted((e - s) / 1e3 for s, e, n in rows if n == k[1])
    print(f"{k[0][:28]:28s} -> {k[1][:28]:28s} n={len(v):3d}  gap median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}   ({k[1][:20]} duration median {d[len(d)//2]:.1f} us)")
