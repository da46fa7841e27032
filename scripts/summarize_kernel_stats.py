#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv with short kernel names: usage summarize_kernel_stats.py <csv>"""
import csv
import re
import sys

tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).split("::")[-1]
    tot += int(r["TotalDurationNs"])
    print(f"{name[:40]:42s} calls {int(r['Calls']):5d}  total {int(r['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(r['AverageNs']) / 1e3:9.1f} us")
print(f"sum of kernels {tot / 1e6:.3f} ms")
