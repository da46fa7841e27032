#!/usr/bin/env python3
"""Steady-state summary of a rocprofv3 kernel trace (csv) of bench.py: per kernel family the mean duration and the median
period between starts, the period of the screening products (= the main stream's time per launch group), and how much of
every other family ran beside a products kernel.  usage: trace_summary.py <dir or csv> [window_us]"""
import csv, glob, os, re, sys
path = sys.argv[1]
if os.path.isdir(path):
    c = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if not c: sys.exit("no kernel trace under " + path)
    path = c[0]
ks = []
for r in csv.DictReader(open(path)):
    name = r.get("Kernel_Name") or r.get("kernel_name")
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).split("::")[-1]
    name = re.sub(r"<.*", "", name)
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
ks.sort()
fam = {}
for k in ks: fam.setdefault(k[2], []).append(k)
prod = fam.get("sc_screen2_kernel") or fam.get("sc_screen_kernel") or []
if len(prod) < 8: sys.exit("no products kernels in the trace")
lo = prod[len(prod) // 4][0]; hi = prod[-2][0]
def steady(l): return [k for k in l if lo <= k[0] < hi]
sp = steady(prod)
per = sorted(sp[i + 1][0] - sp[i][0] for i in range(len(sp) - 1))
print(f"products period (main stream per launch group): median {per[len(per) // 2] / 1e3:.1f} us, p10 {per[len(per) // 10] / 1e3:.1f}, p90 {per[len(per) * 9 // 10] / 1e3:.1f}")
wall = hi - lo
for name, l in sorted(fam.items(), key=lambda kv: -sum(e - s for s, e, *_ in steady(kv[1]))):
    s = steady(l)
    if not s: continue
    dur = [e - b for b, e, *_ in s]
    ov = 0
    if name != sp[0][2]:
        j = 0
        for b, e, *_ in s:
            for b2, e2, *_ in sp:
                if e2 <= b: continue
                if b2 >= e: break
                ov += min(e, e2) - max(b, b2)
    q = sorted(set(k[3] for k in s))
    print(f"{name[:38]:40s} n={len(s):5d} mean {sum(dur) / len(dur) / 1e3:7.1f} us  busy {100.0 * sum(dur) / wall:5.1f} % of wall  beside products {100.0 * ov / max(1, sum(dur)):5.1f} %  queues {','.join(q)}")
if len(sys.argv) > 2:
    mid = sp[len(sp) // 2][0]
    for b, e, n, q in ks:
        if mid <= b < mid + int(float(sys.argv[2]) * 1e3):
            print(f"{(b - mid) / 1e3:9.1f} {(e - mid) / 1e3:9.1f} {(e - b) / 1e3:7.1f}  q{q} {n}")
