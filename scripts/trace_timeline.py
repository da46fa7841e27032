#!/usr/bin/env python3
"""Timeline of the screening pass from a rocprofv3 kernel trace (csv): per kernel the mean duration, the mean period between
consecutive starts, and how much of the alignment kernel ran beside a screening kernel.  usage: trace_timeline.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ks.sort()
def sel(sub): return [k for k in ks if sub in k[2]]
scr, ali, sur = sel("sc_screen_kernel"), sel("sc_align_kernel"), sel("sc_distance_survivors")
def stats(lst, label):
    if len(lst) < 3: return
    lst = lst[len(lst) // 4:]                     # skip warm-up
    dur = [e - s for s, e, _ in lst]
    per = [lst[i + 1][0] - lst[i][0] for i in range(len(lst) - 1)]
    per.sort()
    print(f"{label:10s} n={len(lst):5d} mean duration {sum(dur) / len(dur) / 1e3:8.1f} us   median period {per[len(per) // 2] / 1e3:8.1f} us")
stats(scr, "screen"); stats(ali, "align"); stats(sur, "survivors")
if scr and ali:
    tot = ov = 0
    j = 0
    for s, e, _ in ali[len(ali) // 4:]:
        tot += e - s
        for s2, e2, _ in scr:
            if e2 <= s: continue
            if s2 >= e: break
            ov += min(e, e2) - max(s, s2)
    print(f"alignment time beside a screening kernel: {100.0 * ov / max(tot, 1):.1f} %")
    t0, t1 = scr[len(scr) // 4][0], scr[-1][1]
    busy = sum(e - s for s, e, _ in scr[len(scr) // 4:])
    print(f"screening kernels cover {100.0 * busy / (t1 - t0):.1f} % of the wall time of the steady state")
if len(sys.argv) > 2:                                   # a window of the steady state, one line per kernel
    def short(n):
        for t in ("sc_screen_kernel", "sc_align_kernel", "sc_distance_survivors", "ingest"):
            if t in n: return t
        return n[:30]
    mid = scr[len(scr) // 2][0]
    win = [k for k in ks if mid <= k[0] < mid + int(float(sys.argv[2]) * 1e3)]
    for s, e, n in win:
        print(f"{(s - mid) / 1e3:9.1f} {(e - mid) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {short(n)}")
