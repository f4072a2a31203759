#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 --kernel-trace CSV: wall time covered by >= 1 / >= 2 kernels, idle gaps, and per-kernel
totals inside the window of the last N k_fast launches.  Usage: python tools/trace_overlap.py trace.csv"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")) for r in rows]
ev.sort()
fast = [e for e in ev if e[2].startswith("k_fast")]
# steady state of the timed region: skip the warm-up launches and the five serial "isolated" extractions bench.py appends
t0, t1 = fast[4][0], fast[-7][1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = []
for s, e, _ in win: pts += [(s, 1), (e, -1)]
pts.sort()
cov = defaultdict(int); last = t0; depth = 0
for t, d in pts:
    cov[min(depth, 4)] += t - last; last = t; depth += d
wall = t1 - t0
print(f"window {wall/1e6:.3f} ms, {len(win)} kernels; idle {100*cov[0]/wall:.1f} %  1 kernel {100*cov[1]/wall:.1f} %  2 {100*cov[2]/wall:.1f} %  3 {100*cov[3]/wall:.1f} %  >=4 {100*cov[4]/wall:.1f} %")
alone = defaultdict(int); active = {}; last = t0
for t, d, n in sorted([(s_, 1, n_) for s_, e_, n_ in win] + [(e_, -1, n_) for s_, e_, n_ in win]):
    if len(active) == 1: alone[next(iter(active))] += t - last
    last = t
    if d > 0: active[n] = active.get(n, 0) + 1
    else:
        active[n] -= 1
        if active[n] == 0: del active[n]
print("alone on the chip:", ", ".join(f"{n} {100*v/wall:.1f} %" for n, v in sorted(alone.items(), key=lambda x: -x[1])[:6]))
tot = defaultdict(lambda: [0, 0])
for s, e, n in win: tot[n][0] += e - s; tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:14]:
    print(f"  {n:28s} {c:4d} launches  sum {d/1e6:8.3f} ms = {100*d/wall:5.1f} % of wall   avg {d/c/1e3:8.1f} us")
