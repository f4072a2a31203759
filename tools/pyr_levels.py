"""Per-level durations of the pyramid kernels from a rocprofv3 kernel trace of tools/stage_times.py (developer tool).
Usage: python tools/pyr_levels.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0), r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
acc = collections.defaultdict(list); lvl = 0
for s, e, n, gx, gy, gz in rows:
    if n == "k_level0": lvl = 0; acc[("k_level0", 0, gx, gy, gz)].append((e - s) / 1e3)
    elif n == "k_resize": lvl += 1; acc[("k_resize", lvl, gx, gy, gz)].append((e - s) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: kv[0][1]):
    v = v[len(v) // 2:]
    print(f"{k[0]:10s} level {k[1]} grid {k[2]:>7s} x {k[3]:>5s} x {k[4]:>4s}  {sum(v) / len(v):7.1f} us (n = {len(v)})")
