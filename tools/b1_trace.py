"""Kernel time vs span of ONE stereo frame's extraction (developer tool): from a rocprofv3 kernel trace of tools/latency_b1.py,
the kernels of the last device-resident frames: sum of durations, first start -> last end, per-kernel durations and the gaps in front of them.
Usage: python tools/b1_trace.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24]))
rows.sort()
# frames start at k_level0; take frame 30 of the device-resident loop
starts = [i for i, r in enumerate(rows) if r[2] == "k_level0"]
if len(starts) < 2:
    sys.exit(f"{sys.argv[1]}: fewer than two k_level0 launches in the trace ({len(starts)}): nothing to measure")
f = min(30, len(starts) - 2)   # (frame 30 when the trace is long enough, else the last complete one)
i0, i1 = starts[f], starts[f + 1]
fr = rows[i0:i1]
print(f"kernels {len(fr)}, sum of durations {sum(e - s for s, e, _ in fr) / 1e3:.1f} us, span {(fr[-1][1] - fr[0][0]) / 1e3:.1f} us, frame period {(rows[i1][0] - fr[0][0]) / 1e3:.1f} us")
prev = fr[0][0]
for s, e, n in fr:
    print(f"  {n:22s} start +{(s - fr[0][0]) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap before {(s - prev) / 1e3:6.1f}")
    prev = e
