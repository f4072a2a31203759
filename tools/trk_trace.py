"""Kernel timeline of ONE tracking-chain step (developer tool): from a rocprofv3 kernel trace of `tools/bench_tracking.py <B> <steps> chain-only`,
the kernels of the last complete step (a step starts at k_prep_last): duration, start offset and the gap in front of each.
Usage: python tools/trk_trace.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:28]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_prep_last"]
if len(starts) < 3:
    sys.exit("fewer than three steps in the trace")
i0, i1 = starts[-2], starts[-1]
while i0 > 0 and rows[i0 - 1][2] not in ("k_discard",) and rows[i0][0] - rows[i0 - 1][1] < 20000:   # the step's leading copy / fill
    i0 -= 1
fr = rows[i0:i1]
# cut the trailing launches that belong to the next step's lead-in
while fr and fr[-1][2] != "k_discard":
    fr.pop()
print(f"kernels {len(fr)}, sum of durations {sum(e - s for s, e, _ in fr) / 1e3:.1f} us, span {(fr[-1][1] - fr[0][0]) / 1e3:.1f} us")
prev = fr[0][0]
for s, e, n in fr:
    print(f"  {n:28s} start +{(s - fr[0][0]) / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap before {(s - prev) / 1e3:6.1f}")
    prev = e
