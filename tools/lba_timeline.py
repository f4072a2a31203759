"""Timeline of one LocalBA solve from a rocprofv3 kernel trace (developer tool): per kernel of the last solve, start offset,
duration and the idle gap before it.  Usage: python tools/lba_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:24]))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_ba_reset"
idx = [i for i, r in enumerate(rows) if r[2] == marker]
a = idx[-2] if len(idx) > 1 else idx[-1]
b = idx[-1] if len(idx) > 1 else len(rows)
sel = rows[a:b]
t0 = sel[0][0]; prev_end = t0
busy = 0
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.1f} us  {n:22s} dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}")
    busy += (e - s); prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {len(sel)} launches")
