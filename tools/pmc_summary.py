"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per launch (developer tool).
Usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k.split("(")[0][-40:]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k)
    for c, (s, n) in sorted(cs.items()): print(f"   {c:28s} {s / n:16.1f}  (n={n})")
