"""Per (kernel, grid) averages of the counters collected by tools/pmc_mem.sh (developer tool)."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        m = re.search(r"\bk_\w+", r["Kernel_Name"])
        if not m: continue
        per[(r["Dispatch_Id"], m.group(0), r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for (d, k, g), cs in per.items():
        for c, v in cs.items(): acc[(k, g)][c].append(v)
names = sorted({c for v in acc.values() for c in v})
for (k, g), cs in sorted(acc.items()):
    print(f"{k} grid {g}")
    for c in names:
        if c in cs:
            v = cs[c][len(cs[c]) // 2:]
            print(f"    {c:45s} {sum(v) / len(v):16.0f}")
