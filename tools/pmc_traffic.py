"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (values are KB on gfx950, see
/opt/skills/guides/MI355X_MICROARCH.md): mean per launch per kernel -> JSON on stdout.
Usage: python tools/pmc_traffic.py <dir> <images_per_launch>"""
import csv, glob, collections, json, re, sys
d0, nimg = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d0 + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"): continue
        m = re.search(r"\bk_\w+", r["Kernel_Name"])
        if not m: continue
        a = acc[m.group(0)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
out = {"images_per_launch": nimg, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes)",
       # k_fast runs as two launch groups per extraction (the large cells of the small levels have their own): the per-launch mean below
       # is over both, bench.py multiplies it by this count
       "k_fast_launches": 2}
for k, c in sorted(acc.items()):
    out[k] = {n + "_KB_per_launch": v[0] / max(v[1], 1) for n, v in c.items()}
print(json.dumps(out, indent=1))
