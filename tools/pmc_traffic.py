"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (values are KB on gfx950, see
/opt/skills/guides/MI355X_MICROARCH.md): mean per launch per kernel -> JSON on stdout.
Correction (profiles/r03/fetch_calib.txt, tools/fetch_calib.sh): FETCH_SIZE reads exactly 0.500 x the bytes of coalesced streaming reads
at 4, 8 and 16 bytes per lane, aligned or not (128-byte requests tallied as 64), WRITE_SIZE reads writes exactly at every width
-> traffic = 2 x FETCH_SIZE + WRITE_SIZE.
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
out = {"images_per_launch": nimg, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes; traffic = 2 x FETCH + WRITE, see the calibration)",
       "fetch_correction": 2.0, "write_correction": 1.0}
for k, c in sorted(acc.items()):
    out[k] = {n + "_KB_per_launch": v[0] / max(v[1], 1) for n, v in c.items()}
    out[k]["launches"] = max(v[1] for v in c.values())
    out[k]["traffic_KB_per_launch"] = 2.0 * out[k].get("FETCH_SIZE_KB_per_launch", 0.0) + out[k].get("WRITE_SIZE_KB_per_launch", 0.0)
print(json.dumps(out, indent=1))
