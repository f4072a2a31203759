"""Counter / true-byte ratios of the calibration kernels (tools/micro/fetch_calib.hip under tools/fetch_calib.sh).
Usage: python tools/fetch_calib.py <dir>"""
import collections, csv, glob, json, re, sys
d0 = sys.argv[1]
txt = open(d0 + "/bytes.txt").read()
nbytes = int(re.search(r"bytes_per_kernel (\d+)", txt).group(1))
m = re.search(r"window48_bytes (\d+) \(requested\) (\d+)", txt)
win_req, win_lines = int(m.group(1)), int(m.group(2))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d0 + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def label(k):
    w = {"unsigned char": 1, "unsigned short": 2, "unsigned int": 4, "uint2": 8, "HIP_vector_type<unsigned int, 2u>": 8, "uint4": 16, "HIP_vector_type<unsigned int, 4u>": 16}
    m = re.search(r"k_calib_(read|write)<(.+?)>\(", k)
    if m:
        return f"{m.group(1)} {w.get(m.group(2), m.group(2))} B/lane", nbytes
    if "unaligned16" in k: return "read 16 B/lane at address 1 mod 16", nbytes
    if "window48" in k: return "read 3 x 16 B of every 832-B row (k_fastw window)", win_req
    return None, 0
rows, table = [], {}
for k, c in acc.items():
    name, true = label(k)
    if not name: continue
    for cn, vals in c.items():
        v = sum(vals) / len(vals)
        if cn == "FETCH_SIZE" and name.startswith("write"): continue
        if cn == "WRITE_SIZE" and name.startswith("read"): continue
        ratio = v * 1024 / true      # counters are in KB
        rows.append((name, cn, v * 1024, true, ratio))
        table[name] = {"counter": cn, "counter_bytes": v * 1024, "true_bytes": true, "ratio": ratio}
print(f"{'access':52s} {'counter':10s} {'counter bytes':>16s} {'true bytes':>16s} {'ratio':>7s}")
for r in sorted(rows):
    print(f"{r[0]:52s} {r[1]:10s} {r[2]:16.0f} {r[3]:16d} {r[4]:7.3f}")
if "read 3 x 16 B of every 832-B row (k_fastw window)" in table:
    t = table["read 3 x 16 B of every 832-B row (k_fastw window)"]
    print(f"(window rows: counter / bytes of the 64-byte lines the rows touch = {t['counter_bytes'] / win_lines:.3f})")
json.dump(table, open(d0 + "/fetch_calib.json", "w"), indent=1)
