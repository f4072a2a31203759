"""Per-kernel table from rocprofv3 --pmc passes (developer tool): duration, instructions per wave, VALU issue
utilisation (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz)) and mean active lanes per VALU instruction.
Usage: python tools/pmc_table.py <dir with one sub-directory per --pmc pass>"""
import csv, glob, collections, re, sys
d0 = sys.argv[1]
def kname(s):
    m = re.search(r"\bk_\w+", s)
    return m.group(0) if m else s.split("(")[0][-28:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d0 + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
tr = sorted(glob.glob(d0 + "/**/*kernel_trace.csv", recursive=True))[:1]
for f in tr:
    for r in csv.DictReader(open(f)):
        dur[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':24s} {'us':>8s} {'waves':>8s} {'valu/w':>7s} {'salu/w':>7s} {'lds/w':>6s} {'vmem/w':>6s} {'VALUutil':>8s} {'lanes':>6s} {'busyVALU':>8s} {'busyLDS':>8s} {'waitAny':>8s} {'waitInst':>8s} {'ldsConf':>8s}")
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    c = {x: acc[k][x] / n[k][x] for x in acc[k]}
    d = sum(dur[k]) / max(len(dur[k]), 1)
    if d == 0: continue
    w = max(c.get("SQ_WAVES", 1), 1)
    util = c.get("SQ_INSTS_VALU", 0) * 4 / (1024 * d * 1e-6 * 2.4e9)
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1) / 64
    print(f"{k[:24]:24s} {d:8.1f} {w:8.0f} {c.get('SQ_INSTS_VALU', 0) / w:7.0f} {c.get('SQ_INSTS_SALU', 0) / w:7.0f} "
          f"{c.get('SQ_INSTS_LDS', 0) / w:6.0f} {c.get('SQ_INSTS_VMEM', 0) / w:6.0f} {util:8.2f} {lanes:6.2f} "
          f"{c.get('SQ_ACTIVE_INST_VALU', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} {c.get('SQ_ACTIVE_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} "
          f"{c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):8.3f}")
