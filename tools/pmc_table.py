"""Per-kernel table from rocprofv3 --pmc passes (developer tool): duration, instructions per wave, cyc/valu = SIMD cycles that
passed per VALU wave-instruction ((1024 SIMDs x duration x 2.4 GHz) / SQ_INSTS_VALU) — to be read against the measured issue costs
of profiles/r02/alu_issue.txt: ~1.5 cycles for 4-byte encodings (VOP1/2), ~2.7 for 8-byte ones (VOP3/VOP3P, DPP) with 8 waves per
SIMD, 5.8 - 6.8 for a wave that is alone (round 1 assumed a flat 4) — and mean active lanes per VALU instruction.
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
print(f"{'kernel':24s} {'us':>8s} {'waves':>8s} {'valu/w':>7s} {'salu/w':>7s} {'lds/w':>6s} {'vmem/w':>6s} {'cyc/valu':>8s} {'lanes':>6s} {'busyVALU':>8s} {'busyLDS':>8s} {'waitAny':>8s} {'waitInst':>8s} {'ldsConf':>8s}")
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    c = {x: acc[k][x] / n[k][x] for x in acc[k]}
    d = sum(dur[k]) / max(len(dur[k]), 1)
    if d == 0: continue
    w = max(c.get("SQ_WAVES", 1), 1)
    util = (1024 * d * 1e-6 * 2.4e9) / max(c.get("SQ_INSTS_VALU", 0), 1)
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1) / 64
    print(f"{k[:24]:24s} {d:8.1f} {w:8.0f} {c.get('SQ_INSTS_VALU', 0) / w:7.0f} {c.get('SQ_INSTS_SALU', 0) / w:7.0f} "
          f"{c.get('SQ_INSTS_LDS', 0) / w:6.0f} {c.get('SQ_INSTS_VMEM', 0) / w:6.0f} {util:8.2f} {lanes:6.2f} "
          f"{c.get('SQ_ACTIVE_INST_VALU', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} {c.get('SQ_ACTIVE_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} "
          f"{c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):8.3f} "
          f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):8.3f}")
