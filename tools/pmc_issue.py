"""Instruction-issue accounting per kernel from rocprofv3 --pmc passes (SQ_WAVES, SQ_INSTS_VALU / SALU / LDS / VMEM, SQ_BUSY_CYCLES,
SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY; separate passes, kernels serialised) + the kernel trace of one pass -> JSON on stdout.
For the kernels bound by vector-instruction issue (k_fastw: DESIGN.md 4.2) this is the roof that matters: wave-instructions x the measured
4.1 SIMD cycles each (profiles/r04/valu_rate_saturated.txt) against the SIMD cycles the launch had (1024 SIMDs x duration x 2.4 GHz).
Usage: python tools/pmc_issue.py <dir with one sub-directory per --pmc pass> <images_per_launch>"""
import csv, glob, collections, json, re, sys
d0, nimg = sys.argv[1], int(sys.argv[2])
CLK, SIMDS, CYC_PER_VALU = 2.4e9, 1024, 4.1
def kname(s):
    m = re.search(r"\bk_\w+", s)
    return m.group(0) if m else None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d0 + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in sorted(glob.glob(d0 + "/**/*kernel_trace.csv", recursive=True))[:1]:
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"images_per_launch": nimg, "clock_GHz": CLK / 1e9, "simds": SIMDS, "cycles_per_valu_instruction": CYC_PER_VALU,
       "note": "per launch (mean over the launches of the passes); durations are those of the counter runs (kernels serialised, a few % longer than in the bench)"}
for k in sorted(acc):
    c = {x: acc[k][x] / n[k][x] for x in acc[k]}
    d = sum(dur[k]) / max(len(dur[k]), 1)
    if d == 0 or "SQ_INSTS_VALU" not in c: continue
    w = max(c.get("SQ_WAVES", 1), 1)
    out[k] = {"launch_us_under_counters": d, "waves": w, "valu_per_wave": c["SQ_INSTS_VALU"] / w, "salu_per_wave": c.get("SQ_INSTS_SALU", 0) / w,
              "lds_per_wave": c.get("SQ_INSTS_LDS", 0) / w, "vmem_per_wave": c.get("SQ_INSTS_VMEM", 0) / w,
              "valu_insts_per_launch": c["SQ_INSTS_VALU"],
              "valu_issue_utilisation": c["SQ_INSTS_VALU"] * CYC_PER_VALU / (SIMDS * d * 1e-6 * CLK),
              "busy_valu_frac_of_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
              "wait_inst_frac_of_wave_cycles": c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)}
print(json.dumps(out, indent=1))
