#!/bin/bash
# MFMA utilisation of the Schur kernels, one program (= one problem shape) per call (profiles/<round>/mfma_schur.txt; run through gpurun):
# separate --pmc passes, kernel trace only.   Usage: bash tools/mfma_util.sh <outdir under gpurun_out> [program, default tools/bench_opt.py]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-mfmautil}
PROG=${2:-tools/bench_opt.py}
rm -rf "$O" && mkdir -p "$O"
for p in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/$n" -o q --output-format csv -- python3 $PROG > /dev/null 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, collections, re, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int)); dur = collections.defaultdict(list)
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (re.search(r"\bk_\w+", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        dur[(re.search(r"\bk_\w+", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# per launch.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) (the guide's MfmaUtil with the per-XCD sum undone);")
print("# FP64 matrix rate = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop / duration, against 78.6 TFLOP/s dense FP64 MFMA peak")
print(f"{'kernel':32s} {'us':>8s} {'waves':>7s} {'mfma instr':>10s} {'MFMA busy':>9s} {'TFLOP/s':>8s} {'of peak':>7s}")
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    c = {x: acc[k][x] / n[k][x] for x in acc[k]}
    if c.get("SQ_INSTS_VALU_MFMA_F64", 0) == 0: continue
    d = sum(dur[k]) / max(len(dur[k]), 1)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
    tf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512 / (d * 1e-6) / 1e12
    print(f"{k:32s} {d:8.1f} {c.get('SQ_WAVES', 0):7.0f} {c.get('SQ_INSTS_VALU_MFMA_F64', 0):10.0f} {busy:9.3f} {tf:8.2f} {tf / 78.6:7.3f}")
PY
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
