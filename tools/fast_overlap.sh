#!/bin/bash
# do the two k_fast launch groups overlap? (developer probe, run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/fo && mkdir -p gpurun_out/fo
rocprofv3 --kernel-trace -d gpurun_out/fo -o t --output-format csv -- python3 tools/stage_times.py 64 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/fo/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(r"k_fast|k_resize|k_level0|k_distribute|k_blur", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-14:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(re.search(r"k_\w+", r["Kernel_Name"]).group(0), f"start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
PY
