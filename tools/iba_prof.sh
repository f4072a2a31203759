#!/bin/bash
# kernel stats of LocalInertialBA (developer tool, run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-ibaprof}
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/s" -o s --output-format csv -- python3 tools/bench_iba.py > "$O/out.txt" 2>/dev/null
grep LocalInertial "$O/out.txt"
python3 - "$O" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/s_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    if m: d[(m.group(0), r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])): print(f"{k[0]:24s} grid {k[1]:>8s} calls {len(v):5d} avg {sum(v) / len(v):8.1f} us total {sum(v) / 1e3:8.2f} ms")
PY
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
