#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (run through gpurun): tools/micro/fetch_calib.hip under two separate --pmc passes.
# Usage: bash tools/fetch_calib.sh <outdir under gpurun_out>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-calib}
rm -rf "$O" && mkdir -p "$O"
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/micro/fetch_calib.hip || exit 1
/tmp/fetch_calib 1024 > "$O/bytes.txt"
for p in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/$p" -o q --output-format csv -- /tmp/fetch_calib 1024 > /dev/null 2>&1
done
python3 tools/fetch_calib.py "$O" | tee "$O/fetch_calib.txt"
find "$O" -name "*agent_info.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete
