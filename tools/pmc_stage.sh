#!/bin/bash
# PMC passes (issue mix + stalls per kernel) over the extraction alone, 128 images per launch (developer tool; run through gpurun).
# Usage: bash tools/pmc_stage.sh <outdir under gpurun_out> [env assignments for the python process, e.g. MORB_HIP_LIB=...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-pmcs}
shift
for a in "$@"; do export "$a"; done
rm -rf "$O" && mkdir -p "$O"
for p in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/$n" -o q --output-format csv -- python3 tools/stage_times.py 64 3 > /dev/null 2>&1
done
python3 tools/pmc_table.py "$O" > "$O/pmc_issue_table.txt"
head -12 "$O/pmc_issue_table.txt"
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_trace.csv" -delete
