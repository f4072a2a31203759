#!/bin/bash
# PMC passes of the C2 extraction alone (128 images per launch; developer tool, run through gpurun): issue mix + stalls per kernel.
# Usage: [MORB_HIP_LIB=...] bash tools/pmc_extract.sh <outdir under gpurun_out> [kernel-name filter]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-pmcx}
rm -rf "$O" && mkdir -p "$O"
for p in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/$n" -o q --output-format csv -- python3 tools/stage_times.py 64 3 > /dev/null 2>&1
done
python3 tools/pmc_table.py "$O" | grep -E "kernel|${2:-k_}" > "$O/pmc_issue_table.txt"
cat "$O/pmc_issue_table.txt"
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*.csv" -size +2M -delete
