#!/bin/bash
# Vector-memory-path PMC passes (TA / TCP / TCC) over the extraction alone (developer tool; run through gpurun).
# Usage: bash tools/pmc_mem.sh <outdir under gpurun_out> [stereo frames per batch]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-pmcm}
B=${2:-256}
rm -rf "$O" && mkdir -p "$O"
i=0
for p in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/p$i" -o q --output-format csv -- python3 tools/stage_times.py $B 2 > /dev/null 2>&1
done
python3 tools/pmc_mem.py "$O" > "$O/pmc_mem_table.txt"
cat "$O/pmc_mem_table.txt"
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -delete
