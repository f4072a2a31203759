#!/bin/bash
# kernel stats of the LocalBA micro-benchmark (developer tool, run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-optprof}
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/mfma" -o s --output-format csv -- python3 tools/bench_opt.py > "$O/mfma.txt" 2>/dev/null
cat "$O/mfma.txt" | head -2; python3 tools/kstat.py $(find "$O/mfma" -name "s_kernel_stats.csv") | grep "k_g_\|k_schur\|k_ba"
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
