#!/bin/bash
# kernel stats of the LocalBA micro-benchmark, MFMA Schur vs the round-1 VALU form (developer tool, run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-optprof}
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/mfma" -o s --output-format csv -- python3 tools/bench_opt.py > "$O/mfma.txt" 2>/dev/null
export MORB_SCHUR_VALU=1
rocprofv3 --kernel-trace --stats -d "$O/valu" -o s --output-format csv -- python3 tools/bench_opt.py > "$O/valu.txt" 2>/dev/null
unset MORB_SCHUR_VALU
for v in mfma valu; do echo "== $v"; cat "$O/$v.txt" | head -2; python3 tools/kstat.py $(find "$O/$v" -name "s_kernel_stats.csv") | grep "k_g_\|k_schur" ; done
find "$O" -name "*kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
