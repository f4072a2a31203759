#!/bin/bash
# rocprofv3 kernel stats of the tracking chain (tools/bench_tracking.py).  Usage (through gpurun): bash tools/prof_tracking.sh <tag> [B]
set -u
T=${1:-trk}; B=${2:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$T
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/stats" -o s --output-format csv -- python3 tools/bench_tracking.py $B 10 > "$O/bench.json" 2>/dev/null
tail -1 "$O/bench.json"
cp "$O/stats/s_kernel_stats.csv" "$O/tracking_kernel_stats.csv" 2>/dev/null
python3 tools/kstat.py "$O/tracking_kernel_stats.csv" | head -40
rm -rf "$O/stats"
# one frame at a time: the kernel timeline of a step
rocprofv3 --kernel-trace -d "$O/trace1" -o t --output-format csv -- python3 tools/bench_tracking.py 1 12 chain-only > /dev/null 2>&1
python3 tools/trk_trace.py $(ls $O/trace1/*kernel_trace.csv | head -1) > "$O/tracking_b1_timeline.txt"; cat "$O/tracking_b1_timeline.txt"
rocprofv3 --kernel-trace -d "$O/traceB" -o t --output-format csv -- python3 tools/bench_tracking.py $B 6 chain-only > /dev/null 2>&1
python3 tools/trk_trace.py $(ls $O/traceB/*kernel_trace.csv | head -1) > "$O/tracking_b${B}_timeline.txt"; cat "$O/tracking_b${B}_timeline.txt"
rm -rf "$O/trace1" "$O/traceB"
