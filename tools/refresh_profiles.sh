#!/bin/bash
# Regenerates profiles/<round>/ on the GPU box (run through gpurun): rocprofv3 kernel stats of the default bench
# command plus separate --pmc passes (HBM traffic: FETCH_SIZE / WRITE_SIZE; issue mix: SQ_*), as the MI355X guide
# prescribes (counters never combined with sys/runtime traces).  Usage: bash tools/refresh_profiles.sh r01
set -u
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats -d "$O/stats" -o s --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$O/bench_under_rocprof.json" 2>/dev/null
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  # counter passes at B = 64 (128 images per launch), un-pipelined: kernels are serialised under --pmc anyway
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/pmc/$n" -o q --output-format csv -- python3 bench.py --batch 64 --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
python3 tools/pmc_table.py "$O/pmc" > "$O/pmc_issue_table.txt"
python3 tools/pmc_traffic.py "$O/pmc" 128 > "$O/pmc_traffic_b64.json"
cp "$O/stats/s_kernel_stats.csv" "$O/extract_match_b256_kernel_stats.csv"
# the optimisers (secondary metrics): kernel stats of the tracking / mapping micro-benchmarks
rocprofv3 --kernel-trace --stats -d "$O/opt" -o s --output-format csv -- python3 tools/bench_opt_all.py > "$O/optimisers_bench.txt" 2>/dev/null
cp "$O/opt/s_kernel_stats.csv" "$O/optimisers_kernel_stats.csv"
rm -rf "$O/opt"
# this round's extra evidence: VALU issue microbenchmark, MFMA Schur counters and A/B timings, phase costs of k_fast
hipcc --offload-arch=gfx950 -O3 -o /tmp/alu_issue tools/alu_issue.hip 2>/dev/null && /tmp/alu_issue > "$O/alu_issue.txt" 2>&1
python3 tools/fastw_stats.py 64 > "$O/k_fastw_phase_counts_b64.txt" 2>/dev/null
python3 tools/stage_times.py 64 > "$O/extract_stage_times_isolated.txt" 2>/dev/null; python3 tools/stage_times.py 256 >> "$O/extract_stage_times_isolated.txt" 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -o /tmp/ldp tools/micro/ldlt_phases.hip 2>/dev/null && { /tmp/ldp 120; /tmp/ldp 150; } > "$O/dense_ldlt_phases.txt" 2>&1
rm -rf "$O/stats" "$O"/pmc/*/q_kernel_trace.csv "$O"/pmc/*/q_agent_info.csv
ls -la "$O"
