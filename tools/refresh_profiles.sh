#!/bin/bash
# Regenerates profiles/<round>/ on the GPU box (run through gpurun): rocprofv3 kernel stats of the default bench
# command plus separate --pmc passes (HBM traffic: FETCH_SIZE / WRITE_SIZE; issue mix: SQ_*), as the MI355X guide
# prescribes (counters never combined with sys/runtime traces).  Usage: bash tools/refresh_profiles.sh r01
set -u
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
# every step is bounded (a hung A/B build cost round 6 fifty GPU-minutes): T = seconds per step.  MORB_REFRESH_TAIL=1: only the second half (from the
# one-frame latencies on), into the directory the first half left.
T="timeout 420"
mkdir -p "$O"
if [ -z "${MORB_REFRESH_TAIL:-}" ]; then
rm -rf "$O" && mkdir -p "$O"
$T rocprofv3 --kernel-trace --stats -d "$O/stats" -o s --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 > "$O/bench_under_rocprof.json" 2>/dev/null
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  # counter passes at B = 64 (128 images per launch), un-pipelined: kernels are serialised under --pmc anyway
  timeout 300 rocprofv3 --pmc $p --kernel-trace -d "$O/pmc/$n" -o q --output-format csv -- python3 bench.py --batch 64 --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 > /dev/null 2>&1
done
$T python3 tools/pmc_table.py "$O/pmc" > "$O/pmc_issue_table.txt"
$T python3 tools/pmc_traffic.py "$O/pmc" 128 > "$O/pmc_traffic_b64.json"
# round 5: the same counters AT THE BENCH BATCH (B = 512 stereo frames = 1024 images per launch), which is what bench.py's line quotes
# (`roofline.traffic`, `roofline_issue`): traffic and instruction counts per launch, no scaling from a smaller batch
for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  n=$(echo "$p" | cut -c1-12 | tr " " _)
  timeout 600 rocprofv3 --pmc $p --kernel-trace -d "$O/pmc512/$n" -o q --output-format csv -- python3 bench.py --batch 512 --no-pipeline --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 > /dev/null 2>&1
done
$T python3 tools/pmc_traffic.py "$O/pmc512" 1024 > "$O/pmc_traffic_b512.json"
$T python3 tools/pmc_issue.py "$O/pmc512" 1024 > "$O/pmc_issue_b512.json"
$T python3 tools/pmc_table.py "$O/pmc512" > "$O/pmc_issue_table_b512.txt"
rm -rf "$O/pmc512"
cp "$O/stats/s_kernel_stats.csv" "$O/extract_match_b512_kernel_stats.csv"
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts
$T bash tools/fetch_calib.sh calib_$R > /dev/null 2>&1; cp gpurun_out/calib_$R/fetch_calib.txt gpurun_out/calib_$R/fetch_calib.json "$O/" 2>/dev/null
# the optimisers (secondary metrics): kernel stats of the tracking / mapping micro-benchmarks
$T rocprofv3 --kernel-trace --stats -d "$O/opt" -o s --output-format csv -- python3 tools/bench_opt_all.py > "$O/optimisers_bench.txt" 2>/dev/null
cp "$O/opt/s_kernel_stats.csv" "$O/optimisers_kernel_stats.csv"
rm -rf "$O/opt"
# MFMA Schur counters, one problem shape per table: LocalBA (C5: 20 + 6 keyframes, 3000 points), then LocalInertialBA
{ echo "== LocalBundleAdjustment (tools/bench_opt.py)"; bash tools/mfma_util.sh mfma_lba_$R tools/bench_opt.py; echo "== LocalInertialBA (tools/bench_iba.py)"; bash tools/mfma_util.sh mfma_iba_$R tools/bench_iba.py; } > "$O/mfma_schur.txt" 2>/dev/null
$T bash tools/opt_prof.sh optprof_$R > "$O/localba_kernel_stats.txt" 2>/dev/null
$T python3 tools/lba_sizes.py > "$O/localba_window_sizes.txt" 2>/dev/null
# the dense LDL^T of the reduced camera system alone, with wave 0's phase clocks (n = 120: LocalBA C5, n = 150: LocalInertialBA with ten keyframes)
{ hipcc --offload-arch=gfx950 -O3 -o /tmp/ldlt_phases tools/micro/ldlt_phases.hip 2>/dev/null && for n in 120 150 60; do /tmp/ldlt_phases $n; /tmp/ldlt_phases $n 1; done; } > "$O/ldlt_phases.txt" 2>/dev/null
$T python3 tools/bench_iba.py 2>/dev/null | grep LocalInertialBA > "$O/local_inertial_ba_bench.txt"
# this round's extra evidence: phase counts of k_fastw, stage times alone on the chip, quadtree phases (1 and 256 frames), single-frame latency
$T python3 tools/fastw_stats.py 64 > "$O/k_fastw_phase_counts_b64.txt" 2>/dev/null
$T python3 tools/fastw_cycles.py 256 2>/dev/null | grep -v amdgpu.ids > "$O/k_fastw_phase_cycles.txt"
$T python3 tools/fast_threshold_sweep.py > "$O/k_fastw_threshold_sweep.txt" 2>/dev/null
$T python3 tools/pose_opt_modes.py > "$O/pose_optimization_modes.txt" 2>/dev/null
$T python3 tools/stage_times.py 64 > "$O/extract_stage_times_isolated.txt" 2>/dev/null; python3 tools/stage_times.py 256 >> "$O/extract_stage_times_isolated.txt" 2>/dev/null
fi   # (MORB_REFRESH_TAIL)
{ echo "== 1920 x 1080 / 4000, 1 stereo frame (team of 16 waves per big level)"; MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/fast_phases.py 1 | tail -22;
  echo "== 1920 x 1080 / 4000, 4 stereo frames (team)"; MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/fast_phases.py 4 | tail -22;
  echo "== 752 x 480 / 1200, 1 stereo frame (team)"; $T python3 tools/fast_phases.py 1 | tail -22;
  echo "== 752 x 480 / 1200, 256 stereo frames (one wave per level, the bench's packing)"; $T python3 tools/fast_phases.py 256 | tail -22;
  echo "== the same four with the sweeps one by one (-DQT_FAST_FORWARD=0: round 5's algorithm on this round's team width)";
  export MORB_EXTRA_DEFS=-DQT_FAST_FORWARD=0 MORB_TIMING_TAG=_noff;
  MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/fast_phases.py 1 | tail -22; MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/fast_phases.py 4 | tail -22; $T python3 tools/fast_phases.py 1 | tail -22; $T python3 tools/fast_phases.py 256 | tail -22;
  echo "== and with round 5's team of four waves (-DQT_FAST_FORWARD=0 -DQT_TEAM_WAVES=4)";
  export MORB_EXTRA_DEFS="-DQT_FAST_FORWARD=0 -DQT_TEAM_WAVES=4" MORB_TIMING_TAG=_r05;
  MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/fast_phases.py 1 | tail -22; $T python3 tools/fast_phases.py 1 | tail -22;
  unset MORB_EXTRA_DEFS MORB_TIMING_TAG; } > "$O/k_distribute_phases.txt" 2>/dev/null
{ $T python3 tools/latency_b1.py; MORB_W=1920 MORB_H=1080 MORB_NF=4000 $T python3 tools/latency_b1.py; } 2>/dev/null | grep -v amdgpu.ids > "$O/latency_b1.txt"
# the other headline shapes: BASELINE configs[3] on one GPU, and the --gpus 2 launcher path (two ranks sharing this GPU, gloo)
$T python3 bench.py --workload c4 --no-extras --no-cpu-baseline --sustained-s 0 > "$O/bench_c4.json" 2>/dev/null
$T python3 bench.py --workload c4 --batch 1 --no-cpu-baseline > "$O/bench_c4_batch1.json" 2>/dev/null     # one 1920 x 1080 stereo frame per step: what each rank of the 8-GPU run executes
# the three placements of a step's matchers, each twice
for rep in 1 2; do for m in beside-pyramid under-quadtree under-fast; do python3 bench.py --matchers $m --no-extras --no-cpu-baseline --steps 40 --warmup 5 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('$m', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms |', ' '.join(f'{k} {v:.3f}' for k,v in s.items()))
"; done; done > "$O/matcher_placement_ab.txt"
$T bash tools/r06_pyr_chunk.sh > "$O/pyramid_chunk_sweep.txt" 2>/dev/null
# does v_mfma_f64_4x4x4 add its four products in order, each rounded?  (the probe the edge-order PoseOptimization rests on; the handle repeats a short form of it at creation)
{ hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/micro/mfma_chain.hip 2>&1 && /tmp/mfma_chain; } > "$O/mfma_chain_probe.txt" 2>&1
MORB_DIST_BACKEND=gloo $T python3 bench.py --gpus 2 --batch 64 --steps 10 --no-extras --no-cpu-baseline --sustained-s 0 > "$O/bench_gpus2_gloo_one_gpu.json" 2>/dev/null
MORB_DIST_BACKEND=gloo $T python3 bench.py --gpus 2 --batch 64 --steps 10 --exchange allgather --no-extras --no-cpu-baseline --sustained-s 0 > "$O/bench_gpus2_allgather_gloo_one_gpu.json" 2>/dev/null
MORB_DIST_BACKEND=gloo $T python3 bench.py --gpus 2 --workload c4 --steps 10 --no-extras --no-cpu-baseline --sustained-s 0 > "$O/bench_c4_gpus2_gloo_one_gpu.json" 2>/dev/null
# the tracking-side matchers (north_star's SearchByProjection / SearchForTriangulation kernels): kernel stats of the chain + keyframe searches,
# kernel timelines of one step (one frame, 256 frames).  (No eight-rank one-GPU runs any more: not evidence of anything)
$T bash tools/prof_tracking.sh trk_$R 256 > "$O/tracking_profile_log.txt" 2>&1
cp gpurun_out/trk_$R/tracking_kernel_stats.csv gpurun_out/trk_$R/tracking_b1_timeline.txt gpurun_out/trk_$R/tracking_b256_timeline.txt "$O/" 2>/dev/null
cp gpurun_out/trk_$R/bench.json "$O/tracking_bench.json" 2>/dev/null
# one stereo frame through the front end, kernel by kernel (durations and the gaps in front of them), at configs[3]'s and configs[1]'s shapes
for shape in "1920 1080 4000" "752 480 1200"; do set -- $shape
  MORB_W=$1 MORB_H=$2 MORB_NF=$3 $T rocprofv3 --kernel-trace -d "$O/b1trace_$1" -o t --output-format csv -- python3 tools/latency_b1.py > /dev/null 2>&1
  { echo "== $1 x $2 / $3 features, one stereo frame (rocprofv3 --kernel-trace of tools/latency_b1.py; the profiler adds ~10 us of gaps at the stage events)"; python3 tools/b1_trace.py $(find "$O/b1trace_$1" -name "t_kernel_trace.csv" | head -1); } >> "$O/frontend_one_frame_timeline.txt" 2>&1
  rm -rf "$O/b1trace_$1"
done
$T python3 tools/ablate_matchers.py > "$O/matcher_ablation.txt" 2>/dev/null
$T python3 tools/h2d_bw.py > "$O/h2d_copy_bandwidth_by_streams.txt" 2>/dev/null
$T python3 bench.py > "$O/bench_default.json" 2>/dev/null
$T python3 tools/time_stats.py "$O" 50 10 > /dev/null 2>&1
rm -rf "$O/stats" "$O"/pmc/*/q_kernel_trace.csv "$O"/pmc/*/q_agent_info.csv "$O/pmc"
ls -la "$O"
