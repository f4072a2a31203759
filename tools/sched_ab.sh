#!/bin/bash
# schedule A/B through gpurun: bash tools/sched_ab.sh <variant|product> ...   (bench headline in three schedules per library)
cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  if [ "$n" = product ]; then unset MORB_HIP_LIB; else export MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_$n.so; fi
  for sched in "" "--extract-streams 2" "--extract-streams 2 --stagger"; do
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 $sched 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('$n [$sched]', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms |', ' '.join(f'{k} {v:.2f}' for k,v in s.items()))
"
  done
done
