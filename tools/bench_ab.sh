#!/bin/bash
# bench headline A/B through gpurun: bash tools/bench_ab.sh <variant|product> ...   (each library twice, alternating)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = product ]; then unset MORB_HIP_LIB; else export MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_$n.so; fi
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('$n', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms |', ' '.join(f'{k} {v:.3f}' for k,v in s.items()))
"
done
done
