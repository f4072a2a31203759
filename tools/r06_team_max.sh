#!/bin/bash
# round 6: up to how many images per call does the team packing of k_distribute (16 waves per big level) beat one wave per level?
cd "$GRAFT_REPO_ROOT"
for wl in c2 c4; do
for b in 8 16 32 64 128; do
  if [ $wl = c4 ] && [ $b -gt 32 ]; then continue; fi
  for n in product team64 team256; do
    if [ "$n" = product ]; then unset MORB_HIP_LIB; else export MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_$n.so; fi
    python3 bench.py --workload $wl --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('$wl batch $b $n', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms | distribute', round(s['distribute'],3), 'blur', round(s['blur'],3))
"
  done
done
done
