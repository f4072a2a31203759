#!/bin/bash
# round 6: pyramid launches chunked over groups of images (MORB_PYR_CHUNK = images per group; the bench step has 1024 images)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for c in 0 64 128 256 512; do
  export MORB_PYR_CHUNK=$c
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('chunk $c', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms |', ' '.join(f'{k} {v:.3f}' for k,v in s.items()))
"
done
done
