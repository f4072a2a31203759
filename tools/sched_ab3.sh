#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in "" "1"; do
  MORB_EXP_BSTREAM=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('[bstream=$v]', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms |', ' '.join(f'{k} {v:.2f}' for k,v in s.items()))
"
done
done
