cd "$GRAFT_REPO_ROOT"
for b in 64 128 256 512; do
  for mid in 0 2048; do
    MORB_TEAM_MID=$mid python3 bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-verify --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['extract_stage_ms_per_step']
        print('c2 batch $b mid $mid', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms | distribute', round(s['distribute'],3), 'blur', round(s['blur'],3))
"
  done
done
