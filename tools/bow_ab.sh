for cfg in "MORB_BOW_GLOBAL=1" "MORB_BOW_STAGE=3 MORB_BOW_GRID=768" "MORB_BOW_STAGE=3 MORB_BOW_GRID=1536" "MORB_BOW_STAGE=2 MORB_BOW_GRID=1024" "MORB_BOW_STAGE=2 MORB_BOW_GRID=2048" "MORB_BOW_STAGE=2 MORB_BOW_GRID=4096"; do
  echo "== $cfg"; env $cfg python tools/bow_transform_time.py
  env $cfg python bench.py --no-cpu-baseline --no-extras --no-verify --sustained-s 0 --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'])"
done
