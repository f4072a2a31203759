timeout 900 python -m pytest tests/test_extractor_gpu.py tests/test_bench_chain_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python tools/stage_times.py 256 2>/dev/null | tail -1 | sed 's/^/wide   /'
MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_fwnarrow.so python tools/stage_times.py 256 2>/dev/null | tail -1 | sed 's/^/narrow /'
done
python bench.py --no-extras --no-cpu-baseline --sustained-s 0 --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide  ', d['value'], d['roofline']['avg_launch_ms'])"
MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_fwnarrow.so python bench.py --no-extras --no-cpu-baseline --sustained-s 0 --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('narrow', d['value'], d['roofline']['avg_launch_ms'])"
