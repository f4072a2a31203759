timeout 900 python -m pytest tests/test_optimizer_gpu.py tests/test_tracking_gpu.py tests/test_fisheye_gpu.py tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -3
for n in 600; do echo "features $n"; python tools/pose_opt_modes.py $n 2>/dev/null | grep "exact\|tree"; done
timeout 300 python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-420
