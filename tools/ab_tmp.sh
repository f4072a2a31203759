timeout 600 python -m pytest tests/test_optimizer_gpu.py tests/test_tracking_gpu.py tests/test_fisheye_gpu.py tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 120 python tools/pose_opt_modes.py 2>/dev/null | grep -i "exact\|tree"
timeout 300 python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-100
