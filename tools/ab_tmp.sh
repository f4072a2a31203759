timeout 900 python -m pytest tests/test_optimizer_gpu.py tests/test_tracking_gpu.py tests/test_fisheye_gpu.py tests/test_stress_gpu.py -m gpu -x -q 2>&1 | tail -2
for n in 600 1600 2000 4000; do echo "features $n"; python tools/pose_opt_modes.py $n 2>/dev/null | grep "exact"; done
