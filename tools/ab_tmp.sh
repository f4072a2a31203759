timeout 300 python tools/bench_tracking.py 256 10 | tail -1
timeout 300 python tools/bench_tracking.py 1024 5 | tail -1 | cut -c1-100
