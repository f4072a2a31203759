MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_srchcyc.so python tools/bench_tracking.py 1 5 search-cycles 2>/dev/null | tail -1
timeout 300 python tools/bench_tracking.py 256 10 | tail -1 | cut -c1-420
