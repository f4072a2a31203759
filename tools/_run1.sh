cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_extractor_gpu.py -x -q 2>&1 | tail -3
python3 tools/stage_times.py 256
python3 tools/fastw_stats.py 64 | head -16
