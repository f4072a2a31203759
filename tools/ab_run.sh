#!/bin/bash
# A/B timing of library variants built by tools/ab_build.py (run through gpurun): bash tools/ab_run.sh <name> [<name> ...]
# prints the isolated per-stage times of the C2 extraction (512 images) for the product library and every variant.
cd "$GRAFT_REPO_ROOT"
echo "product: $(python3 tools/stage_times.py 256 20 2>/dev/null | tail -1)"
for n in "$@"; do
  echo "$n: $(MORB_HIP_LIB=$PWD/morb_slam_amd/libmorb_hip_$n.so python3 tools/stage_times.py 256 20 2>/dev/null | tail -1)"
done
