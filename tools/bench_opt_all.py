#!/usr/bin/env python3
"""All optimiser micro-benchmarks in one process (profiled by tools/refresh_profiles.sh): PoseOptimization / LocalBA,
the visual-inertial tracking pair, LocalInertialBA."""
import os, runpy, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for script in ("bench_opt.py", "bench_inertial.py", "bench_iba.py"):
    sys.argv = [script]
    runpy.run_path(os.path.join(HERE, script), run_name="__main__")
