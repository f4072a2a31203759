"""Seeded subsets of the randomised parity sweeps (tools/stress_parity.py, tools/stress_optimizers.py) inside `-m gpu`, so the driver's
round-end run sees them: random image sizes / feature counts / threshold pairs / batch sizes, the noise / flat / saturated images that
drive k_fastw's strip mode and the empty-cell paths, odd shapes; random PoseOptimization / LocalBundleAdjustment problem shapes."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_extraction_and_stereo_stress_subset():
    import stress_parity
    log = []
    n, bad = stress_parity.run(10, seed=7, log=log.append, odd=[(83, 97), (333, 217), (1023, 511)], max_pixels=5 * 752 * 480)
    assert n == 18 and bad == 0, "\n".join(l for l in log if "MISMATCH" in l)


def test_optimizer_stress_subset():
    import stress_optimizers
    log = []
    n1, b1 = stress_optimizers.run_pose(24, seed=11, log=log.append)
    n2, b2 = stress_optimizers.run_ba(6, seed=11, log=log.append, max_points=1500)
    assert b1 + b2 == 0, "\n".join(log)


def test_other_pyramids_and_rig_pose_stress_subset():
    """Other pyramids (scale factors 1.1 - 1.5, 3 - 12 levels, 100 - 5000 features: the library may refuse a configuration loudly, never answer it wrongly) and
    PoseOptimization on the KannalaBrandt8 rig in both modes (the deterministic one: iterations and trials of the oracle)."""
    import stress_optimizers
    import stress_parity
    log = []
    n, bad = stress_parity.run_pyramids(12, seed=9, log=log.append)
    assert n >= 8 and bad == 0, "\n".join(l for l in log if "MISMATCH" in l)
    n3, b3 = stress_optimizers.run_pose_rig(12, seed=11, log=log.append)
    assert b3 == 0, "\n".join(log)
