"""The BENCHED chain against the oracle: the exact object bench.py times (morb_slam_amd/frontend.py StereoFrontEnd — two buffer
sets, pipelined streams, the k=10 / L=6 / levelsup=4 vocabulary, the has_mp mask, bench.py's default batch of stereo frames per step) is
run for three steps and sampled frames spread over the batch are compared with the CPU oracle field by field: keypoint records,
descriptors, mvuRight / mvDepth bit patterns, BoW word + node ids, the SearchByBoW table and count (tests/chain_check.py).

Reference: ORBextractor.cc:1006-1086, Frame.cc:889-1047, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259,
ORBmatcher.cc:218-395."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _front_end(B, **kw):
    import torch
    import bench
    from morb_slam_amd.frontend import StereoFrontEnd
    host = bench.make_batch(list(range(B)), B, seed=0).reshape(2 * B, bench.H, bench.W)
    images = torch.from_numpy(host).cuda()
    return StereoFrontEnd(images, 1200, B, vocab=(10, 6, 4), **kw), host


def test_bench_step_matches_oracle():
    import bench
    import chain_check
    B = bench.WORKLOADS["c2"][3]      # the batch bench.py runs by default (512 stereo frames = 1024 images per launch)
    assert B >= 128
    fe, host = _front_end(B)
    sets = [fe.step() for _ in range(3)]
    fe.sync()
    # step 3 wrote buffer set 0 again (its readers of step 1 were waited for), step 2 wrote set 1: both are checked
    frames = sorted(set(int(x) for x in np.linspace(0, B - 1, 8)) | {1, B // 2 + 1})
    assert chain_check.verify_frames(fe, sets[2], frames, host) == len(frames)
    assert chain_check.verify_frames(fe, sets[1], [0, 37, B - 1], host) == 3
    assert sets[2] is sets[0] and sets[1] is not sets[0]
    nm = sets[2].match_out[1].cpu().numpy()
    assert nm[1:].mean() > 50                     # frames really match their predecessors
    fe.close()


@pytest.mark.parametrize("kw", [dict(matchers="under-quadtree"), dict(matchers="under-fast"), dict(extract_streams=2), dict(nset=1)])
def test_other_schedules_match_oracle(kw):
    """bench.py's other schedules (--matchers under-quadtree | under-fast, --extract-streams 2, --no-pipeline) produce the same bytes."""
    import chain_check
    B = 32
    fe, host = _front_end(B, **kw)
    for _ in range(3):
        S = fe.step()
    fe.sync()
    assert chain_check.verify_frames(fe, S, [0, 13, B - 1], host) == 3
    fe.close()


@pytest.mark.multiprocess
def test_bench_line_reports_verified_frames_and_sustained(tmp_path):
    """bench.py's own post-region self-check and the sustained block (short run)."""
    from procs import describe, spawn
    p = spawn([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "16", "--steps", "3", "--warmup", "1", "--no-extras",
               "--no-cpu-baseline", "--verify-frames", "3", "--sustained-s", "0.3"], dict(os.environ), tmp_path / "log", timeout=900)
    assert p.returncode == 0, describe([p])
    line = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][-1]
    assert line["verified_frames"] == 3 and line["verified"]["frames"] == [0, 7, 15]
    s = line["sustained"]
    assert s["steps"] >= 1000 and s["window_min"] <= s["window_median"] <= s["window_max"] and s["value"] > 0
