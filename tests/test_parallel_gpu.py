"""Frame sharding on a GPU: two ranks (gloo, sharing GPU 0) run the front end on a stream dealt round-robin, ship their left-image
features one rank up the ring and match every frame against its predecessor; the SearchByBoW tables must equal those of the same
stream on one rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_match_one_rank(tmp_path):
    total = 6
    one, two = tmp_path / "w1", tmp_path / "w2"
    one.mkdir(); two.mkdir()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    subprocess.check_call([sys.executable, os.path.join(HERE, "dist_stream_worker.py"), str(one), str(total)], env=env)
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_stream_worker.py"), str(two), str(total)],
                              env=dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                                       MORB_DIST_BACKEND="gloo")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    ref = np.load(one / "rank0.npz")
    by_g = {int(g): (ref["match"][i], int(ref["nmatch"][i]), int(ref["count"][i])) for i, g in enumerate(ref["gids"])}
    seen = set()
    for r in range(2):
        d = np.load(two / f"rank{r}.npz")
        for i, g in enumerate(d["gids"]):
            m, n, c = by_g[int(g)]
            assert int(d["count"][i]) == c
            assert int(d["nmatch"][i]) == n, (g, int(d["nmatch"][i]), n)
            np.testing.assert_array_equal(d["match"][i], m, err_msg=f"global frame {g}")
            seen.add(int(g))
    assert seen == set(range(total))
    assert sum(v[1] for v in by_g.values()) > 100          # the frames really match their predecessors
