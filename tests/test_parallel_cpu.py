"""world_size-2 gloo test (CPU) of the multi-GPU frame sharding + feature all-gather (morb_slam_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest

from morb_slam_amd import parallel


def test_round_robin_index_math():
    world, S = 4, 3
    seen = set()
    for r in range(world):
        kf, fr = parallel.predecessor_pairs(r, world, S)
        for s in range(S):
            g = parallel.global_frame(r, world, s)
            assert parallel.owner(world, g) == (r, s)
            assert fr[s] == r * S + s
            gp = max(g - 1, 0)
            assert kf[s] == (gp % world) * S + gp // world
            seen.add(g)
    assert seen == set(range(world * S))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, cap = 3, 17
        g = torch.Generator().manual_seed(100 + rank)
        kps = torch.randint(0, 256, (S, cap, 28), dtype=torch.uint8, generator=g)
        desc = torch.randint(0, 256, (S, cap, 32), dtype=torch.uint8, generator=g)
        cnt = torch.tensor([cap - rank - s for s in range(S)], dtype=torch.int32)
        node = torch.randint(0, 100, (S, cap), dtype=torch.int32, generator=g)
        ex = parallel.FeatureExchange()
        pk, pd, pc, pn = ex.exchange(kps, desc, cnt, node)
        # rebuild every rank's slab locally and compare
        ok = True
        for r in range(world):
            gg = torch.Generator().manual_seed(100 + r)
            k2 = torch.randint(0, 256, (S, cap, 28), dtype=torch.uint8, generator=gg)
            d2 = torch.randint(0, 256, (S, cap, 32), dtype=torch.uint8, generator=gg)
            c2 = torch.tensor([cap - r - s for s in range(S)], dtype=torch.int32)
            n2 = torch.randint(0, 100, (S, cap), dtype=torch.int32, generator=gg)
            ok &= bool((pk[r * S:(r + 1) * S] == k2).all() and (pd[r * S:(r + 1) * S] == d2).all()
                       and (pc[r * S:(r + 1) * S] == c2).all() and (pn[r * S:(r + 1) * S] == n2).all())
        kf, fr = parallel.predecessor_pairs(rank, world, S)
        # the predecessor of local slot s (global g) must be the slab row of global g-1
        for s in range(S):
            gidx = parallel.global_frame(rank, world, s)
            gp = max(gidx - 1, 0)
            r2, s2 = parallel.owner(world, gp)
            ok &= int(pc[kf[s]]) == cap - r2 - s2 and int(pc[fr[s]]) == cap - rank - s
        # weak-scaling timing reduction used by bench.py: MAX over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok &= float(t) == float(world)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_feature_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_neighbour_pairs_index_math():
    # pool = [own slab; previous rank's slab]: the predecessor of (rank r, slot s) is (r - 1, s), or (last rank, s - 1) for r = 0
    for world, S in ((1, 4), (2, 3), (4, 3), (8, 2)):
        for r in range(world):
            kf, fr = parallel.neighbour_pairs(r, world, S)
            assert fr.tolist() == list(range(S))
            for s in range(S):
                g = parallel.global_frame(r, world, s)
                if g == 0:
                    assert kf[s] == s
                    continue
                pr, ps = parallel.owner(world, g - 1)
                if world == 1:
                    assert kf[s] == ps
                else:
                    assert pr == (r - 1) % world and kf[s] == S + ps


def _ring_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, cap = 3, 17
        def slab(r):
            g = torch.Generator().manual_seed(100 + r)
            return (torch.randint(0, 256, (S, cap, 28), dtype=torch.uint8, generator=g), torch.randint(0, 256, (S, cap, 32), dtype=torch.uint8, generator=g),
                    torch.tensor([cap - r - s for s in range(S)], dtype=torch.int32), torch.randint(0, 100, (S, cap), dtype=torch.int32, generator=g))
        ex = parallel.NeighbourExchange()
        ok = True
        for _ in range(2):     # pools are reused from call to call
            pools = ex.exchange(*slab(rank))
            own, prev = slab(rank), slab((rank - 1) % world)
            for pool, a, b in zip(pools, own, prev):
                ok &= bool((pool[:S] == a).all() and (pool[S:] == b).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_neighbour_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1] and all(ok for _, ok in res)


def _run_bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr.decode()


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 3` (no torchrun, WORLD_SIZE unset — the driver's command shape) must start 3 rank processes that
    find each other; --launch-probe stops them after the rendezvous, before anything needs a GPU."""
    rc, lines, err = _run_bench(["--gpus", "3", "--launch-probe"])
    assert rc == 0, err
    assert len(lines) == 1                        # ONE JSON line, from rank 0, relayed by the launcher
    assert lines[0]["n_gpus"] == 3 and lines[0]["ranks_counted"] == 3 and lines[0]["rank_sum"] == 0 + 1 + 2


def test_bench_under_external_launcher_does_not_respawn():
    """Under torchrun (WORLD_SIZE set by the launcher) bench.py is a rank, not a launcher."""
    port = _free_port()
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-probe"],
                              env=dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port)), stdout=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert '"n_gpus": 2' in outs[0] and "{" not in outs[1]


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="needs a box WITHOUT a GPU: the ranks must fail")
def test_bench_launcher_propagates_rank_failure():
    """Without a GPU every rank dies in torch.cuda.set_device: the launcher must return non-zero and print no JSON line."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                                env_extra={"MORB_DIST_BACKEND": "gloo"})
    assert rc != 0 and not lines
    assert "rank" in err
