"""Frame sharding on a GPU: two ranks (gloo, sharing GPU 0) run the front end on a stream dealt round-robin, ship their left-image
features one rank up the ring and match every frame against its predecessor; the SearchByBoW tables must equal those of the same
stream on one rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from procs import describe, run_ranks, spawn

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


WORKER = os.path.join(HERE, "dist_stream_worker.py")


def _one_rank_then_world(tmp_path, dims, total, exchange, world):
    """The stream on one rank, then on `world` ranks sharing GPU 0 (gloo): tests/procs.py keeps every rank in its own process group, kills
    them all on the way out and keeps their stderr."""
    one, two = tmp_path / "w1", tmp_path / "w2"
    one.mkdir(); two.mkdir()
    extra = [str(x) for x in dims] + [exchange]
    r1 = spawn([sys.executable, WORKER, str(one), str(total)] + extra, dict(os.environ, WORLD_SIZE="1", RANK="0"), tmp_path / "log1", timeout=600)
    assert r1.returncode == 0, describe([r1])
    port = str(_free_port())
    res = run_ranks([([sys.executable, WORKER, str(two), str(total)] + extra,
                      dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                           MORB_DIST_BACKEND="gloo")) for r in range(world)], tmp_path / "logw", timeout=600)
    assert all(r.returncode == 0 for r in res), describe(res)
    return one, two


# 640 x 480 / 600 features / 6 frames through every transport; and BASELINE configs[3]'s size (1920 x 1080 / 4000 features, 4 frames) through the
# two the bench offers (--exchange ring | allgather): the exchanged 4000-feature slabs must give the tables of the world-1 run
@pytest.mark.multiprocess
@pytest.mark.parametrize("dims,total,exchange,world", [((640, 480, 600), 6, "ring", 2), ((640, 480, 600), 6, "ring4", 2), ((640, 480, 600), 6, "allgather", 2),
                                                       ((1920, 1080, 4000), 4, "ring", 2), ((1920, 1080, 4000), 4, "allgather", 2)])
def test_ranks_match_one_rank(tmp_path, dims, total, exchange, world):
    _check_tables(*_one_rank_then_world(tmp_path, dims, total, exchange, world), total, world)


# EIGHT ranks sharing GPU 0, two slots per rank: the only layout in which rank 0's second frame takes its predecessor from the LAST rank's
# previous slot (the ring's wrap) on device data.  The wrap does not depend on the image size, so these run at 640 x 480 / 600 features: nine
# processes at 1920 x 1080 on one time-sliced device is the regime in which round 5's driver run aborted (profiles/r06/README.md), and it
# says nothing about the exchange that the small shape does not.  configs[3]'s SIZE is covered at world 2 above.
@pytest.mark.multiprocess
@pytest.mark.manyranks
@pytest.mark.parametrize("exchange", ["ring", "allgather"])
def test_eight_ranks_ring_wrap(tmp_path, exchange):
    _check_tables(*_one_rank_then_world(tmp_path, (640, 480, 600), 16, exchange, 8), 16, 8)


def _check_tables(one, two, total, world):
    ref = np.load(one / "rank0.npz")
    by_g = {int(g): (ref["match"][i], int(ref["nmatch"][i]), int(ref["count"][i])) for i, g in enumerate(ref["gids"])}
    seen = set()
    for r in range(world):
        d = np.load(two / f"rank{r}.npz")
        for i, g in enumerate(d["gids"]):
            m, n, c = by_g[int(g)]
            assert int(d["count"][i]) == c
            assert int(d["nmatch"][i]) == n, (g, int(d["nmatch"][i]), n)
            np.testing.assert_array_equal(d["match"][i], m, err_msg=f"global frame {g}")
            seen.add(int(g))
    assert seen == set(range(total))
    assert sum(v[1] for v in by_g.values()) > 100          # the frames really match their predecessors


def _bench_line(tmp_path, args, gloo=True, timeout=900):
    """One `python bench.py ...` child (tests/procs.py: own process group, stderr kept) -> its ONE JSON line."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    if gloo:
        env["MORB_DIST_BACKEND"] = "gloo"
    r = spawn([sys.executable, os.path.join(root, "bench.py")] + args + ["--no-extras", "--no-cpu-baseline"], env, tmp_path / ("bench_" + "_".join(a.strip("-") for a in args[:6])),
              timeout=timeout)
    assert r.returncode == 0, describe([r])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


@pytest.mark.multiprocess
@pytest.mark.parametrize("workload", ["c2", "c4"])
def test_bench_gpus2_launches_two_ranks(tmp_path, workload):
    """`python bench.py --gpus 2` (the driver's command shape, WORLD_SIZE unset) spawns two ranks; they share GPU 0 here, so
    the exchange goes through gloo.  The world > 1 branch of bench.py — ring exchange, cross-frame SearchByBoW on the
    [own; received] pool, MAX-over-ranks timing — must run and produce matches, for BASELINE configs[1] and configs[3]."""
    line = _bench_line(tmp_path, ["--gpus", "2", "--workload", workload, "--batch", "4", "--steps", "2", "--warmup", "1"])
    assert line["n_gpus"] == 2
    assert "feature_exchange" in line["config"]["stages_in_step"]
    assert line["config"]["mean_bow_matches_per_frame"] > 0
    assert line["config"]["mean_stereo_matches_per_frame"] > 0
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    want = (752, 1200) if workload == "c2" else (1920, 4000)
    assert f"{want[0]}x" in line["metric"] and f"{want[1]} feat" in line["metric"]


@pytest.mark.multiprocess
@pytest.mark.manyranks
@pytest.mark.parametrize("ex", ["ring", "allgather"])
def test_bench_gpus8_one_frame_per_rank(tmp_path, ex):
    """The launch SHAPE of BASELINE configs[3] — eight ranks, ONE stereo frame per rank and step — through bench.py's own launcher (ranks share GPU 0,
    gloo), ring and all-gather: the line carries the per-rank step-time spread and the exchange's HIP-event time.  Functional coverage of the launcher
    and of `multi_gpu`, so at 640 x 480 / 600 features (`--workload vga`): eight 1080p ranks time-slicing one device measure nothing (round 5: 8.2 s per
    step) and are the regime the round-5 driver run aborted in; configs[3]'s image size runs at world 2 in test_bench_gpus2_launches_two_ranks[c4]."""
    line = _bench_line(tmp_path, ["--gpus", "8", "--workload", "vga", "--batch", "1", "--exchange", ex, "--steps", "3", "--warmup", "0"], timeout=1200)
    assert line["n_gpus"] == 8 and line["config"]["stereo_frames_per_step_per_gpu"] == 1
    assert "NOT a BASELINE configuration" in line["metric"]
    mg = line["multi_gpu"]
    assert mg["exchange"] == ex and mg["rank_ms_per_step"]["min"] > 0 and mg["rank_ms_per_step"]["max"] >= mg["rank_ms_per_step"]["min"]
    assert mg["exchange_ms_per_step"]["max_over_ranks_of_mean"] > 0
    assert line["config"]["mean_bow_matches_per_frame"] > 0


@pytest.mark.multiprocess
def test_bench_matcher_placements_agree(tmp_path):
    """bench.py's schedules — a step's matchers right behind its extraction, or held back behind the NEXT extraction's after-FAST event
    (morb_extractor_event_after_fast) — process the same frames: same stereo and BoW match counts, every step's matchers inside the region."""
    got = {m: _bench_line(tmp_path, ["--matchers", m, "--batch", "8", "--steps", "3", "--warmup", "1"], gloo=False)["config"]
           for m in ("beside-pyramid", "under-quadtree", "under-fast")}
    for k in ("mean_keypoints_per_image", "mean_stereo_matches_per_frame", "mean_bow_matches_per_frame"):
        assert got["beside-pyramid"][k] == got["under-quadtree"][k] == got["under-fast"][k] and got["beside-pyramid"][k] > 0, k


@pytest.mark.multiprocess
def test_a_rank_without_its_peer_times_out_with_a_non_zero_exit(tmp_path):
    """Every wait on a peer is bounded: a rank whose partner never shows up exits non-zero after the process-group timeout instead of hanging
    (a fresh child process; nothing re-execs)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), MORB_DIST_BACKEND="gloo",
               MORB_DIST_TIMEOUT_S="8")
    r = spawn([sys.executable, WORKER, str(tmp_path), "4"], env, tmp_path / "log", timeout=300)
    assert r.returncode != 0


def test_feature_slab_round_trip():
    """morb_feature_slab_pack -> (one contiguous buffer: what hipMemcpyPeerAsync / an RCCL send moves) -> morb_feature_slab_unpack."""
    import torch
    from morb_slam_amd import ORBmatcher
    m = ORBmatcher()
    rng = np.random.default_rng(0)
    nimg, cap, S = 10, 333, 5
    cu = lambda a: torch.from_numpy(a).cuda()
    kps = cu(rng.integers(0, 256, (nimg, cap, 28), dtype=np.uint8)); desc = cu(rng.integers(0, 256, (nimg, cap, 32), dtype=np.uint8))
    node = cu(rng.integers(0, 1000, (nimg, cap), dtype=np.int32)); cnt = cu(rng.integers(0, cap, nimg).astype(np.int32))
    rows = torch.arange(0, nimg, 2, dtype=torch.int32, device="cuda")
    slab = m.pack_slab(kps, desc, cnt, node, rows=rows)
    assert slab.numel() == S * cap * 64 + S * 4 == m.slab_bytes(S, cap)
    k2, d2, n2, c2 = torch.zeros_like(kps), torch.zeros_like(desc), torch.zeros_like(node), torch.zeros_like(cnt)
    dst = torch.tensor([9, 1, 4, 0, 7], dtype=torch.int32, device="cuda")
    m.unpack_slab(slab, S, k2, d2, c2, n2, rows=dst)
    torch.cuda.synchronize()
    for f in range(S):
        a, b = int(rows[f]), int(dst[f])
        assert torch.equal(k2[b], kps[a]) and torch.equal(d2[b], desc[a]) and torch.equal(n2[b], node[a]) and int(c2[b]) == int(cnt[a])
    assert int(k2[2].sum()) == 0      # rows that are not destinations stay untouched


@pytest.mark.multiprocess
def test_rccl_world_of_one():
    """The "nccl" backend (RCCL) on this box's one GPU: communicator creation, barrier, MAX all-reduce and the feature all-gather the
    multi-GPU path calls, in a fresh process (tests/rccl_world1_worker.py).  No second GPU, so no xGMI transfer — but librccl runs."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_worker.py"), str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "rccl world-1 ok" in p.stdout


@pytest.mark.multiprocess
def test_cpp_shard_ring_without_torch(tmp_path):
    """INTEGRATION.md section 5 compiled and run: a C++ caller (g++, libmorb_hip + the HIP runtime, no torch) deals a stream over 1 / 2 / 3
    "GPUs" (sets of handles and streams on this box's one device), ships one feature slab per GPU with hipMemcpyPeerAsync and matches every frame
    against its predecessor; the program itself requires identical tables for the three world sizes, and the tables must also equal what the
    Python mirror computes for the whole stream in one batch (and, on two sampled frames, the CPU oracle's SearchByBoW)."""
    import torch
    import oracle_lib as O
    from morb_slam_amd import ORBextractor, ORBmatcher
    from morb_slam_amd.synth import make_stereo_pair, make_vocabulary, shift_image
    root = os.path.dirname(HERE)
    W, H, NF, F, VK, VL, VUP = 640, 480, 600, 6, 10, 3, 1
    l, r = make_stereo_pair(W, H, seed=11)
    imgs = np.stack([np.stack([shift_image(l, 3 * g, 2 * g), shift_image(r, 3 * g, 2 * g)]) for g in range(F)])     # [F][2][H][W]
    vd, vf = make_vocabulary(VK, VL, seed=2)
    d = tmp_path / "io"
    d.mkdir()
    np.array([W, H, NF, F, VK, VL, VUP], np.int32).tofile(str(d / "dims.bin"))
    imgs.tofile(str(d / "imgs.bin")); vd.tofile(str(d / "voc_desc.bin")); vf.tofile(str(d / "voc_first.bin"))
    exe = str(tmp_path / "shard_ring_check")
    libdir = os.path.join(root, "morb_slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-o", exe,
                           os.path.join(root, "tests", "native", "shard_ring_check.cc"), "-L" + libdir, "-lmorb_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical for 1 / 2 / 3 GPUs" in out.stdout
    # ---- the same stream through the Python mirror, one batch on one rank
    dev = torch.device("cuda", 0)
    ext = ORBextractor(NF, 1.2, 8, 20, 7)
    m = ORBmatcher(0.7, True)
    kps, desc, cnt, _ = ext.extract_batch(torch.from_numpy(imgs.reshape(2 * F, H, W)).to(dev))
    cap = kps.shape[1]
    cl = cnt.clone(); cl[1::2] = 0
    word, node = m.bow_transform(desc, cl, torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev), VK, VL, VUP)
    has = np.zeros((2 * F, cap), np.uint8)
    ii = np.arange(cap)
    for g in range(F):
        has[2 * g] = ((g * 131 + ii * 7) % 5) != 0
    kf = torch.tensor([2 * max(g - 1, 0) for g in range(F)], dtype=torch.int32, device=dev)
    fr = torch.tensor([2 * g for g in range(F)], dtype=torch.int32, device=dev)
    mt, nm = m.SearchByBoW(kf, fr, kps, desc, node, cl, torch.from_numpy(has).to(dev))
    torch.cuda.synchronize()
    got = np.fromfile(str(d / "out_match.bin"), np.int32).reshape(F, cap)
    gotn = np.fromfile(str(d / "out_nmatch.bin"), np.int32)
    np.testing.assert_array_equal(gotn, nm.cpu().numpy())
    np.testing.assert_array_equal(got, mt.cpu().numpy())
    assert int(gotn[1:].sum()) > 100
    # ---- and two of the frames against the CPU oracle (the restated ORBmatcher::SearchByBoW on the restated extraction)
    K, D, C, ND = kps.cpu().numpy(), desc.cpu().numpy(), cnt.cpu().numpy(), node.cpu().numpy()
    from morb_slam_amd.capi import KP_DTYPE
    for g in (1, F - 1):
        a, b = 2 * (g - 1), 2 * g
        ka = K[a, :C[a]].view(KP_DTYPE).reshape(-1); kb = K[b, :C[b]].view(KP_DTYPE).reshape(-1)
        on, om = O.search_by_bow(D[a, :C[a]], ka["angle"], has[a, :C[a]].astype(bool), ND[a, :C[a]], D[b, :C[b]], kb["angle"], ND[b, :C[b]], 0.7, True)
        assert on == int(gotn[g])
        np.testing.assert_array_equal(got[g, :C[b]], om)
