"""Worker of tests/test_parallel_gpu.py: one rank of a `world`-rank run of the frame-sharded front end on a small synthetic stream
(extract -> ComputeBoW -> feature exchange -> SearchByBoW against the previous frame).  Writes, per LOCAL frame, its global id and
its SearchByBoW match table to <out>/rank<r>.npz.  Ranks may share one GPU (MORB_DIST_BACKEND=gloo).
    dist_stream_worker.py <out> <total frames> [width height nfeat] [ring|ring4|allgather]
ring = one slab per step (morb_feature_slab_pack / _unpack), ring4 = one transfer per array, allgather = every rank's slabs pooled."""
import datetime
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def run(rank, world, total, out_dir, width=640, height=480, nfeat=600, exchange="ring"):
    import torch
    from morb_slam_amd import ORBextractor, ORBmatcher, parallel
    from morb_slam_amd.synth import make_stereo_pair, make_vocabulary, shift_image
    dev = torch.device("cuda", 0)
    S = total // world
    gids = [parallel.global_frame(rank, world, s) for s in range(S)]
    base, _ = make_stereo_pair(width, height, seed=5)
    imgs = np.stack([shift_image(base, 3 * g, 2 * g) for g in gids])            # left images of this rank's frames
    ext = ORBextractor(nfeat, 1.2, 8, 20, 7, device=0)
    m = ORBmatcher(0.7, True, device=0)
    vd, vf = make_vocabulary(10, 5, seed=0)
    vd, vf = torch.from_numpy(vd).to(dev), torch.from_numpy(vf).to(dev)
    kps, desc, cnt, _ = ext.extract_batch(torch.from_numpy(imgs).to(dev))
    bow = m.bow_transform(desc, cnt, vd, vf, 10, 5, 4)
    cap = kps.shape[1]
    rng = np.random.default_rng(3)
    has_all = (rng.random((total, cap)) < 0.8).astype(np.uint8)                 # per GLOBAL frame, so that every layout sees the same flags
    if world == 1:
        kf = torch.tensor([max(s - 1, 0) for s in range(S)], dtype=torch.int32, device=dev)
        fr = torch.arange(S, dtype=torch.int32, device=dev)
        has = torch.from_numpy(has_all[gids]).to(dev)
        res = m.SearchByBoW(kf, fr, kps, desc, bow[1], cnt, has)
    elif exchange == "allgather":
        ex = parallel.FeatureExchange()
        pk, pd, pc, pn = ex.exchange(kps, desc, cnt, bow[1])
        kfp, frp = parallel.predecessor_pairs(rank, world, S)
        pool_gids = [parallel.global_frame(r, world, s) for r in range(world) for s in range(S)]   # rank-major pool rows
        has = torch.from_numpy(has_all[pool_gids]).to(dev)
        res = m.SearchByBoW(torch.from_numpy(kfp).to(dev), torch.from_numpy(frp).to(dev), pk, pd, pn, pc, has)
    else:
        ex = parallel.NeighbourExchange(matcher=m if exchange == "ring" else None)
        pk, pd, pc, pn = ex.exchange(kps, desc, cnt, bow[1])
        kfp, frp = parallel.neighbour_pairs(rank, world, S)
        prev_gids = [parallel.global_frame((rank - 1) % world, world, s) for s in range(S)]
        has = torch.from_numpy(np.concatenate([has_all[gids], has_all[prev_gids]])).to(dev)
        res = m.SearchByBoW(torch.from_numpy(kfp).to(dev), torch.from_numpy(frp).to(dev), pk, pd, pn, pc, has)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gids=np.array(gids), match=res[0].cpu().numpy(), nmatch=res[1].cpu().numpy(),
             count=cnt.cpu().numpy())


if __name__ == "__main__":
    out_dir, total = sys.argv[1], int(sys.argv[2])
    dims = [int(x) for x in sys.argv[3:6]] if len(sys.argv) >= 6 else [640, 480, 600]
    exchange = sys.argv[6] if len(sys.argv) >= 7 else "ring"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank whose peer never arrives fails after the timeout (non-zero exit) instead of hanging
        dist.init_process_group(os.environ.get("MORB_DIST_BACKEND", "gloo"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("MORB_DIST_TIMEOUT_S", "180"))))
    run(rank, world, total, out_dir, *dims, exchange=exchange)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
