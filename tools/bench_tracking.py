"""Tracking chain (TrackWithMotionModel + TrackLocalMap) and LocalMapping's keyframe searches: throughput at B frames per step
and latency one frame at a time.  `python tools/bench_tracking.py [B] [steps]`; under rocprofv3 --kernel-trace --stats for the kernel table."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from morb_slam_amd.tracking import TrackingChain, build_chains  # noqa: E402


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    G = 256
    imgs = bench.make_batch(range(G), G, seed=0)
    ch, ks, host = build_chains(imgs, B=B, npairs=20, seq_len=64)
    out = {"B": B}

    def timed(fn, sync, n):
        for _ in range(3):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t0) / n
    if only == "search-cycles":   # developer build -DMORB_SEARCH_CYCLES (tools/ab_build.py)
        import ctypes as C
        from morb_slam_amd.capi import lib
        one = {k: v[:1] for k, v in host["scene"].items()}
        c1 = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[:1].contiguous(), one)
        c1.step(); c1.sync()
        z = (C.c_ulonglong * 8)(); lib().morb_search_cycles(z)
        c1.step(); c1.sync(); lib().morb_search_cycles(z)
        print("two searches of one frame: grid+stage %d walk %d fixed-point %d outputs %d passes %d thread0: binsearch %d query-loads %d inner %d" % tuple(list(z)))
        return
    if only.startswith("split"):   # the B frames as K sub-batches on K streams (a sub-batch does not wait for another one's slowest frame): python tools/bench_tracking.py 256 10 split4
        K = int(only[5:] or 2)
        subs = []
        for k in range(K):
            a, b = k * B // K, (k + 1) * B // K
            sub = {key: v[a:b] for key, v in host["scene"].items()}
            subs.append(TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[a:b].contiguous(), sub))
        def step_all():
            for c in subs:
                c.step()
        def sync_all():
            for c in subs:
                c.sync()
        dt1 = timed(ch.step, ch.sync, steps)
        dtk = timed(step_all, sync_all, steps)
        print(json.dumps({"B": B, "one_batch_ms": dt1 * 1e3, "one_batch_frames_per_s": B / dt1, "sub_batches": K, "split_ms": dtk * 1e3, "split_frames_per_s": B / dtk}))
        return
    if only == "chain-only":     # (for a kernel trace of one step: tools/trk_trace.py)
        timed(ch.step, ch.sync, steps)
        return
    dt = timed(ch.step, ch.sync, steps)
    out["tracking_ms_per_step"] = dt * 1e3
    out["tracking_frames_per_s"] = B / dt
    out["mean_matches_last"] = float(ch.nmLast.float().mean()); out["mean_matches_local"] = float(ch.nmLocal.float().mean())
    out["mean_inliers"] = float(ch.nInl.float().mean()); out["mean_local_points"] = float(ch.nMP.float().mean())
    dk = timed(ks.step, ks.sync, steps)
    out["keyframe_pairs"] = 20
    out["keyframe_searches_ms"] = dk * 1e3
    # b = 1
    one = {k: v[:1] for k, v in host["scene"].items()}
    c1 = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[:1].contiguous(), one)
    out["tracking_b1_ms"] = timed(c1.step, c1.sync, 50) * 1e3
    c1.close()
    # the optimiser's default mode (tree sums: same poses to ~1e-9 and same flags, LM trial counts within +-2 of the oracle's)
    cht = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight, host["scene"], exact_order=False)
    dtt = timed(cht.step, cht.sync, steps)
    out["tracking_tree_sums_ms_per_step"] = dtt * 1e3; out["tracking_tree_sums_frames_per_s"] = B / dtt
    cht.close()
    c1t = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[:1].contiguous(), one, exact_order=False)
    out["tracking_tree_sums_b1_ms"] = timed(c1t.step, c1t.sync, 50) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
