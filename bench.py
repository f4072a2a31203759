#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: stereo frames/s, ORB extract+match.

Workloads: c2 (default) = BASELINE configs[1], EuRoC-shaped stereo 752x480 / 1200 features (the configuration the metric is
quoted on; the metric string's "1000 feat" is configs[0]'s mono plumbing case); c4 = configs[3], 1920x1080 / 4000 features.
One "step" = one pass of the hot path over one batch of B synthetic stereo frames per GPU that are already resident in HBM:
ORBextractor x2, ComputeStereoMatches, ComputeBoW, SearchByBoW against the previous frame.  N > 1: one process per GPU
(torch.distributed, RCCL), frames dealt round-robin; the previous frame lives on the previous rank, so every step ships the
left-image features one rank up the ring (point-to-point over xGMI, morb_slam_amd/parallel.py) -> weak scaling.  The timed
region is bracketed by barrier + synchronize and the MAX over ranks is reported.  Rank 0 prints ONE JSON line: the contract's keys, `roofline` (HBM, the
longest extraction stage alone on the chip) with `roofline_issue` (what really bounds that kernel: vector-instruction issue), `cpu_baseline` (the oracle on the
host's cores, bounded sample), `latency` (round 6: ONE frame / keyframe per call, host to host — the reference's call pattern — each with the oracle on one thread
beside it), `h2d_inclusive`, `sustained`, `extra_metrics` (optimisers with the PoseOptimization path that ran, tracking chain, the other BASELINE configs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4] [--batch B] [--matchers beside-pyramid|under-quadtree|under-fast]
                    [--exchange ring|allgather] [--no-cpu-baseline] [--no-extras]
    (--workload vga: 640x480 / 600 features, for tests that drive the launcher with many ranks on one GPU; not a BASELINE configuration)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# width, height, features, default stereo frames per step per GPU (c2 at 512 frames: ~14 GB of HBM for the two
# buffer sets).  c2: 256 / 512 / 1024 / 2048 / 4096 frames per step -> 120.9 / 125.9 / 126.3 / 126.9 / 125.6 k frames/s on one MI355X (round 4,
# profiles/r04/batch_sweep.txt): launch tails amortise up to 512 frames (1024 images per launch), nothing beyond.  c4: BASELINE configs[3] is "8 frames in flight".
# "vga" is not a BASELINE configuration: 640 x 480 / 600 features, the shape tests/ use to drive the --gpus N launcher and the exchange with many
# ranks on ONE time-sliced GPU (functional coverage only; a line produced from it says so in `metric`)
WORKLOADS = {"c2": (752, 480, 1200, 512), "c4": (1920, 1080, 4000, 8), "vga": (640, 480, 600, 8)}
W, H, NFEAT = 752, 480, 1200
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def level_sizes(w, h, nlevels=8, sf=1.2):
    out, inv = [], np.float32(1.0)
    scale = np.float32(1.0)
    for l in range(nlevels):
        if l:
            scale = np.float32(scale * np.float32(sf))
        inv = np.float32(1.0) / scale
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algorithmic_bytes(w, h):
    """SURVEY.md §8(d) per-image algorithmic bytes of each extractor stage."""
    lv = level_sizes(w, h)
    P = sum(a * b for a, b in lv)
    Pp = sum((a + 38) * (b + 38) for a, b in lv)
    last = lv[-1][0] * lv[-1][1]
    return {"pyramid": w * h + (P - last) + Pp, "fast": P, "blur": 2 * P}


def make_batch(global_ids, total, seed=0):
    """Stereo frames `global_ids` of a synthetic stream of `total` frames: seeded sequences (four base scenes, taken in turn) of
    min(total / 4, 64) consecutive frames with a 3x2 px/frame global shift, so frame g-1 is a real 'previous frame' of frame g except at
    the sequence starts.  Sequences are capped at 64 frames: shift_image replicates the edge, and beyond ~190 px of shift a 752-px image
    degenerates into streaks with no corners (a 1024-frame sequence would be mostly featureless frames — a batch sweep without the cap
    "found" 187 k frames/s at 4096 frames per step)."""
    from morb_slam_amd.synth import make_stereo_pair, shift_image
    base = [make_stereo_pair(W, H, seed=seed * 16 + i) for i in range(4)]
    imgs = np.empty((len(global_ids), 2, H, W), np.uint8)
    per = min(max(total // 4, 1), 64)
    for k, g in enumerate(global_ids):
        l, r = base[(g // per) % 4]
        dx, dy = 3 * (g % per), 2 * (g % per)
        imgs[k, 0] = shift_image(l, dx, dy)
        imgs[k, 1] = shift_image(r, dx, dy)
    return imgs


def cpu_baseline(target_s=12.0):
    """The CPU oracle ("port": an OpenCV-free restatement, not the OpenCV-backed binary) timed on this host's
    cores on a bounded sample of the SAME per-frame workload as the GPU step: ORBextractor x2 +
    ComputeStereoMatches + ComputeBoW + SearchByBoW against the previous frame.  `value` = frame-parallel on every
    hardware thread; beside it the reference's own shapes: one thread, and two threads with the left / right extraction
    side by side (Frame.cc:194-197)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from morb_slam_amd.synth import make_vocabulary
    cores = os.cpu_count() or 1
    frames = make_batch(range(4), 4, seed=99)
    vd, vf = make_vocabulary(10, 6, seed=0)
    mbf, mb = np.float32(458.654 * 0.11), np.float32(0.11)
    exts = [(O.OracleExtractor(NFEAT), O.OracleExtractor(NFEAT)) for _ in range(cores)]
    prev = [None] * cores
    rng = np.random.default_rng(7)
    pair = ThreadPoolExecutor(2)

    def work(i, two_threads=False):
        l, r = exts[i]
        f = frames[i % len(frames)]
        if two_threads:      # the reference's threadLeft / threadRight
            fl, fr = pair.submit(l, f[0]), pair.submit(r, f[1])
            (_, kl, dl), (_, kr, dr) = fl.result(), fr.result()
        else:
            _, kl, dl = l(f[0]); _, kr, dr = r(f[1])
        O.stereo_matches(l, r, kl, dl, kr, dr, mbf, mb)
        _, node = O.bow_transform(dl, vd, vf, 10, 6, 4)
        if prev[i] is not None:
            pk, pd, pn, has = prev[i]
            O.search_by_bow(pd, pk["angle"], has, pn, dl, kl["angle"], node, 0.7, True)
        prev[i] = (kl, dl, node, (rng.random(len(kl)) < 0.8))
        return 1

    def serial(two, budget):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            n += work(0, two)
        return n / (time.perf_counter() - t0)
    # C1 = BASELINE configs[0]: mono 752x480, 1000 features, CPU ORBextractor only (Frame.cc:428 passes the lapping area [0, 1000])
    c1e = O.OracleExtractor(1000)
    n1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 1.5:
        c1e(frames[n1 % len(frames)][0], (0, 1000)); n1 += 1
    c1 = n1 / (time.perf_counter() - t1)
    fps1 = serial(False, target_s / 6)
    fps2 = serial(True, target_s / 6)
    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < target_s * 2 / 3:
            done += sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "stereo frames/s", "cores": cores, "kind": "port",
            "c1_mono_1000feat_extract_only": {"value": c1, "unit": "mono frames/s", "cores": 1,
                                              "config": "BASELINE configs[0]: mono 752x480, 1000 features, CPU ORBextractor only"},
            "one_thread": {"value": fps1, "cores": 1}, "two_threads_left_right": {"value": fps2, "cores": 2},
            "sample": f"{done} stereo {W}x{H} frames, {NFEAT} features: oracle extract x2 + stereo match + BoW descent + "
                      f"SearchByBoW (frame-parallel on {cores} threads); the 1- and 2-thread figures on ~{target_s / 6:.0f} s each"}


def optimizer_extras(dev_index):
    """Secondary BASELINE metrics on rank 0: LocalBA LM iterations/s (config 5: 20 free + 6 fixed KFs, 3000 points)
    and PoseOptimization frames/s (config 3 stage), device-resident problems, CPU oracle timed beside them."""
    import torch
    from morb_slam_amd import BAProblem, Optimizer
    from morb_slam_amd.synth import make_ba_problem, make_pose_problem
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    opt = Optimizer(device=dev_index)
    b = make_ba_problem(seed=1)
    p = BAProblem(opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
    p.solve(); _, _, _, st = p.results()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        p.solve()
    _, _, _, st = p.results()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); its, *_ = O.local_ba(b); dc = time.perf_counter() - t0
    # the one-shot ABI call of the reference binding (host graph in -> host poses / points / erase flags out, problem built per call)
    from morb_slam_amd.optimizer import local_bundle_adjustment_oneshot
    oargs = (opt, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"])
    local_bundle_adjustment_oneshot(*oargs)
    t0 = time.perf_counter()
    for _ in range(5):
        local_bundle_adjustment_oneshot(*oargs)
    d1 = (time.perf_counter() - t0) / 5
    # eight replicas of the same problem solved side by side (eight handles = eight streams, eight host threads: morb_ba_solve blocks its caller
    # until the last LM decision): LocalBundleAdjustment does not shard ("replicas only"), this is what one GPU sustains
    from concurrent.futures import ThreadPoolExecutor
    NREP = 8
    ropts = [Optimizer(device=dev_index) for _ in range(NREP)]
    rprobs = [BAProblem(o, b["kfPose"], b["kfFixed"], b["mpPos"], b["eKF"], b["eMP"], b["eObs"], b["eInvSigma2"], b["cam"]) for o in ropts]
    def rsolve(q):
        for _ in range(5):
            q.solve()
        return q.results()[3]
    with ThreadPoolExecutor(NREP) as tp:
        list(tp.map(lambda q: q.solve(), rprobs))
        t0 = time.perf_counter()
        rst = list(tp.map(rsolve, rprobs))
        drep = time.perf_counter() - t0
    replicas = {"replicas": NREP, "solves_per_replica": 5, "lm_iters_per_s_aggregate": float(sum(int(x[0]) for x in rst) * 5 / drep),
                "ms_per_solve_per_replica": drep / 5 * 1e3}
    for q in rprobs:
        q.close()
    sms, sflops, suseful = p.schur_profile(50)     # the Schur product alone, HIP events on the handle's stream
    # the reference takes every covisible keyframe (Optimizer.cc:1058-1070): windows beyond the LDS-resident LDL^T (global-memory solver)
    large = {}
    for nf in (30, 45, 60):
        bl = make_ba_problem(seed=5, n_free=nf, n_fixed=6, n_points=3000)
        pl = BAProblem(opt, bl["kfPose"], bl["kfFixed"], bl["mpPos"], bl["eKF"], bl["eMP"], bl["eObs"], bl["eInvSigma2"], bl["cam"])
        pl.solve(); pl.results()
        t0 = time.perf_counter()
        for _ in range(5):
            pl.solve()
        stl = pl.results()[3]
        dl = (time.perf_counter() - t0) / 5
        large[f"{nf}_free_keyframes"] = {"ms_per_solve": dl * 1e3, "lm_iters_per_s": float(stl[0] / dl), "outer_lm_iters": int(stl[0]), "lm_trials": int(stl[1])}
    F = 256
    probs = [make_pose_problem(600, seed=s % 8) for s in range(F)]
    dev = torch.device("cuda", dev_index)
    t = [torch.from_numpy(np.stack([q[k] for q in probs])).to(dev) for k in ("hasMP", "obs", "invSigma2", "Xw")]
    pose0 = torch.from_numpy(np.stack([q["pose0"] for q in probs])).to(dev)
    out = None
    opt.set_exact_order(False)   # the tree-sum mode first (the trial count may differ from g2o's by one), then the default
    for _ in range(2):
        out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"], out=out)
    torch.cuda.synchronize(dev)
    poses = [pose0.clone() for _ in range(10)]   # in/out argument: one copy per call, made outside the timed loop
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(10):
        out = opt.PoseOptimization(t[0], t[1], t[2], t[3], poses[k], probs[0]["cam"], out=out)
    torch.cuda.synchronize(dev)
    dtp = (time.perf_counter() - t0) / 10
    # the default mode (morb_optimizer_set_exact_order(1): sums in edge order on the FP64 matrix core, g2o's LM path decision for decision)
    opt.set_exact_order(True)
    info_timed = opt.info()
    out = opt.PoseOptimization(t[0], t[1], t[2], t[3], pose0.clone(), probs[0]["cam"], out=out)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(10):
        poses[k].copy_(pose0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(10):
        out = opt.PoseOptimization(t[0], t[1], t[2], t[3], poses[k], probs[0]["cam"], out=out)
    torch.cuda.synchronize(dev)
    dtpe = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for q in probs[:8]:
        O.pose_optimization(q)
    dcp = (time.perf_counter() - t0) / 8
    # visual-inertial tracking (SURVEY 8(f) N1 slice): keyframe -> frame A (LastKeyFrame) -> frame B (LastFrame), IMU at 200 Hz
    from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_sequence
    FI = 256
    seq = [make_inertial_sequence(600, seed=s % 8, n_imu=20) for s in range(FI)]
    nga, walk = imu_calib_diagonals()
    stk = lambda which, k: torch.from_numpy(np.stack([q[which][k] for q in seq])).to(dev)
    cat = lambda which, k: torch.from_numpy(np.concatenate([q[which][k] for q in seq])).to(dev)
    def starts(which, k):
        return torch.from_numpy(np.cumsum([0] + [len(q[which][k]) for q in seq]).astype(np.int32)).to(dev)
    A = [stk(0, k) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "kfState", "state0", "bias")]
    B = [stk(1, k) for k in ("hasMP", "obs", "invSigma2", "Xw", "close", "state0")]
    imuA = (starts(0, "dt"), cat(0, "acc"), cat(0, "gyro"), cat(0, "dt"))
    imuF = (starts(1, "dtF"), cat(1, "accF"), cat(1, "gyroF"), cat(1, "dtF"))
    imuK = (starts(1, "dt"), cat(1, "acc"), cat(1, "gyro"), cat(1, "dt"))
    camI, Tbc = seq[0][0]["cam"], seq[0][0]["Tbc12"]
    preAll = outA = outB = None
    stA, stB = A[6].clone(), B[5].clone()
    # the three preintegrations of a frame pair (keyframe -> A, A -> B, keyframe -> B) are one launch over 3 x FI measurement sequences, and the
    # whole step runs on one stream without a host round trip (Tracking::PreintegrateIMU fills both of a frame's preintegrations in one call too)
    nA, nF = int(imuA[1].shape[0]), int(imuF[1].shape[0])
    startAll = torch.cat([imuA[0], imuF[0][1:] + nA, imuK[0][1:] + nA + nF]).contiguous()
    accAll, gyroAll, dtAll = (torch.cat([imuA[k], imuF[k], imuK[k]]).contiguous() for k in (1, 2, 3))
    biasAll = A[7].repeat(3, 1).contiguous()
    s_in = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    def inertial_step():
        nonlocal preAll, outA, outB
        st = s_in.cuda_stream
        preAll = opt.PreintegrateIMU(startAll, accAll, gyroAll, dtAll, biasAll, nga, walk, out=preAll, stream=st)
        with torch.cuda.stream(s_in):
            stA.copy_(A[6]); stB.copy_(B[5])
        outA = opt.PoseInertialOptimizationLastKeyFrame(A[0], A[1], A[2], A[3], A[4], camI, Tbc, A[5], preAll[:FI], stA, out=outA, stream=st)
        outB = opt.PoseInertialOptimizationLastFrame(B[0], B[1], B[2], B[3], B[4], camI, Tbc, stA, preAll[FI:2 * FI], preAll[2 * FI:], outA[2], stB,
                                                     out=outB, stream=st)
    for _ in range(2):
        inertial_step()
    torch.cuda.synchronize(dev)
    chk = opt.PreintegrateIMU(*imuF, A[7], nga, walk)   # (outside the timed region: the middle third of the one launch = its own launch)
    torch.cuda.synchronize(dev)
    if not torch.equal(chk, preAll[FI:2 * FI]):
        raise RuntimeError("batched preintegration differs from the per-set launch")
    t0 = time.perf_counter()
    for _ in range(5):
        inertial_step()
    torch.cuda.synchronize(dev)
    dti = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for pA, pB in seq[:4]:
        oa = O.imu_preintegrate(pA["bias"], nga, walk, pA["acc"], pA["gyro"], pA["dt"])
        rA = O.pose_inertial_optimization_last_keyframe(pA, oa)
        of = O.imu_preintegrate(pB["bias"], nga, walk, pB["accF"], pB["gyroF"], pB["dtF"])
        ok = O.imu_preintegrate(pB["bias"], nga, walk, pB["acc"], pB["gyro"], pB["dt"])
        O.pose_inertial_optimization_last_frame(pB, rA[1], of, ok, rA[3])
    dci = (time.perf_counter() - t0) / 4
    inertial = {"frame_pairs": FI, "edges_per_frame": 600, "imu_samples_per_interval": 20,
                "stages": ["PreintegrateIMU (3 sequences per frame pair, one launch)", "PoseInertialOptimizationLastKeyFrame", "PoseInertialOptimizationLastFrame"],
                "ms_per_batch": dti * 1e3, "frame_pairs_per_s": FI / dti, "cpu_oracle_frame_pairs_per_s_1core": 1.0 / dci}
    # LocalInertialBA: 10-keyframe window + 6 fixed keyframes, 3000 points (one-shot call incl. graph upload)
    from morb_slam_amd.synth import make_inertial_ba_problem
    pi = make_inertial_ba_problem(n_opt=10, seed=1, n_points=3000)
    nlink = len(pi["imuStart"]) - 1
    prei = opt.PreintegrateIMU(torch.from_numpy(np.asarray(pi["imuStart"], np.int32)).to(dev), torch.from_numpy(pi["acc"]).to(dev),
                               torch.from_numpy(pi["gyro"]).to(dev), torch.from_numpy(pi["dt"]).to(dev),
                               torch.from_numpy(np.tile(pi["bias"], (nlink, 1)).astype(np.float32)).to(dev), nga, walk)
    torch.cuda.synchronize(dev)
    prei = prei.cpu().numpy()      # k_imu_preintegrate's records (bit-identical to the oracle's, tests/test_inertial_gpu.py)
    iargs = (pi["kfState"], pi["kfKind"], pi["mpPos"], pi["mpClose"], pi["eKF"], pi["eMP"], pi["eObs"], pi["eInvSigma2"], pi["iKF1"],
             pi["iKF2"], prei, pi["iRobust"], pi["iInfoScale"], pi["cam"], pi["Tbc12"])
    opt.LocalInertialBA(*iargs)
    tl = []
    for _ in range(7):
        t0 = time.perf_counter(); _, _, _, sti = opt.LocalInertialBA(*iargs); tl.append(time.perf_counter() - t0)
    dtl = sorted(tl)[len(tl) // 2]   # median of the one-shot calls (each includes the graph upload)
    t0 = time.perf_counter(); ro = O.local_inertial_ba(pi, prei); dcl = time.perf_counter() - t0
    inertial_ba = {"keyframes_opt_fixed": [10, 7], "points": 3000, "edges": int(len(pi["eKF"])), "inertial_links": 10,
                   "outer_lm_iters": int(sti[0]), "lm_trials": int(sti[1]), "ms_per_solve": dtl * 1e3,
                   "lm_iters_per_s": float(sti[0] / dtl), "cpu_oracle_lm_iters_per_s": float(ro[4][0] / dcl)}
    return {"pose_inertial_tracking": inertial, "local_inertial_ba": inertial_ba,
            "local_ba": {"edges": int(len(b["eKF"])), "keyframes_free_fixed": [20, 6], "points": 3000,
                         "outer_lm_iters": int(st[0]), "lm_trials": int(st[1]), "ms_per_solve": dt * 1e3, "one_shot_call_ms": d1 * 1e3,
                         "lm_iters_per_s": float(st[0] / dt), "cpu_oracle_lm_iters_per_s": float(its / dc), "cpu_oracle_ms_per_solve_1core": dc * 1e3,
                         # the MFMA kernel of the solve (north_star: Schur reduction on the matrix cores); peak = dense FP64 MFMA,
                         # 256 CUs x 128 flop/clk x 2.4 GHz (tools/alu_issue.hip: v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD)
                         "roofline": {"bound": "mfma", "kernel": "k_schur_mfma", "achieved": sflops / (sms * 1e-3) / 1e12, "peak": 78.6,
                                      "unit": "TFLOP/s", "frac": sflops / (sms * 1e-3) / 1e12 / 78.6, "avg_launch_ms": sms,
                                      "flops_per_launch": sflops, "sparse_form_flops": suseful,
                                      # the same launch priced by the flops of g2o's sparse block-pair form (what the reference computes)
                                      "useful_achieved": suseful / (sms * 1e-3) / 1e12, "useful_frac": suseful / (sms * 1e-3) / 1e12 / 78.6,
                                      "note": "frac counts the dense product over all landmarks (zero blocks included), useful_frac only the "
                                              "block pairs g2o forms; ~11 us launch, latency-bound"},
                         "large_windows": large, "concurrent_replicas": replicas},
            "pose_optimization": {"frames": F, "edges_per_frame": 600, "frames_per_s": F / dtpe, "frames_per_s_exact_order_mode": F / dtpe, "frames_per_s_tree_sum_mode": F / dtp,
                                  "mode": "edge-order sums (the default): g2o's LM path decision for decision",
                                  # morb_optimizer_info of the handle that was timed: mfma_chain 1 = the edge-order sums ran on v_mfma_f64_4x4x4 (0 = on dependent
                                  # v_add_f64, same bits, ~1.5 x slower); mfma_selftest 1 = this device passed the create-time order self-test, 0 = REJECTED it
                                  **info_timed,
                                  "cpu_oracle_frames_per_s_1core": 1.0 / dcp}}


def tracking_extras(dev_index):
    """The searches and optimisations Tracking / LocalMapping run once a map exists (north_star's SearchByProjection / SearchForTriangulation
    kernels), device-resident, on rank 0: TrackWithMotionModel + TrackLocalMap as one chain (morb_slam_amd/tracking.py: SearchByProjection(Cur,
    Last) -> PoseOptimization -> isInFrustum over 2048 local map points -> SearchByProjection(F, MapPoints) -> PoseOptimization) for 256 frames
    per step and one frame at a time, SearchForTriangulation + Fuse over 20 keyframe pairs, the CPU oracle timed beside each, and frames of the
    LAST timed step checked against the oracle stage by stage (tests/tracking_check.py: the checker, never the product)."""
    import torch
    from morb_slam_amd.tracking import TrackingChain, build_chains
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import tracking_check
    B, G = 256, 256
    imgs = make_batch(range(G), G, seed=0)
    ch, ks, host = build_chains(imgs, B=B, npairs=20, seq_len=64, device=dev_index)
    sc = host["scene"]

    def timed(fn, sync, n, warm=3):
        for _ in range(warm):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t0) / n
    dt = timed(ch.step, ch.sync, 10)
    ch.step(snapshot=True); ch.sync()
    nver = tracking_check.verify_tracking(ch, [0, 37, 101, 255], host["kps"], host["cnt"], host["desc"], host["curUR"], sc)
    out = {"frames_per_step": B, "local_map_points": int(ch.mpCap), "mode": "PoseOptimization in edge order (the LM path of g2o, as the C++ drop-in runs it)",
           "stages": ["SearchByProjection(Cur, Last, th 7)", "PoseOptimization", "discard outliers", "isInFrustum", "SearchByProjection(F, MapPoints, th 1)",
                      "PoseOptimization", "inlier count"],
           "ms_per_step": dt * 1e3, "frames_per_s": B / dt, "verified_frames": nver,
           "mean_matches_last_frame": float(ch.nmLast.float().mean()), "mean_matches_local_map": float(ch.nmLocal.float().mean()),
           "mean_inliers": float(ch.nInl.float().mean()), **{k: v for k, v in ch.opt.info().items()}}
    one = {k: v[:1] for k, v in sc.items()}
    c1 = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[:1].contiguous(), one, device=dev_index)
    out["latency_b1_ms"] = timed(c1.step, c1.sync, 50) * 1e3
    c1.close()
    cht = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight, sc, device=dev_index, exact_order=False)
    dtt = timed(cht.step, cht.sync, 10)
    cht.close()
    c1t = TrackingChain(ch.P, ch.cam, ch.kps, ch.desc, ch.count, ch.uRight[:1].contiguous(), one, device=dev_index, exact_order=False)
    out["tree_sum_mode"] = {"frames_per_s": B / dtt, "ms_per_step": dtt * 1e3, "latency_b1_ms": timed(c1t.step, c1t.sync, 50) * 1e3,
                            "note": "morb_optimizer_set_exact_order(0), tree sums: same poses to ~1e-9 and same outlier flags, LM trial counts within +-2 of g2o's"}
    c1t.close()
    # the CPU oracle on the same frames, one thread (the reference runs Tracking on one thread): the chain stage by stage
    P = ch.P
    invS2 = (np.float32(1.0) / np.array(list(P.levelSigma2)[:P.nlevels], np.float32)).astype(np.float32)

    def oracle_frame(f):
        ci, li = int(sc["curImg"][f]), int(sc["lastImg"][f])
        nc, nl = int(host["cnt"][ci]), int(host["cnt"][li])
        F = O.make_frame(P, host["kps"][ci, :nc], host["desc"][ci, :nc], host["curUR"][f, :nc])
        nMP = int(sc["nMP"][f]); lastMP = sc["lastMP"][f, :nl]; lv = (lastMP >= 0).astype(np.uint8); lm = np.maximum(lastMP, 0)
        mpXw, mpDesc, mpHasObs = sc["mpXw"][f], sc["mpDesc"][f], sc["mpHasObs"][f]
        _, me = O.search_by_projection_last(F, np.zeros(nc, np.uint8), sc["pose0"][f], host["kps"][li, :nl], lv, mpXw[lm], mpDesc[lm],
                                            mpHasObs[lm] * lv, 7.0, 0, 0, True)
        fm = np.where(me >= 0, lastMP[np.maximum(me, 0)], -1).astype(np.int32)
        he, oe, se, Xe = O.pose_edges(F, invS2, fm, mpXw)
        _, pe, ole, _ = O.pose_optimization(dict(hasMP=he, obs=oe, invSigma2=se, Xw=Xe, pose0=sc["pose0"][f], cam=ch.cam))
        _, _, fm2, blk, seen = O.discard_outliers(fm, ole, mpHasObs)
        Re, te, Oe = O.frame_set_pose(pe)
        trk = O.is_in_frustum(F, Re, te, Oe, mpXw[:nMP], sc["mpNormal"][f, :nMP], sc["mpMaxD"][f, :nMP], sc["mpMinD"][f, :nMP], 0.5)
        _, me2 = O.search_by_projection_mps(F, blk, trk, seen[:nMP], mpDesc[:nMP], mpHasObs[:nMP], 1.0, False, 0.0, 0.8, match_init=fm2)
        he, oe, se, Xe = O.pose_edges(F, invS2, me2, mpXw)
        O.pose_optimization(dict(hasMP=he, obs=oe, invSigma2=se, Xw=Xe, pose0=pe, cam=ch.cam))
    t0 = time.perf_counter(); nf = 0
    while time.perf_counter() - t0 < 3.0:
        oracle_frame(nf % B); nf += 1
    out["cpu_oracle_frames_per_s_1core"] = nf / (time.perf_counter() - t0)
    # LocalMapping's searches per new keyframe
    dk = timed(ks.step, ks.sync, 10)
    nk = tracking_check.verify_keyframe_searches(ks, [0, 7, 19], host["kps"], host["cnt"], host["desc"], host["node"], host["uR_img"], host["kscene"])
    kout = {"keyframe_pairs": 20, "stages": ["SearchForTriangulation", "Fuse (the search)"], "ms_per_step": dk * 1e3, "pairs_per_s": 20 / dk,
            "verified_pairs": nk, "mean_triangulation_matches": float(ks.tri[1].float().mean()),
            "mean_fused_candidates": float((ks.fused[0] >= 0).sum(1).float().mean())}
    ksc = host["kscene"]
    sig2 = list(P.levelSigma2)[:P.nlevels]; sf = list(P.scaleFactors)[:P.nlevels]
    t0 = time.perf_counter(); npr = 0
    while time.perf_counter() - t0 < 2.0:
        p = npr % 20
        a, b = int(ksc["img1"][p]), int(ksc["img2"][p]); na, nb = int(host["cnt"][a]), int(host["cnt"][b])
        O.search_for_triangulation(host["kps"][a, :na], host["desc"][a, :na], host["node"][a, :na], ksc["hasMP"][a, :na], host["uR_img"][a, :na],
                                   host["kps"][b, :nb], host["desc"][b, :nb], host["node"][b, :nb], ksc["hasMP"][b, :nb], host["uR_img"][b, :nb],
                                   sig2, sf, [P.fx, P.fy, P.cx, P.cy], ksc["R12"][p].reshape(3, 3), ksc["t12"][p], ksc["ep"][p], False, False, False)
        KF = O.make_frame(P, host["kps"][b, :nb], host["desc"][b, :nb], ksc["fuseUR"][p, :nb])
        n = int(ksc["nMP"][p])
        O.fuse_search(KF, invS2, ksc["Tcw"][p], ksc["Ow"][p], ksc["valid"][p, :n], ksc["Pw"][p, :n], ksc["normal"][p, :n], ksc["maxD"][p, :n],
                      ksc["minD"][p, :n], ksc["mpDesc"][p, :n], 3.0, False)
        npr += 1
    kout["cpu_oracle_pairs_per_s_1core"] = npr / (time.perf_counter() - t0)
    ch.close(); ks.close()
    return {"tracking_chain": out, "keyframe_searches": kout}


def config_extras(dev_index):
    """Throughput of the other BASELINE configs on rank 0 (short runs; parity for these shapes is in tests/):
    C3 = TUM-VI-shaped fisheye stereo 512x512, 1500 features, lapping areas: extract x2 + ComputeStereoFishEyeMatches
    (all-pairs Hamming knn2 + KB8 triangulation) + PoseOptimization on the rig; C4 = 1920x1080 stereo, 4000 features."""
    import torch
    from morb_slam_amd import ORBextractor, ORBmatcher, Optimizer
    from morb_slam_amd.synth import (TUMVI_CAM_L, TUMVI_CAM_R, TUMVI_T_C1_C2, make_pose_problem_fisheye, make_stereo_pair)
    dev = torch.device("cuda", dev_index)
    out = {}
    st = torch.cuda.Stream(device=dev)

    def timed(fn, n):
        fn(); st.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        st.synchronize()
        return (time.perf_counter() - t0) / n
    # ---- C3
    B3 = 256   # stereo frames per step (PoseOptimization is one workgroup per frame, ~1.2 ms whatever the batch: 32 / 128 / 256 / 512 frames per step -> 15 / 50 / 67 / 68 k frames/s)
    base = [make_stereo_pair(512, 512, seed=100 + i) for i in range(4)]
    imgs = torch.from_numpy(np.stack([base[i % 4][k] for i in range(B3) for k in (0, 1)])).to(dev)
    ext = ORBextractor(1500, 1.2, 8, 20, 7, device=dev_index)
    mt = ORBmatcher(0.7, True, device=dev_index)
    lap = np.tile(np.array([[0, 511], [0, 511]], np.int32), (B3, 1))     # TUM-VI.yaml: the lapping areas span the images
    sigma2 = ext.GetScaleSigmaSquares()
    Rlr = TUMVI_T_C1_C2[:3, :3].astype(np.float32); tlr = TUMVI_T_C1_C2[:3, 3].astype(np.float32)
    probs = [make_pose_problem_fisheye(seed=s % 4) for s in range(B3)]
    capP = max(len(q["hasMP"]) for q in probs)
    t = {k: torch.from_numpy(np.stack([np.pad(q[k], [(0, capP - len(q[k]))] + [(0, 0)] * (q[k].ndim - 1)) for q in probs])).to(dev)
         for k in ("hasMP", "obs", "invSigma2", "Xw")}
    pose0 = torch.from_numpy(np.stack([q["pose0"] for q in probs])).to(dev)
    nl = torch.tensor([q["Nleft"] for q in probs], dtype=torch.int32, device=dev)
    cn = torch.tensor([len(q["hasMP"]) for q in probs], dtype=torch.int32, device=dev)
    opt = Optimizer(device=dev_index)
    eo = None

    def c3():
        nonlocal eo
        eo = ext.extract_batch(imgs, lap=lap, out=eo, stream=st.cuda_stream)
        mt.ComputeStereoFishEyeMatches(eo[0], eo[1], eo[2], eo[3], TUMVI_CAM_L, TUMVI_CAM_R, Rlr, tlr, sigma2, stream=st.cuda_stream)
        opt.PoseOptimizationFisheye(t["hasMP"], t["obs"], t["invSigma2"], t["Xw"], pose0.clone(), nl, cn, TUMVI_CAM_L, TUMVI_CAM_R,
                                    probs[0]["Trl"], stream=st.cuda_stream)
    with torch.cuda.stream(st):
        dt3 = timed(c3, 5)
    out["c3_fisheye_512x512_1500feat"] = {"stereo_frames_per_step": B3, "frames_per_s": B3 / dt3, "ms_per_step": dt3 * 1e3,
                                          "stages": ["extract_left+right(lapping areas)", "ComputeStereoFishEyeMatches",
                                                     "PoseOptimization(700 edges, KB8 rig)"],
                                          "mean_keypoints_per_image": float(eo[2].float().mean().item())}
    ext.close()
    # ---- C4
    B4 = 8
    base = [make_stereo_pair(1920, 1080, seed=200 + i) for i in range(2)]
    imgs4 = torch.from_numpy(np.stack([base[i % 2][k] for i in range(B4) for k in (0, 1)])).to(dev)
    ext4 = ORBextractor(4000, 1.2, 8, 20, 7, device=dev_index)
    mbf, mb = 458.654 * 0.11, 0.11
    e4 = s4 = None

    def c4():
        nonlocal e4, s4
        e4 = ext4.extract_batch(imgs4, out=e4, stream=st.cuda_stream)
        s4 = mt.ComputeStereoMatches(ext4, e4[0], e4[1], e4[2], mbf, mb, out=s4, stream=st.cuda_stream)
    dt4 = timed(c4, 5)
    out["c4_1920x1080_4000feat"] = {"stereo_frames_per_step": B4, "frames_per_s": B4 / dt4, "ms_per_step": dt4 * 1e3,
                                    "stages": ["extract_left+right", "ComputeStereoMatches"],
                                    "mean_keypoints_per_image": float(e4[2].float().mean().item())}
    # single 1080p frame latency (host image in -> host keypoints / descriptors / uRight / depth out): C4 as BASELINE states it is ONE frame per GPU per step
    lst = torch.cuda.Stream(device=dev)
    pin = torch.from_numpy(np.stack(base[0])).pin_memory()
    d1 = torch.empty((2, 1080, 1920), dtype=torch.uint8, device=dev)
    cap4 = ext4.max_keypoints
    hk = torch.empty((2, cap4, 28), dtype=torch.uint8).pin_memory(); hd = torch.empty((2, cap4, 32), dtype=torch.uint8).pin_memory()
    hu = torch.empty((1, cap4), dtype=torch.float32).pin_memory()
    o1 = so1 = None

    def one4():
        nonlocal o1, so1
        with torch.cuda.stream(lst):
            d1.copy_(pin, non_blocking=True)
            o1 = ext4.extract_batch(d1, out=o1, stream=lst.cuda_stream)
            so1 = mt.ComputeStereoMatches(ext4, o1[0], o1[1], o1[2], mbf, mb, out=so1, stream=lst.cuda_stream)
            hk.copy_(o1[0], non_blocking=True); hd.copy_(o1[1], non_blocking=True); hu.copy_(so1[0], non_blocking=True)
        lst.synchronize()
    for _ in range(3):
        one4()
    t0 = time.perf_counter()
    for _ in range(20):
        one4()
    out["c4_1920x1080_4000feat"]["latency_b1_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    ext4.close()
    # ---- BASELINE's literal metric string says "1000 feat": the C2 chain at nfeatures = 1000 beside the 1200-feature headline
    B2 = 256
    base2 = [make_stereo_pair(752, 480, seed=i) for i in range(4)]
    imgs2 = torch.from_numpy(np.stack([base2[i % 4][k] for i in range(B2) for k in (0, 1)])).to(dev)
    ext2 = ORBextractor(1000, 1.2, 8, 20, 7, device=dev_index)
    e2 = s2 = None

    def c2k():
        nonlocal e2, s2
        e2 = ext2.extract_batch(imgs2, out=e2, stream=st.cuda_stream)
        s2 = mt.ComputeStereoMatches(ext2, e2[0], e2[1], e2[2], mbf, mb, out=s2, stream=st.cuda_stream)
    dt2 = timed(c2k, 10)
    out["c2_752x480_1000feat"] = {"stereo_frames_per_step": B2, "frames_per_s": B2 / dt2, "ms_per_step": dt2 * 1e3,
                                  "stages": ["extract_left+right", "ComputeStereoMatches"], "schedule": "un-pipelined, one stream",
                                  "mean_keypoints_per_image": float(e2[2].float().mean().item())}
    ext2.close()
    return out


def launch_ranks(n):
    """`bench.py --gpus N` run directly (no torchrun): start N rank processes of this same script — fresh interpreters with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per GPU — relay rank 0's stdout (the ONE JSON line),
    let the other ranks' output through to stderr, and return non-zero if any rank fails (the survivors are terminated)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []

    import ctypes
    import signal
    libc = ctypes.CDLL("libc.so.6")     # (loaded in the launcher, before any fork)

    def die_with_launcher():        # Linux PR_SET_PDEATHSIG: a rank never outlives a launcher that was killed (it would sit on the GPU in a collective)
        libc.prctl(1, signal.SIGKILL, 0, 0, 0)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True, preexec_fn=die_with_launcher))
    rc, out0 = 0, b""
    pending = set(range(n))
    try:
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0:
                    try:
                        o, _ = p.communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                else:
                    try:
                        p.wait(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        continue
                pending.discard(r)
                if p.returncode != 0:
                    rc = rc or (p.returncode if p.returncode > 0 else 1)
                    print(f"bench.py: rank {r} exited with {p.returncode}", file=sys.stderr)
            if rc and pending:          # a rank died: the others would wait in the rendezvous / collective for ever
                time.sleep(2.0)
                for r in pending:
                    if procs[r].poll() is None:
                        procs[r].terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    # rank 0's stdout: the JSON line goes to stdout, anything else a library printed there (gloo's "[Gloo] Rank 0 is connected ...") to stderr
    for ln in out0.decode(errors="replace").splitlines():
        print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr)
    sys.stdout.flush()
    return rc


def main():
    global W, H, NFEAT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = BASELINE configs[1] (752x480 / 1200 feat, the configuration the metric is quoted on); c4 = configs[3] (1920x1080 / 4000 feat); "
                         "vga = 640x480 / 600 feat, functional tests of the launcher only")
    ap.add_argument("--batch", type=int, default=0,
                    help="stereo frames per step per GPU (default 512 for c2, see WORKLOADS; 8 for c4)")
    ap.add_argument("--sets", type=int, default=2, help="buffer sets = steps in flight (>= 2)")
    ap.add_argument("--extract-streams", type=int, default=1, help="1: one extraction stream for all buffer sets (default); 2: one per set")
    ap.add_argument("--stagger", action="store_true", help="with --extract-streams 2: step i + 1's extraction starts when step i's FAST stage is done")
    ap.add_argument("--matchers", choices=["beside-pyramid", "under-quadtree", "under-fast"], default="beside-pyramid",
                    help="where a step's matchers run: right after its extraction, i.e. beside the NEXT step's pyramid (default), or held back until the "
                         "next step's FAST stage is done (morb_extractor_event_after_fast), i.e. underneath its quadtree, or until its pyramid is done "
                         "(morb_extractor_event_after_pyramid), i.e. underneath its FAST stage")
    ap.add_argument("--exchange", choices=["ring", "allgather"], default="ring",
                    help="N > 1: how a frame's predecessor features reach its rank: one send / recv to the next rank (ring, 1/N of the bytes) "
                         "or an all-gather of every rank's slabs (what north_star names)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="join all streams at the end of every step instead of running step i's matchers underneath step "
                         "i + 1's extraction (one buffer set instead of two)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-region oracle self-check of the last timed step")
    ap.add_argument("--verify-frames", type=int, default=8, help="stereo frames of the last timed step compared with the oracle")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="seconds of the sustained run (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only (used for the committed profiles)")
    ap.add_argument("--launch-probe", action="store_true",
                    help="ranks only rendezvous (gloo, CPU), count each other and exit: checks the --gpus N launcher without a GPU")
    args = ap.parse_args()
    W, H, NFEAT, defB = WORKLOADS[args.workload]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: THIS process becomes the launcher.  It has not touched the GPU
        # (nothing above imports torch.cuda or the HIP library), starts N fresh rank processes and never exec()s.
        sys.exit(launch_ranks(args.gpus))

    # what RCCL needs on this pool (dmabuf IPC), for ranks started by ANY launcher — torchrun included — and before torch initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch
    pg_timeout = datetime.timedelta(seconds=float(os.environ.get("MORB_DIST_TIMEOUT_S", "180")))   # a rank that waits longer for a peer fails (non-zero exit)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch_probe:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=pg_timeout)
        t = torch.tensor([1.0, float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_probe": True, "n_gpus": world, "ranks_counted": int(t[0]), "rank_sum": int(t[1]),
                              "gpus_flag": args.gpus}))
        dist.barrier()
        dist.destroy_process_group()
        return
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MORB_DIST_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then
        # share devices); the driver's multi-GPU runs use the default: nccl = RCCL over xGMI, one rank per GPU
        backend = os.environ.get("MORB_DIST_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from morb_slam_amd import ORBextractor, parallel
    from morb_slam_amd.frontend import StereoFrontEnd
    B = args.batch or defB
    # global frame g lives on rank g % world, slot g // world (morb_slam_amd/parallel.py)
    gids = [parallel.global_frame(rank, world, s) for s in range(B)]
    host_frames = torch.from_numpy(make_batch(gids, B * world, seed=0))
    frames = host_frames.to(dev)                                                # [B, 2, H, W] resident in HBM
    images = frames.view(2 * B, H, W)
    # The step itself is morb_slam_amd/frontend.py (StereoFrontEnd): two buffer sets, step i's matchers on their own stream beside
    # step i + 1's extraction, k=10 / L=6 / levelsup=4 vocabulary (the ORBvoc shape; synthetic: ORBvoc.txt is a missing blob, SURVEY
    # finding 3), 80 % of the keyframe features holding a MapPoint.  tests/test_bench_chain_gpu.py builds the SAME object and compares
    # sampled frames with the oracle.  N GPUs: the previous frame lives on the previous rank -> its left-image features travel one rank
    # up the ring (--exchange ring, RCCL send / recv) or are all-gathered (--exchange allgather, north_star's wording) once per step.
    NSET = 1 if args.no_pipeline else max(2, args.sets)
    exch = args.exchange if world > 1 else None
    fe = StereoFrontEnd(images, NFEAT, B, device=local_rank, rank=rank, world=world, nset=NSET, extract_streams=args.extract_streams,
                        matchers=args.matchers, stagger=args.stagger, vocab=(10, 6, 4), exchange=exch)
    exts, sets, estreams, matcher, cap, mbf, mb = fe.exts, fe.sets, fe.estreams, fe.matcher, fe.cap, fe.mbf, fe.mb
    ext = exts[0]
    step, sync_streams = fe.step, fe.sync

    def sync_all():
        sync_streams()
        if dist is not None:
            dist.barrier()

    if fe.exch is not None:
        fe.exch.prime(dev)      # communicator + one dummy exchange BEFORE any step: RCCL builds it lazily, and --warmup 0 is a legal request
    for _ in range(args.warmup):
        step()
    for e in exts:
        e.set_profiling(True)   # stage-boundary HIP events on the launch stream; no host sync inside the timed region
    sync_all()
    if fe.exch is not None:
        fe.drain_exchange_ms(); fe.exchange_ms = []       # from here on: HIP events around every step's pack -> transfer -> unpack
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_streams()
    dt = time.perf_counter() - t0
    multi = None
    if dist is not None:
        own = dt
        xms = fe.drain_exchange_ms() or [0.0]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank figures (never `value`): the spread of the ranks' own step times and of their exchange times, so that a multi-GPU run shows
        # whether xGMI time or compute sets the step
        v = torch.tensor([own / args.steps * 1e3, -own / args.steps * 1e3, float(np.mean(xms)), -float(np.mean(xms)), float(np.max(xms))],
                         dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        multi = {"rank_ms_per_step": {"min": -float(v[1]), "max": float(v[0])},
                 "exchange_ms_per_step": {"min_over_ranks_of_mean": -float(v[3]), "max_over_ranks_of_mean": float(v[2]), "max_single": float(v[4]),
                                          "what": "HIP events on the exchange stream around slab pack -> transfer -> unpack (runs beside the next step's extraction)"},
                 "exchange": args.exchange, "backend": dist.get_backend()}
        dist.barrier()
    # per-stage ms per extract call (= per step), averaged over the handles of the buffer sets
    per_set = [e.stage_ms() for e in exts]
    stages = {k: sum(ps[k] for ps in per_set) / len(per_set) for k in per_set[0]}
    # ---- post-region self-check (outside the timed region, like cpu_baseline): sampled frames of the LAST timed step against the CPU
    # oracle, every field of the chain (tests/chain_check.py) — the configuration in the headline is one the oracle has seen
    verified = None
    if world == 1 and rank == 0 and not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import chain_check
        sample = sorted(set(int(x) for x in np.linspace(0, B - 1, min(B, args.verify_frames))))
        tv = time.perf_counter()
        nver = chain_check.verify_frames(fe, fe.last, sample, host_frames.view(2 * B, H, W).numpy())   # raises on the first difference
        verified = {"verified_frames": nver, "of_step": "last timed step", "frames": sample, "seconds": time.perf_counter() - tv,
                    "fields": ["keypoints", "descriptors", "mvuRight", "mvDepth", "bow_word", "bow_node", "SearchByBoW table", "nmatches"]}
    # ---- sustained figure (never `value`): the same step for >= 2 s and >= 1000 steps in windows of <= 100 steps; the headline's K steps fit inside one
    # boost-clock burst, this does not
    sustained = None
    if world == 1 and rank == 0 and args.sustained_s > 0:
        # windows of ~0.2 s (at least 10 steps, at most 100), at least 10 of them, at least --sustained-s seconds and at least 1000 steps in all
        win = int(min(100, max(10, round(0.2 / max(dt / args.steps, 1e-6)))))
        wfps, wst, tot = [], [], 0.0
        while tot < args.sustained_s or len(wfps) < 10 or win * len(wfps) < 1000:   # >= 2 s AND >= 1000 steps
            tw = time.perf_counter()
            for _ in range(win):
                step()
            sync_streams()
            d = time.perf_counter() - tw
            tot += d
            wfps.append(B * win / d)
            ps = [e.stage_ms() for e in exts]      # (each handle's last <= 64 calls of the window)
            wst.append({k: sum(q[k] for q in ps) / len(ps) for k in ps[0]})
            if len(wfps) >= 200:
                break
        srt = sorted(wfps)
        sustained = {"value": B * win * len(wfps) / tot, "unit": "frames/s", "seconds": tot, "steps": win * len(wfps), "window_steps": win,
                     "window_min": srt[0], "window_median": srt[len(srt) // 2], "window_max": srt[-1],
                     "extract_stage_ms_per_step": {k: {"min": min(w[k] for w in wst), "median": sorted(w[k] for w in wst)[len(wst) // 2],
                                                       "max": max(w[k] for w in wst)} for k in wst[0]}}
    # reference point outside the timed region: the same extraction with nothing else on the chip (the pipelined steps
    # above share the CUs between two extractions and the matchers, which stretches every kernel's launch duration)
    iso_n = 5
    for _ in range(iso_n):
        exts[0].extract_batch(images, out=sets[0].out, stream=estreams[0].cuda_stream)
        estreams[0].synchronize()
    iso = exts[0].stage_ms()
    for e in exts:
        e.set_profiling(False)
    cnt = sets[0].out[2].cpu().numpy()
    n_stereo = float((sets[0].st_out[0] >= 0).sum().item()) / B
    n_bow = float(sets[0].match_out[1].float().mean().item())

    h2d = lat = alt = None
    if world == 1 and not args.no_extras and NSET >= 2 and len(set(id(x) for x in estreams)) == 1:
        # ---- the other schedule, for the record (never `value`): one extraction stream per buffer set, so two extractions overlap each
        # other too.  A few per cent more frames/s, but every extraction kernel then shares the chip with another one and its launch
        # time in the region says little about the kernel.
        saved = list(estreams)
        estreams[:] = [torch.cuda.Stream(device=dev) for _ in range(NSET)]
        for _ in range(2):
            step()
        sync_streams()
        ksa = max(20, args.steps)
        t1 = time.perf_counter()
        for _ in range(ksa):
            step()
        sync_streams()
        dta = (time.perf_counter() - t1) / ksa
        alt = {"extract_streams": NSET, "value": B / dta, "unit": "frames/s", "ms_per_step": dta * 1e3, "steps": ksa}
        # ... and staggered: step i + 1's extraction gated on step i's after-FAST event (its pyramid beside step i's quadtree)
        if not fe.stagger:
            fe.stagger = True
            for _ in range(2):
                step()
            sync_streams()
            t1 = time.perf_counter()
            for _ in range(ksa):
                step()
            sync_streams()
            dts = (time.perf_counter() - t1) / ksa
            alt["staggered"] = {"value": B / dts, "ms_per_step": dts * 1e3, "steps": ksa}
            fe.stagger = False
        estreams[:] = saved
    if world == 1 and not args.no_extras:
        # ---- the same steps with the images arriving over PCIe: pinned host frames, uploaded on a copy stream into one of two
        # device buffers while the previous step computes (never `value`: the contract's number is HBM-resident)
        pinned = host_frames.view(2 * B, H, W).pin_memory()
        dbuf = [torch.empty_like(images) for _ in range(2)]
        # two copy streams, half the batch each (chunks of ~185 MB): one stream's copies reach ~35 GB/s on this pool's hosts, two ~45 (tools/h2d_bw.py)
        cstreams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        cstream = cstreams[0]
        up_half = [[torch.cuda.Event() for _ in range(2)] for _ in range(2)]
        up_done = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        ksteps = max(20, args.steps)
        half = (2 * B) // 2

        def h2d_step(i):
            b = i % 2
            for k, cs in enumerate(cstreams):
                if i >= 2:
                    cs.wait_event(consumed[b])
                lo, hi = (0, half) if k == 0 else (half, 2 * B)
                with torch.cuda.stream(cs):
                    dbuf[b][lo:hi].copy_(pinned[lo:hi], non_blocking=True)
                up_half[b][k].record(cs)
            cstream.wait_event(up_half[b][1])
            up_done[b].record(cstream)
            es = estreams[fe.nstep % NSET]
            step(src=dbuf[b], src_ready=up_done[b])
            consumed[b].record(es)
        for i in range(2):
            h2d_step(i)
        sync_streams(); cstreams[0].synchronize(); cstreams[1].synchronize()
        t1 = time.perf_counter()
        for i in range(ksteps):
            h2d_step(i)
        sync_streams(); cstreams[0].synchronize(); cstreams[1].synchronize()
        dth = (time.perf_counter() - t1) / ksteps
        h2d = {"value": B / dth, "unit": "frames/s", "ms_per_step": dth * 1e3, "steps": ksteps,
               "pcie_GBps": 2 * B * W * H / dth / 1e9,
               "note": "pinned host images uploaded on two copy streams (half the batch each), double-buffered, overlapped with the previous step's compute"}
        # ---- one stereo frame at a time, host image in -> host arrays out (the reference's call pattern, Frame.cc:190-226):
        # upload 2 images, ORBextractor x2, ComputeStereoMatches, download keypoints / descriptors / counts / uRight / depth
        one = ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
        lst = torch.cuda.Stream(device=dev)
        pin1 = host_frames[0].pin_memory()                                     # [2, H, W]
        d1 = torch.empty((2, H, W), dtype=torch.uint8, device=dev)
        o1 = so1 = None
        hk = torch.empty((2, cap, 28), dtype=torch.uint8).pin_memory(); hd = torch.empty((2, cap, 32), dtype=torch.uint8).pin_memory()
        hc = torch.empty((2,), dtype=torch.int32).pin_memory(); hu = torch.empty((1, cap), dtype=torch.float32).pin_memory()
        hz = torch.empty((1, cap), dtype=torch.float32).pin_memory()

        def one_frame():
            nonlocal o1, so1
            with torch.cuda.stream(lst):
                d1.copy_(pin1, non_blocking=True)
                o1 = one.extract_batch(d1, out=o1, stream=lst.cuda_stream)
                so1 = matcher.ComputeStereoMatches(one, o1[0], o1[1], o1[2], mbf, mb, out=so1, stream=lst.cuda_stream)
                hk.copy_(o1[0], non_blocking=True); hd.copy_(o1[1], non_blocking=True); hc.copy_(o1[2], non_blocking=True)
                hu.copy_(so1[0], non_blocking=True); hz.copy_(so1[1], non_blocking=True)
            lst.synchronize()
        for _ in range(5):
            one_frame()
        t1 = time.perf_counter()
        for _ in range(30):
            one_frame()
        lat_dev = (time.perf_counter() - t1) / 30
        # and through the host-pointer ABI entry (morb_extract == ORBextractor::operator()) on two threads like Frame.cc:194-197
        from concurrent.futures import ThreadPoolExecutor
        eL, eR = ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank), ORBextractor(NFEAT, 1.2, 8, 20, 7, device=local_rank)
        imL, imR = host_frames[0, 0].numpy(), host_frames[0, 1].numpy()
        with ThreadPoolExecutor(2) as tp:
            for _ in range(3):
                a, b_ = tp.submit(eL, imL), tp.submit(eR, imR); a.result(); b_.result()
            t1 = time.perf_counter()
            for _ in range(20):
                a, b_ = tp.submit(eL, imL), tp.submit(eR, imR); a.result(); b_.result()
            lat_host = (time.perf_counter() - t1) / 20
        lat = {"latency_b1_ms": lat_dev * 1e3, "stages": ["H2D 2 images", "extract_left+right", "ComputeStereoMatches", "D2H keypoints/descriptors/uRight/depth"],
               "extract_host_api_two_threads_ms": lat_host * 1e3,
               "note": "one stereo frame per call, host image in -> host arrays out"}
        if not args.no_cpu_baseline:
            # the CPU oracle on the same frame and stages, one thread, and with the reference's two extraction threads (Frame.cc:194-197)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            ol, orr = O.OracleExtractor(NFEAT), O.OracleExtractor(NFEAT)

            def oracle_frame(tp=None):
                if tp is None:
                    (_, kl, dl), (_, kr, dr) = ol(imL), orr(imR)
                else:
                    a, b_ = tp.submit(ol, imL), tp.submit(orr, imR)
                    (_, kl, dl), (_, kr, dr) = a.result(), b_.result()
                O.stereo_matches(ol, orr, kl, dl, kr, dr, np.float32(mbf), np.float32(mb))
            oracle_frame()
            t1 = time.perf_counter()
            for _ in range(5):
                oracle_frame()
            lat["cpu_oracle_1thread_ms"] = (time.perf_counter() - t1) / 5 * 1e3
            with ThreadPoolExecutor(2) as tp:
                t1 = time.perf_counter()
                for _ in range(5):
                    oracle_frame(tp)
                lat["cpu_oracle_2threads_left_right_ms"] = (time.perf_counter() - t1) / 5 * 1e3
        one.close(); eL.close(); eR.close()

    if rank == 0:
        fps = B * world * args.steps / dt
        ab = algorithmic_bytes(W, H)
        nimg = 2 * B
        kern = {"pyramid": "k_level0+k_resize", "blur": "k_blur", "fast": "k_fastw"}
        # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of the extraction
        # at 128 images per launch; profiles/<round>/pmc_traffic_b64.json) — not measurable live.  Corrected as the calibration kernel
        # prescribes (profiles/r03/fetch_calib.txt: FETCH_SIZE reads 0.500 x the bytes of coalesced 4 / 8 / 16-byte reads, WRITE_SIZE 1.0 x):
        # traffic = 2 x FETCH_SIZE + WRITE_SIZE
        pm = pmi = None
        pm_src = None
        for rnd, fn in (("r06", "pmc_traffic_b512.json"), ("r05", "pmc_traffic_b512.json"), ("r04", "pmc_traffic_b64.json"), ("r03", "pmc_traffic_b64.json")):   # newest first; since r05: taken AT the bench batch
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", rnd, fn)))
                pm_src = f"profiles/{rnd}/{fn}"
                break
            except Exception:
                pm = None
        pmi_src = None
        for rnd in ("r06", "r05"):
            try:
                pmi = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_issue_b512.json")))
                pmi_src = f"profiles/{rnd}/pmc_issue_b512.json"
                break
            except Exception:
                pmi = None

        def traffic_of(stage):
            if pm is None or args.workload != "c2":
                return None
            try:
                # launches per extraction: level 0, seven resizes; two k_fastw launch groups (the per-launch mean is over both)
                names = {"pyramid": [("k_level0", 1), ("k_resize", 7)], "blur": [("k_blur", 1)], "fast": [("k_fastw", 2)]}[stage]
                return sum(pm[k]["traffic_KB_per_launch"] * 1024 * m for k, m in names) * nimg / pm.get("images_per_launch", 128)
            except Exception:
                return None

        def roof(stage):
            ach = ab[stage] * nimg / (stages[stage] * 1e-3) / 1e9    # algorithmic bytes of the stage's launches / their in-region time
            return {"bound": "hbm", "kernel": kern[stage], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic_of(stage), "algorithmic_bytes_per_launch": ab[stage] * nimg, "avg_launch_ms": stages[stage],
                    # not part of the timed region: the stage alone on the chip (steps not overlapped)
                    "isolated_avg_launch_ms": iso[stage], "isolated_frac": ab[stage] * nimg / (iso[stage] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the dominant streaming stage of the extractor = the longest of the three when each runs ALONE on the chip (the same launches,
        # not overlapped with another step: `iso`).  Inside the timed region the pyramid runs beside the previous step's matchers and is
        # stretched by them (1.17 ms against 0.76 alone at B = 512), so since round 4's k_fastw (1.08 ms) the in-region maximum would name
        # a stage for what the matchers cost it.  `achieved` is still the stage's in-region time (HIP events on the launch stream; "pyramid"
        # is the dependent launches of k_level0 / k_resize, "fast" the two k_fastw launch groups); all three stages are in `stage_roofline`.
        dom = max(("pyramid", "blur", "fast"), key=lambda k: iso[k])
        wl = {"c2": f"BASELINE configs[1]: EuRoC-shaped stereo {W}x{H}, {NFEAT} feat", "c4": f"BASELINE configs[3]: synthetic stereo {W}x{H}, {NFEAT} feat",
              "vga": f"FUNCTIONAL TEST SHAPE (not a BASELINE configuration): synthetic stereo {W}x{H}, {NFEAT} feat"}[args.workload]
        cfg_tag = {"c2": "= BASELINE configs[1]", "c4": "= BASELINE configs[3]", "vga": "functional test shape, NOT a BASELINE configuration"}[args.workload]
        line = {
            "metric": f"stereo frames/sec ORB extract+match ({W}x{H}, {NFEAT} feat {cfg_tag})",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl + ": ORBextractor x2 + ComputeStereoMatches + ComputeBoW (synthetic k=10 L=6 vocabulary) + "
                                   "SearchByBoW vs previous frame",
                       "stereo_frames_per_step_per_gpu": B, "steps_in_flight": NSET, "extract_streams": len(set(id(x) for x in estreams)), "parallelism": f"frames dealt round-robin over {world} GPU(s)" + ("" if world == 1 else
                                       ("; per step one RCCL send/recv of the left-image feature slab (keypoints | descriptors | BoW ids | counts) to the next rank (ring)"
                                                                                    if args.exchange == "ring" else "; per step an RCCL all-gather of every rank's left-image features")),
                       "stages_in_step": ["extract_left+right", "stereo_match", "bow_transform"] + (["feature_exchange"] if world > 1 else []) + ["search_by_bow"],
                       "mean_keypoints_per_image": float(cnt.mean()), "mean_stereo_matches_per_frame": n_stereo,
                       "mean_bow_matches_per_frame": n_bow},
            "roofline": roof(dom),
            "stage_roofline": {k: {kk: vv for kk, vv in roof(k).items() if kk in ("kernel", "achieved", "frac", "avg_launch_ms", "isolated_avg_launch_ms", "isolated_frac", "traffic")}
                               for k in ("pyramid", "blur", "fast")},
            "extract_stage_ms_per_step": stages,
        }
        if pm_src is not None:
            line["roofline"]["traffic_source"] = pm_src + (" (counters at this batch)" if pm.get("images_per_launch") == nimg else
                                                           f" (counters at {pm.get('images_per_launch')} images per launch, scaled)")
        line["roofline"]["peak_measured_copy"] = 6290.0      # GB/s, the guide's measured HBM copy rate beside the 8 TB/s specification
        line["roofline"]["frac_of_measured_copy"] = line["roofline"]["achieved"] / 6290.0
        if pmi is not None and args.workload == "c2" and "k_fastw" in pmi and pmi.get("images_per_launch") == nimg:
            # k_fastw is bound by vector-instruction ISSUE, not by HBM (traffic = 1.07 x algorithmic, every instruction removed is
            # time): its real roof.  Instruction counts from the committed counter pass at this batch (not measurable live), duration measured live.
            kf = pmi["k_fastw"]
            per_step = kf["valu_insts_per_launch"] * 2             # two launch groups per extraction
            simd_cycles = pmi["simds"] * stages["fast"] * 1e-3 * pmi["clock_GHz"] * 1e9
            # measured quantities only (round 5 reported min(1, instructions x a separately measured 4.1 cycles / cycles), a model that overshot its own
            # ceiling): SIMD cycles available per VALU wave-instruction of the stage, live, and the utilisation that follows from the FASTEST issue
            # rate this chip has for a wave64 VALU instruction (4 cycles on a 16-lane SIMD) — a lower bound on how busy the issue port is, since the
            # packed / three-operand / DPP forms the kernel is made of take longer.  `roofline.frac` stays the HBM figure.
            line["roofline_issue"] = {"bound": "valu_issue", "kernel": "k_fastw", "valu_wave_instructions_per_step": per_step,
                                      "simd_cycles_per_valu_instruction": simd_cycles / per_step,
                                      "issue_utilisation": 4.0 / (simd_cycles / per_step), "issue_utilisation_basis": "4.0 SIMD cycles per wave64 VALU instruction (the minimum) at an ASSUMED 2.4 GHz; uncapped — a value a few per cent above 1 "
                                                                 "is the error of that clock assumption / of the counter pass taken at another time, not a faster chip",
                                      "valu_lane_slots_per_pixel": per_step * 64 / (ab["fast"] * nimg),
                                      "valu_per_cell_wave": kf["valu_per_wave"], "salu_per_cell_wave": kf["salu_per_wave"], "lds_per_cell_wave": kf["lds_per_wave"],
                                      "source": f"{pmi_src} (SQ_INSTS_VALU per launch) / live k_fastw time x 1024 SIMDs x 2.4 GHz"}
        if multi is not None:
            line["multi_gpu"] = multi
        if verified is not None:
            line.update(verified_frames=verified["verified_frames"], verified=verified)
        if sustained is not None:
            line["sustained"] = sustained
        if alt is not None:
            line["two_extraction_streams"] = alt
        if h2d is not None:
            line["h2d_inclusive"] = h2d
        if lat is not None:
            line["latency_b1"] = lat
        if world == 1 and not args.no_extras:
            line["extra_metrics"] = optimizer_extras(local_rank)
            line["extra_metrics"].update(tracking_extras(local_rank))
            if args.workload == "c2":
                line["extra_metrics"].update(config_extras(local_rank))
        if world == 1 and not args.no_cpu_baseline:      # reported on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and lat is not None:
            # What a drop-in sees: the reference calls this path ONE frame at a time (src/Frame.cc:190-226 extraction + ComputeStereoMatches per frame,
            # src/Tracking.cc:2655-2710 TrackWithMotionModel / TrackLocalMap per frame, src/LocalMapping.cc:173-181 one LocalBundleAdjustment per new
            # keyframe), host arrays in -> host arrays out.  Every entry: the GPU call (milliseconds, wall clock around the blocking call) and the CPU
            # oracle on ONE thread on the same input.  `value` above is batch throughput; these are latencies and are not comparable with it.
            em = line.get("extra_metrics", {})
            tc, lb = em.get("tracking_chain", {}), em.get("local_ba", {})
            cb = line.get("cpu_baseline", {})
            L = {"call_pattern": "one frame / one keyframe per call, host in -> host out (Frame.cc:190-226, Tracking.cc:2655-2710, LocalMapping.cc:173-181)",
                 "front_end_one_stereo_frame": {"gpu_ms": lat["latency_b1_ms"], "stages": lat["stages"], "cpu_oracle_1thread_ms": lat.get("cpu_oracle_1thread_ms"),
                                                "cpu_oracle_2threads_left_right_ms": lat.get("cpu_oracle_2threads_left_right_ms"),
                                                "gpu_extract_only_host_api_two_threads_ms": lat["extract_host_api_two_threads_ms"]}}
            if tc:
                L["tracking_chain_one_frame"] = {"gpu_ms": tc["latency_b1_ms"], "stages": tc["stages"], "mfma_chain": tc.get("mfma_chain"),
                                                 "cpu_oracle_1thread_ms": 1e3 / tc["cpu_oracle_frames_per_s_1core"]}
            if lb:
                L["local_ba_one_shot_call"] = {"gpu_ms": lb["one_shot_call_ms"], "gpu_resident_solve_ms": lb["ms_per_solve"], "keyframes_free_fixed": lb["keyframes_free_fixed"],
                                               "points": lb["points"], "edges": lb["edges"], "cpu_oracle_1thread_ms": lb["cpu_oracle_ms_per_solve_1core"]}
            if h2d is not None:
                L["h2d_inclusive_stream"] = {"gpu_frames_per_s": h2d["value"], "pcie_GBps": h2d["pcie_GBps"], "stereo_frames_per_step": B,
                                             "cpu_oracle_frames_per_s_1thread": cb.get("one_thread", {}).get("value"),
                                             "cpu_oracle_frames_per_s_all_threads": cb.get("value"), "cpu_threads": cb.get("cores"),
                                             "note": "the headline chain with the images arriving over PCIe each step (pinned host memory, copy streams overlapped with compute)"}
            line["latency"] = L
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
