"""Seeded synthetic inputs: there is no dataset and no network, so the tests and bench.py
generate EuRoC-shaped images here.  Pure numpy; deterministic for a given seed."""
import numpy as np


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_image(w=752, h=480, seed=0, n_shapes=None, noise_sigma=2.0):
    """u8 HxW image: 3 octaves of value noise + random rectangles/discs + Gaussian pixel noise."""
    rng = np.random.default_rng(0x4D0B + seed)
    img = 96 * _value_noise(rng, h, w, 64) + 64 * _value_noise(rng, h, w, 16) + 32 * _value_noise(rng, h, w, 4)
    img += 20
    if n_shapes is None:
        n_shapes = int(400 * (w * h) / 360960)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_shapes):
        cx, cy = rng.integers(0, w), rng.integers(0, h)
        s = int(rng.integers(3, 40))
        grey = float(rng.integers(0, 256))
        kind = rng.integers(0, 3)
        x0, x1 = max(cx - s, 0), min(cx + s + 1, w)
        y0, y1 = max(cy - s, 0), min(cy + s + 1, h)
        if x1 <= x0 or y1 <= y0:
            continue
        sub = img[y0:y1, x0:x1]
        if kind == 0:
            sub[:] = grey
        elif kind == 1:
            m = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= s * s
            sub[m] = grey
        else:  # rotated rectangle
            th = rng.random() * np.pi
            u = (xx[y0:y1, x0:x1] - cx) * np.cos(th) + (yy[y0:y1, x0:x1] - cy) * np.sin(th)
            v = -(xx[y0:y1, x0:x1] - cx) * np.sin(th) + (yy[y0:y1, x0:x1] - cy) * np.cos(th)
            m = (np.abs(u) <= s * 0.8) & (np.abs(v) <= s * 0.4)
            sub[m] = grey
    img += rng.normal(0, noise_sigma, size=img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def make_stereo_pair(w=752, h=480, seed=0, dmin=2.0, dmax=60.0):
    """(left, right): right = left warped by a smooth horizontal disparity field d(x,y) in [dmin, dmax]
    (left pixel (x,y) appears at (x-d, y) in the right image), plus independent pixel noise."""
    left = make_image(w, h, seed)
    rng = np.random.default_rng(0x57E0 + seed)
    d = dmin + (dmax - dmin) * _value_noise(rng, h, w, 128)
    xs = np.arange(w, dtype=np.float32)[None, :] + d  # right(x) samples left(x + d)
    x0 = np.clip(np.floor(xs).astype(np.int32), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    f = xs - np.floor(xs)
    rows = np.arange(h)[:, None]
    L = left.astype(np.float32)
    right = L[rows, x0] * (1 - f) + L[rows, x1] * f + rng.normal(0, 1.0, size=L.shape)
    return left, np.ascontiguousarray(np.clip(np.rint(right), 0, 255).astype(np.uint8))


def shift_image(img, dx, dy):
    """Integer global shift with edge replication (feeds the frame-to-frame matchers)."""
    h, w = img.shape
    ys = np.clip(np.arange(h) - dy, 0, h - 1)
    xs = np.clip(np.arange(w) - dx, 0, w - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def make_vocabulary(k=10, L=6, seed=0):
    """Synthetic DBoW2-shaped ORB vocabulary (the real ORBvoc.txt is a missing blob, SURVEY.md finding 3):
    complete k-ary tree of depth L in level order (node 0 = root, children of n = k*n+1 .. k*n+k), each child
    descriptor = parent descriptor with a level-dependent fraction of random bits flipped.
    Returns (nodeDesc [nnodes, 32] u8, firstChild [nnodes] i32, -1 for leaves)."""
    rng = np.random.default_rng(0xB0 + seed)
    nn = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((nn, 32), np.uint8)
    first = np.full(nn, -1, np.int32)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    start, cnt = 0, 1
    for lvl in range(L):
        parents = np.arange(start, start + cnt)
        first[parents] = k * parents + 1
        p = 0.5 / (lvl + 1.5)
        child_ids = (k * parents[:, None] + 1 + np.arange(k)[None, :]).reshape(-1)
        flips = np.packbits(rng.random((len(child_ids), 256)) < p, axis=1)
        desc[child_ids] = np.repeat(desc[parents], k, axis=0) ^ flips
        start, cnt = start + cnt, cnt * k
    return desc, first


# ---- synthetic geometry for the optimisers ------------------------------------------------
EUROC_CAM = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, bf=458.654 * 0.11)


def _quat_from_rotvec(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.array([0, 0, 0, 1.0])
    a = r / th
    return np.concatenate([a * np.sin(th / 2), [np.cos(th / 2)]])


def _quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _quat_rot(q, v):
    u, w = q[:3], q[3]
    uv = 2 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def _project(pose, X, cam):
    """pose = (qx,qy,qz,qw,tx,ty,tz) world->camera; returns (u, v, uRight, z)."""
    Xc = np.array([_quat_rot(pose[:4], x) for x in X]) + pose[4:]
    z = Xc[:, 2]
    u = cam["fx"] * Xc[:, 0] / z + cam["cx"]
    v = cam["fy"] * Xc[:, 1] / z + cam["cy"]
    return u, v, u - cam["bf"] / z, z


def make_pose_problem(n=600, seed=0, outlier_frac=0.1, mono_frac=0.3, rot_deg=2.0, trans=0.05, cam=EUROC_CAM):
    """One PoseOptimization input: n features, ~all with a MapPoint, stereo/mono mix, gross outliers, perturbed
    initial pose.  Returns dict of float32 arrays + the true pose."""
    rng = np.random.default_rng(0x9050 + seed)
    true = np.concatenate([_quat_from_rotvec(rng.normal(0, 0.1, 3)), rng.normal(0, 0.3, 3)])
    # points in front of the camera: sample camera-frame points and map them to the world
    Xc = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2, 10, n)], 1)
    qinv = true[:4] * np.array([-1, -1, -1, 1])
    Xw = np.array([_quat_rot(qinv, x - true[4:]) for x in Xc])
    u, v, ur, z = _project(true, Xw, cam)
    octave = rng.integers(0, 8, n)
    sigma = 1.2 ** octave
    u = u + rng.normal(0, 1, n) * sigma; v = v + rng.normal(0, 1, n) * sigma; ur = ur + rng.normal(0, 1, n) * sigma
    out = rng.random(n) < outlier_frac
    u[out] += rng.choice([-1, 1], out.sum()) * rng.uniform(15, 40, out.sum())
    v[out] += rng.choice([-1, 1], out.sum()) * rng.uniform(15, 40, out.sum())
    mono = rng.random(n) < mono_frac
    ur[mono] = -1
    has = (rng.random(n) < 0.9).astype(np.uint8)
    dq = _quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(rot_deg))
    init = np.concatenate([_quat_mul(dq, true[:4]), _quat_rot(dq, true[4:]) + rng.normal(0, trans, 3)])
    return dict(hasMP=has, obs=np.stack([u, v, ur], 1).astype(np.float32), invSigma2=(1.0 / sigma ** 2).astype(np.float32),
                Xw=Xw.astype(np.float32), pose0=init.astype(np.float32), true=true, outlier_truth=out, cam=cam)


def make_ba_problem(n_free=20, n_fixed=6, n_points=3000, seed=0, outlier_frac=0.05, mono_frac=0.15, cam=EUROC_CAM):
    """LocalBundleAdjustment input shaped like BASELINE config 5: n_free + n_fixed keyframes on an arc looking at
    a point slab, each point observed by 4-10 keyframes."""
    rng = np.random.default_rng(0xBA00 + seed)
    nkf = n_free + n_fixed
    poses = []
    for i in range(nkf):
        ang = (i / max(nkf - 1, 1) - 0.5) * 0.5
        c = np.array([2.0 * np.sin(ang) * 2, rng.normal(0, 0.05), -2.0 * (1 - np.cos(ang))])   # camera centre
        q_wc = _quat_from_rotvec(np.array([0, -ang * 0.8, 0]) + rng.normal(0, 0.01, 3))      # camera->world
        q_cw = q_wc * np.array([-1, -1, -1, 1])
        poses.append(np.concatenate([q_cw, -_quat_rot(q_cw, c)]))
    poses = np.array(poses)
    X = np.stack([rng.uniform(-3, 3, n_points), rng.uniform(-2, 2, n_points), rng.uniform(3, 10, n_points)], 1)
    eKF, eMP, eObs, eInv = [], [], [], []
    for j in range(n_points):
        k = int(rng.integers(4, 11))
        kfs = rng.choice(nkf, size=min(k, nkf), replace=False)
        u, v, ur, z = _project_many(poses[kfs], X[j], cam)
        for a, kf in enumerate(kfs):
            if z[a] <= 0.5 or not (0 <= u[a] < 752 and 0 <= v[a] < 480):
                continue
            octv = int(rng.integers(0, 8)); s = 1.2 ** octv
            o = np.array([u[a], v[a], ur[a]]) + rng.normal(0, 1, 3) * s
            if rng.random() < outlier_frac:
                o[:2] += rng.choice([-1, 1], 2) * rng.uniform(15, 30, 2)
            if rng.random() < mono_frac:
                o[2] = -1
            eKF.append(kf); eMP.append(j); eObs.append(o); eInv.append(1.0 / s ** 2)
    fixed = np.zeros(nkf, np.uint8); fixed[n_free:] = 1
    pose0 = poses.copy()
    for i in range(n_free):
        dq = _quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(2.0))
        pose0[i] = np.concatenate([_quat_mul(dq, poses[i][:4]), _quat_rot(dq, poses[i][4:]) + rng.normal(0, 0.05, 3)])
    X0 = X + rng.normal(0, 0.05, X.shape)
    return dict(kfPose=pose0.astype(np.float32), kfFixed=fixed, mpPos=X0.astype(np.float32),
                eKF=np.array(eKF, np.int32), eMP=np.array(eMP, np.int32), eObs=np.array(eObs, np.float32),
                eInvSigma2=np.array(eInv, np.float32), true_poses=poses, true_points=X, cam=cam)


def _project_many(poses, x, cam):
    Xc = np.array([_quat_rot(p[:4], x) + p[4:] for p in poses])
    z = Xc[:, 2]
    u = cam["fx"] * Xc[:, 0] / z + cam["cx"]
    v = cam["fy"] * Xc[:, 1] / z + cam["cy"]
    return u, v, u - cam["bf"] / z, z


# ---- fisheye rig (TUM-VI, Examples/Stereo/TUM-VI.yaml) -------------------------------------------------------
TUMVI_CAM_L = np.array([190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504,
                        0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182], np.float32)
TUMVI_CAM_R = np.array([190.44236969414825, 190.4344384721956, 252.59949716835982, 254.91723064636983,
                        0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039], np.float32)
TUMVI_T_C1_C2 = np.array([[0.999999445773493, 0.000791687752817, 0.000694034010224, 0.101063427414194],
                          [-0.000823363992158, 0.998899461915674, 0.046895490788700, 0.001946204678584],
                          [-0.000656143613644, -0.046896036240590, 0.998899560146304, 0.001015350132563],
                          [0, 0, 0, 1.0]])   # Stereo.T_c1_c2 = Tlr (right-camera coordinates -> left-camera coordinates)


def kb8_project(cam, X):
    """KannalaBrandt8::project in float64 numpy (reference formula) for an [n, 3] array."""
    x, y, z = X[:, 0], X[:, 1], X[:, 2]
    th = np.arctan2(np.sqrt(x * x + y * y), z); psi = np.arctan2(y, x)
    r = th + cam[4] * th ** 3 + cam[5] * th ** 5 + cam[6] * th ** 7 + cam[7] * th ** 9
    return np.stack([cam[0] * r * np.cos(psi) + cam[2], cam[1] * r * np.sin(psi) + cam[3]], 1)


def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def make_fisheye_features(n_pairs=600, n_mono=80, n_distract=150, seed=0):
    """Synthetic fisheye stereo feature sets (no images): n_pairs 3-D points seen by both KB8 cameras with matching
    descriptors (a few bits flipped), mono-area features in front of the arrays, unmatched distractors behind."""
    rng = np.random.default_rng(0xF15E + seed)
    Tlr = TUMVI_T_C1_C2
    Trl = np.linalg.inv(Tlr)
    X = np.stack([rng.uniform(-4, 4, n_pairs), rng.uniform(-3, 3, n_pairs), rng.uniform(0.8, 8, n_pairs)], 1)
    uvL = kb8_project(TUMVI_CAM_L, X) + rng.normal(0, 0.2, (n_pairs, 2))
    Xr = X @ Trl[:3, :3].T + Trl[:3, 3]
    uvR = kb8_project(TUMVI_CAM_R, Xr) + rng.normal(0, 0.2, (n_pairs, 2))
    ok = (uvL > 20).all(1) & (uvL < 492).all(1) & (uvR > 20).all(1) & (uvR < 492).all(1) & (Xr[:, 2] > 0.3)
    uvL, uvR = uvL[ok], uvR[ok]
    m = len(uvL)
    base = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    flip = np.packbits(rng.random((m, 256)) < 0.03, axis=1)
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

    def side(uv, desc, nm, nd):
        n = nm + len(uv) + nd
        k = np.zeros(n, dt); d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        k["x"] = rng.uniform(20, 492, n); k["y"] = rng.uniform(20, 492, n)
        k["octave"] = rng.integers(0, 8, n); k["size"] = 31; k["angle"] = rng.uniform(0, 360, n); k["class_id"] = -1
        perm = rng.permutation(len(uv))
        k["x"][nm:nm + len(uv)] = uv[perm, 0]; k["y"][nm:nm + len(uv)] = uv[perm, 1]
        d[nm:nm + len(uv)] = desc[perm]
        return k, d, nm

    kL, dL, monoL = side(uvL, base, n_mono, n_distract)
    kR, dR, monoR = side(uvR, base ^ flip, n_mono // 2, n_distract)
    return dict(kL=kL, dL=dL, monoL=monoL, kR=kR, dR=dR, monoR=monoR, Rlr=Tlr[:3, :3].astype(np.float32),
                tlr=Tlr[:3, 3].astype(np.float32), camL=TUMVI_CAM_L, camR=TUMVI_CAM_R)


def make_pose_problem_fisheye(n_left=400, n_right=300, seed=0, outlier_frac=0.1, rot_deg=2.0, trans=0.05):
    """PoseOptimization input for the fisheye rig: left-camera and right-camera observations of map points."""
    rng = np.random.default_rng(0xF0E5 + seed)
    Trl_m = np.linalg.inv(TUMVI_T_C1_C2)
    true = np.concatenate([_quat_from_rotvec(rng.normal(0, 0.1, 3)), rng.normal(0, 0.3, 3)])
    n = n_left + n_right
    Xc = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(1, 8, n)], 1)
    qinv = true[:4] * np.array([-1, -1, -1, 1])
    Xw = np.array([_quat_rot(qinv, x - true[4:]) for x in Xc])
    uv = np.zeros((n, 2))
    uv[:n_left] = kb8_project(TUMVI_CAM_L, Xc[:n_left])
    Xr = Xc[n_left:] @ Trl_m[:3, :3].T + Trl_m[:3, 3]
    uv[n_left:] = kb8_project(TUMVI_CAM_R, Xr)
    octave = rng.integers(0, 8, n); sigma = 1.2 ** octave
    uv += rng.normal(0, 0.5, (n, 2)) * sigma[:, None]
    out = rng.random(n) < outlier_frac
    uv[out] += rng.choice([-1, 1], (out.sum(), 2)) * rng.uniform(15, 40, (out.sum(), 2))
    dq = _quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(rot_deg))
    init = np.concatenate([_quat_mul(dq, true[:4]), _quat_rot(dq, true[4:]) + rng.normal(0, trans, 3)])
    obs = np.concatenate([uv, np.zeros((n, 1))], 1).astype(np.float32)
    Trl7 = np.concatenate([_quat_from_R(Trl_m[:3, :3]), Trl_m[:3, 3]]).astype(np.float32)
    return dict(hasMP=(rng.random(n) < 0.92).astype(np.uint8), obs=obs, invSigma2=(1 / sigma ** 2).astype(np.float32),
                Xw=Xw.astype(np.float32), pose0=init.astype(np.float32), true=true, Nleft=n_left, camL=TUMVI_CAM_L, camR=TUMVI_CAM_R,
                Trl=Trl7)


def make_ba_problem_fisheye(n_free=10, n_fixed=4, n_points=1500, seed=0, outlier_frac=0.05, right_frac=0.45):
    """LocalBundleAdjustment input on the TUM-VI KannalaBrandt8 rig: left-camera observations (EdgeSE3ProjectXYZ) and
    right-camera observations behind Trl (EdgeSE3ProjectXYZToBody)."""
    rng = np.random.default_rng(0xFBA0 + seed)
    Trl_m = np.linalg.inv(TUMVI_T_C1_C2)
    nkf = n_free + n_fixed
    poses = []
    for i in range(nkf):
        ang = (i / max(nkf - 1, 1) - 0.5) * 0.5
        c = np.array([2.0 * np.sin(ang) * 2, rng.normal(0, 0.05), -2.0 * (1 - np.cos(ang))])
        q_wc = _quat_from_rotvec(np.array([0, -ang * 0.8, 0]) + rng.normal(0, 0.01, 3))
        q_cw = q_wc * np.array([-1, -1, -1, 1])
        poses.append(np.concatenate([q_cw, -_quat_rot(q_cw, c)]))
    poses = np.array(poses)
    X = np.stack([rng.uniform(-3, 3, n_points), rng.uniform(-2, 2, n_points), rng.uniform(2, 8, n_points)], 1)
    eKF, eMP, eObs, eInv, eRight = [], [], [], [], []
    for j in range(n_points):
        kfs = rng.choice(nkf, size=min(int(rng.integers(4, 9)), nkf), replace=False)
        for kf in kfs:
            Xc = _quat_rot(poses[kf][:4], X[j]) + poses[kf][4:]
            for right in (0, 1):
                if right and rng.random() > right_frac:
                    continue
                Xs = Trl_m[:3, :3] @ Xc + Trl_m[:3, 3] if right else Xc
                if Xs[2] <= 0.4:
                    continue
                uv = kb8_project(TUMVI_CAM_R if right else TUMVI_CAM_L, Xs[None])[0]
                if not (5 < uv[0] < 507 and 5 < uv[1] < 507):
                    continue
                octv = int(rng.integers(0, 8)); sg = 1.2 ** octv
                o = uv + rng.normal(0, 0.7, 2) * sg
                if rng.random() < outlier_frac:
                    o += rng.choice([-1, 1], 2) * rng.uniform(15, 30, 2)
                eKF.append(kf); eMP.append(j); eObs.append(o); eInv.append(1.0 / sg ** 2); eRight.append(right)
    fixed = np.zeros(nkf, np.uint8); fixed[n_free:] = 1
    pose0 = poses.copy()
    for i in range(n_free):
        dq = _quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(1.5))
        pose0[i] = np.concatenate([_quat_mul(dq, poses[i][:4]), _quat_rot(dq, poses[i][4:]) + rng.normal(0, 0.03, 3)])
    X0 = X + rng.normal(0, 0.04, X.shape)
    Trl7 = np.concatenate([_quat_from_R(Trl_m[:3, :3]), Trl_m[:3, 3]]).astype(np.float32)
    return dict(kfPose=pose0.astype(np.float32), kfFixed=fixed, mpPos=X0.astype(np.float32), eKF=np.array(eKF, np.int32),
                eMP=np.array(eMP, np.int32), eObs=np.array(eObs, np.float32), eInvSigma2=np.array(eInv, np.float32),
                eRight=np.array(eRight, np.uint8), camL=TUMVI_CAM_L, camR=TUMVI_CAM_R, Trl=Trl7, true_poses=poses, true_points=X)


# ---- visual-inertial tracking (PoseInertialOptimizationLastKeyFrame) ------------------------------------------------------
def _rot_from_rotvec(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-9:
        return np.eye(3) + K
    return np.eye(3) + K * np.sin(th) / th + K @ K * (1 - np.cos(th)) / th ** 2


# EuRoC-like IMU: body -> camera-0 extrinsics, noise densities at 200 Hz (Examples/Stereo-Inertial/EuRoC.yaml)
EUROC_TBC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                      [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                      [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                      [0, 0, 0, 1]])
IMU_FREQ = 200.0
IMU_NOISE = dict(ng=1.7e-4, na=2.0e-3, ngw=1.9393e-05, naw=3.0e-3)


def imu_calib_diagonals(freq=IMU_FREQ, noise=IMU_NOISE):
    """Diagonals of IMU::Calib::Cov / CovWalk the way Tracking builds them (Tracking.cc ParseIMUParamFile:
    Calib(Tbc, Ng * sqrt(freq), Na * sqrt(freq), Ngw / sqrt(freq), Naw / sqrt(freq)); ImuTypes.cc:375-388)."""
    sf = np.sqrt(freq)
    ng, na, ngw, naw = noise["ng"] * sf, noise["na"] * sf, noise["ngw"] / sf, noise["naw"] / sf
    return (np.array([ng * ng] * 3 + [na * na] * 3, np.float32), np.array([ngw * ngw] * 3 + [naw * naw] * 3, np.float32))


def make_inertial_sequence(n=500, seed=0, n_imu=20, **kw):
    """Keyframe -> frame A -> frame B: (pA, pB).  pA is a PoseInertialOptimizationLastKeyFrame input; pB a
    PoseInertialOptimizationLastFrame input whose IMU samples continue the same motion: accF / gyroF / dtF since frame A
    (mpImuPreintegratedFrame) and acc / gyro / dt since the keyframe (mpImuPreintegrated).  The previous-frame state and
    prior of pB are frame A's optimisation results (not part of the dict)."""
    pA = make_inertial_problem(n, seed, n_imu, **kw)
    pB = make_inertial_problem(n, seed, 2 * n_imu, _obs_seed=1, **kw)
    assert np.array_equal(pA["acc"], pB["acc"][:n_imu])      # same motion and noise stream
    pB["accF"], pB["gyroF"], pB["dtF"] = pB["acc"][n_imu:], pB["gyro"][n_imu:], pB["dt"][n_imu:]
    return pA, pB


def tumvi_rig28():
    """Fisheye rig as the inertial entry points take it: left KB8 (8), right KB8 (8), Trl rotation (9, row-major) +
    translation (3)."""
    Trl = np.linalg.inv(TUMVI_T_C1_C2)
    return np.concatenate([TUMVI_CAM_L, TUMVI_CAM_R, Trl[:3, :3].ravel(), Trl[:3, 3]]).astype(np.float32)


def make_inertial_problem(n=500, seed=0, n_imu=20, outlier_frac=0.1, mono_frac=0.3, rot_deg=1.0, trans=0.03, cam=EUROC_CAM,
                          _obs_seed=0, rig=False):
    """One PoseInertialOptimizationLastKeyFrame input: the last keyframe's state, n_imu IMU samples of a smooth motion
    (constant body angular rate, constant world acceleration) between keyframe and frame, map points seen from the
    frame's true pose, and a perturbed initial frame state.  States: Rwb (9), twb, v, bg, ba."""
    rng = np.random.default_rng(0x1A70 + seed)
    g = np.array([0, 0, -9.81])
    R1 = _rot_from_rotvec(rng.normal(0, 0.3, 3)); p1 = rng.normal(0, 0.5, 3); v1 = rng.normal(0, 0.8, 3)
    bg = rng.normal(0, 0.01, 3); ba = rng.normal(0, 0.05, 3)
    w_b = rng.normal(0, 0.4, 3); a_w = rng.normal(0, 1.0, 3)
    dt = 1.0 / IMU_FREQ
    acc, gyro, dts = [], [], []
    for i in range(n_imu):
        tm = (i + 0.5) * dt
        Rm = R1 @ _rot_from_rotvec(w_b * tm)
        acc.append(Rm.T @ (a_w - g) + ba + rng.normal(0, 0.02, 3))
        gyro.append(w_b + bg + rng.normal(0, 0.002, 3))
        dts.append(dt)
    T = n_imu * dt
    rng = np.random.default_rng([0x1A71 + seed, _obs_seed, n_imu])   # map points / observations / initial guess
    R2 = R1 @ _rot_from_rotvec(w_b * T); v2 = v1 + a_w * T; p2 = p1 + v1 * T + 0.5 * a_w * T * T
    Tbc = EUROC_TBC
    Rcb = Tbc[:3, :3].T; tcb = -Rcb @ Tbc[:3, 3]
    Rcw = Rcb @ R2.T; tcw = Rcb @ (-R2.T @ p2) + tcb
    Xc = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(1.5, 25, n)], 1)
    Xw = (Xc - tcw) @ Rcw          # Rcw^T (Xc - tcw)
    z = Xc[:, 2]
    octave = rng.integers(0, 8, n); sigma = 1.2 ** octave
    n_left = n
    if rig:   # fisheye rig: features [0, n_left) seen by the left KB8 camera, the rest by the right one; all monocular
        n_left = int(0.55 * n)
        Trl = np.linalg.inv(TUMVI_T_C1_C2)
        uv = np.zeros((n, 2))
        uv[:n_left] = kb8_project(TUMVI_CAM_L, Xc[:n_left])
        uv[n_left:] = kb8_project(TUMVI_CAM_R, Xc[n_left:] @ Trl[:3, :3].T + Trl[:3, 3])
        u, v, ur = uv[:, 0], uv[:, 1], np.full(n, -1.0)
        sigma = sigma * 0.5
    else:
        u = cam["fx"] * Xc[:, 0] / z + cam["cx"]; v = cam["fy"] * Xc[:, 1] / z + cam["cy"]; ur = u - cam["bf"] / z
    u = u + rng.normal(0, 1, n) * sigma; v = v + rng.normal(0, 1, n) * sigma; ur = ur + rng.normal(0, 1, n) * sigma
    out = rng.random(n) < outlier_frac
    u[out] += rng.choice([-1, 1], out.sum()) * rng.uniform(15, 40, out.sum())
    v[out] += rng.choice([-1, 1], out.sum()) * rng.uniform(15, 40, out.sum())
    mono = (rng.random(n) < mono_frac) | rig
    ur[mono] = -1
    has = (rng.random(n) < 0.9).astype(np.uint8)
    R0 = R2 @ _rot_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(rot_deg))
    state0 = np.concatenate([R0.ravel(), p2 + rng.normal(0, trans, 3), v2 + rng.normal(0, 0.05, 3), bg, ba]).astype(np.float32)
    kf = np.concatenate([R1.ravel(), p1, v1, bg, ba]).astype(np.float32)
    return dict(hasMP=has, obs=np.stack([u, v, ur], 1).astype(np.float32), invSigma2=(1.0 / sigma ** 2).astype(np.float32),
                Xw=Xw.astype(np.float32), close=(z < 10).astype(np.uint8), state0=state0, kfState=kf,
                acc=np.array(acc, np.float32), gyro=np.array(gyro, np.float32), dt=np.array(dts, np.float32),
                bias=np.concatenate([ba, bg]).astype(np.float32),      # IMU::Bias order: acc then gyro
                Tbc12=np.concatenate([Tbc[:3, :3].ravel(), Tbc[:3, 3]]).astype(np.float32), cam=cam,
                true=np.concatenate([R2.ravel(), p2, v2, bg, ba]), outlier_truth=out, Nleft=n_left,
                rig28=tumvi_rig28() if rig else None)


def make_inertial_ba_problem(n_opt=10, n_fixed_vis=6, n_points=1500, n_imu=40, seed=0, outlier_frac=0.03, mono_frac=0.25,
                             cam=EUROC_CAM, rig=False):
    """LocalInertialBA input: a temporal chain keyframe 0 (fixed, with IMU state) -> n_opt optimizable keyframes, plus
    n_fixed_vis fixed keyframes that only observe points.  Each link carries n_imu IMU samples of a smooth motion
    (per-link constant body rate and world acceleration).  States: Rwb (9), twb, v, bg, ba; kfKind 0 / 1 / 2 as in
    morb_local_inertial_ba."""
    rng = np.random.default_rng(0x1BA0 + seed)
    g = np.array([0, 0, -9.81])
    dt = 1.0 / IMU_FREQ
    bg = rng.normal(0, 0.01, 3); ba = rng.normal(0, 0.05, 3)
    Tbc = EUROC_TBC
    Rcb = Tbc[:3, :3].T; tcb = -Rcb @ Tbc[:3, 3]
    # the camera looks along +z of the camera frame: start with the body oriented so that the camera faces world +x
    R = _rot_from_rotvec(rng.normal(0, 0.1, 3)); p = rng.normal(0, 0.2, 3); v = np.array([0.6, 0.1, 0.0]) + rng.normal(0, 0.1, 3)
    states = [(R, p, v)]
    acc, gyro, dts, start = [], [], [], [0]
    for k in range(n_opt):
        w_b = rng.normal(0, 0.15, 3); a_w = rng.normal(0, 0.5, 3)
        for i in range(n_imu):
            tm = (i + 0.5) * dt
            Rm = R @ _rot_from_rotvec(w_b * tm)
            acc.append(Rm.T @ (a_w - g) + ba + rng.normal(0, 0.02, 3))
            gyro.append(w_b + bg + rng.normal(0, 0.002, 3))
            dts.append(dt)
        T = n_imu * dt
        R, p, v = R @ _rot_from_rotvec(w_b * T), p + v * T + 0.5 * a_w * T * T, v + a_w * T
        states.append((R, p, v))
        start.append(len(dts))
    # visual-only fixed keyframes: near the start of the chain, slightly displaced
    for k in range(n_fixed_vis):
        R0, p0, _ = states[rng.integers(0, 3)]
        states.append((R0 @ _rot_from_rotvec(rng.normal(0, 0.05, 3)), p0 + rng.normal(0, 0.15, 3), np.zeros(3)))
    nKF = len(states)
    kind = np.array([1] + [0] * n_opt + [2] * n_fixed_vis, np.uint8)
    cams = [(Rcb @ Rk.T, Rcb @ (-Rk.T @ pk) + tcb) for Rk, pk, _ in states]
    # points in front of the middle keyframe's camera
    Rm, tm_ = cams[n_opt // 2]
    Xc = np.stack([rng.uniform(-4, 4, n_points), rng.uniform(-2.5, 2.5, n_points), rng.uniform(2, 20, n_points)], 1)
    X = (Xc - tm_) @ Rm
    eKF, eMP, eObs, eInv, eRight = [], [], [], [], []
    Trl = np.linalg.inv(TUMVI_T_C1_C2)
    depth_ref = np.full(n_points, 1e9)
    for j in range(n_points):
        ks = rng.choice(nKF, size=min(int(rng.integers(3, 9)), nKF), replace=False)
        for k in ks:
            Rcw, tcw = cams[k]
            xc = Rcw @ X[j] + tcw
            if xc[2] < 0.5:
                continue
            octv = int(rng.integers(0, 8)); s = 1.2 ** octv
            right = 0
            if rig:   # one monocular observation on the left or on the right KB8 camera
                right = int(rng.random() < 0.45)
                xr = Trl[:3, :3] @ xc + Trl[:3, 3] if right else xc
                if xr[2] < 0.5:
                    continue
                u, vv = kb8_project(TUMVI_CAM_R if right else TUMVI_CAM_L, xr[None])[0]
                if not (0 <= u < 512 and 0 <= vv < 512):
                    continue
                s *= 0.5
                o = np.array([u, vv, -1.0]) + np.array([rng.normal(0, 1) * s, rng.normal(0, 1) * s, 0.0])
            else:
                u = cam["fx"] * xc[0] / xc[2] + cam["cx"]; vv = cam["fy"] * xc[1] / xc[2] + cam["cy"]
                if not (0 <= u < 752 and 0 <= vv < 480):
                    continue
                o = np.array([u, vv, u - cam["bf"] / xc[2]]) + rng.normal(0, 1, 3) * s
            if rng.random() < outlier_frac:
                o[:2] += rng.choice([-1, 1], 2) * rng.uniform(15, 30, 2)
            if not rig and rng.random() < mono_frac:
                o[2] = -1
            eKF.append(k); eMP.append(j); eObs.append(o); eInv.append(1.0 / (s * s)); eRight.append(right)
            depth_ref[j] = min(depth_ref[j], xc[2])
    true = np.stack([np.concatenate([Rk.ravel(), pk, vk, bg, ba]) for Rk, pk, vk in states])
    init = true.copy()
    for k in range(nKF):
        if kind[k] == 0:
            Rk = states[k][0] @ _rot_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(0.5))
            init[k, :9] = Rk.ravel(); init[k, 9:12] += rng.normal(0, 0.02, 3); init[k, 12:15] += rng.normal(0, 0.03, 3)
            init[k, 15:18] += rng.normal(0, 0.001, 3); init[k, 18:21] += rng.normal(0, 0.005, 3)
    iKF1 = np.arange(0, n_opt, dtype=np.int32); iKF2 = np.arange(1, n_opt + 1, dtype=np.int32)
    return dict(kfState=init.astype(np.float32), kfKind=kind, mpPos=(X + rng.normal(0, 0.03, X.shape)).astype(np.float32),
                mpClose=(depth_ref < 10).astype(np.uint8), eKF=np.array(eKF, np.int32), eMP=np.array(eMP, np.int32),
                eObs=np.array(eObs, np.float32), eInvSigma2=np.array(eInv, np.float32), iKF1=iKF1, iKF2=iKF2,
                iRobust=(iKF1 == 0).astype(np.uint8), iInfoScale=np.where(iKF1 == 0, 1e-2, 1.0).astype(np.float32),
                imuStart=np.array(start, np.int32), acc=np.array(acc, np.float32), gyro=np.array(gyro, np.float32),
                dt=np.array(dts, np.float32), bias=np.concatenate([ba, bg]).astype(np.float32),
                Tbc12=np.concatenate([Tbc[:3, :3].ravel(), Tbc[:3, 3]]).astype(np.float32), cam=cam, true=true, truePts=X,
                eRight=np.array(eRight, np.uint8), rig28=tumvi_rig28() if rig else None)
