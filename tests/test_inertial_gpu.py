"""SURVEY 8(f) row N1, first slice: IMU preintegration and PoseInertialOptimizationLastKeyFrame, HIP vs the CPU oracle
(oracle/inertial.cc; parity unpinned — see its header)."""
import numpy as np
import pytest

import oracle_lib as orc
from morb_slam_amd.synth import imu_calib_diagonals, make_inertial_problem

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def opt():
    from morb_slam_amd import Optimizer
    o = Optimizer(0)
    yield o
    o.close()


def _preintegrate_gpu(opt, probs):
    dev = torch.device("cuda", 0)
    nga, walk = imu_calib_diagonals()
    start = np.cumsum([0] + [len(p["dt"]) for p in probs]).astype(np.int32)
    cat = lambda k: torch.from_numpy(np.concatenate([p[k] for p in probs])).to(dev)
    bias = torch.from_numpy(np.stack([p["bias"] for p in probs])).to(dev)
    pre = opt.PreintegrateIMU(torch.from_numpy(start).to(dev), cat("acc"), cat("gyro"), cat("dt"), bias, nga, walk)
    torch.cuda.synchronize()
    return pre


def test_imu_preintegration_matches_oracle(opt):
    from morb_slam_amd.optimizer import PREINT_FIELDS
    probs = [make_inertial_problem(50, seed=s, n_imu=n) for s, n in enumerate([1, 5, 20, 40, 100, 7])]
    nga, walk = imu_calib_diagonals()
    pre = _preintegrate_gpu(opt, probs).cpu().numpy()
    for i, p in enumerate(probs):
        ref = orc.imu_preintegrate(p["bias"], nga, walk, p["acc"], p["gyro"], p["dt"])
        for name, (o, l) in PREINT_FIELDS.items():
            a, b = pre[i, o:o + l], ref[o:o + l]
            # float recursions in the same order; device sinf / cosf differ from glibc by ulps
            assert np.allclose(a, b, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(b).max())), (i, name, np.abs(a - b).max())
        R = pre[i, 1:10].reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
        assert abs(pre[i, 0] - p["dt"].sum()) < 1e-5


@pytest.mark.parametrize("n,n_imu,seeds", [(500, 20, range(6)), (60, 10, range(3)), (25, 40, range(3))])
def test_pose_inertial_optimization_last_keyframe(opt, n, n_imu, seeds):
    dev = torch.device("cuda", 0)
    probs = [make_inertial_problem(n, seed=s, n_imu=n_imu) for s in seeds]
    F = len(probs)
    nga, walk = imu_calib_diagonals()
    # the optimiser is compared on identical preintegrated inputs: the oracle's records go to both sides
    pre_o = np.stack([orc.imu_preintegrate(p["bias"], nga, walk, p["acc"], p["gyro"], p["dt"]) for p in probs])
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in probs])).to(dev)
    state = st("state0").clone()
    nin, outl, prior = opt.PoseInertialOptimizationLastKeyFrame(st("hasMP"), st("obs"), st("invSigma2"), st("Xw"), st("close"),
                                                                probs[0]["cam"], probs[0]["Tbc12"], st("kfState"),
                                                                torch.from_numpy(pre_o).to(dev), state)
    torch.cuda.synchronize()
    nin, outl, prior, state = nin.cpu().numpy(), outl.cpu().numpy(), prior.cpu().numpy(), state.cpu().numpy()
    for i, p in enumerate(probs):
        r, s_o, out_o, prior_o = orc.pose_inertial_optimization_last_keyframe(p, pre_o[i])
        # FP64 Gauss-Newton on both sides; tolerance 1e-4 on the state (north_star's FP bar)
        assert np.allclose(state[i], s_o, rtol=0, atol=1e-4), (i, np.abs(state[i] - s_o).max())
        diff = int((outl[i] != out_o).sum())
        assert diff <= 1, (i, diff)            # a chi2 within rounding of its threshold may flip
        assert abs(int(nin[i]) - r) <= 1
        assert np.allclose(prior[i][:21], prior_o[:21], atol=1e-6)
        if diff == 0:
            H, Ho = prior[i][21:].reshape(15, 15), prior_o[21:].reshape(15, 15)
            assert np.allclose(H, Ho, rtol=1e-5, atol=1e-6 * np.abs(Ho).max()), np.abs(H - Ho).max()
            assert np.allclose(H, H.T, rtol=1e-9, atol=1e-9 * np.abs(H).max())
        # sanity against the generating motion: the optimum is near the true state
        if n >= 60:
            R, Rt = state[i][:9].reshape(3, 3), p["true"][:9].reshape(3, 3)
            ang = np.degrees(np.arccos(np.clip((np.trace(R.T @ Rt) - 1) / 2, -1, 1)))
            assert ang < 0.5 and np.abs(state[i][9:12] - p["true"][9:12]).max() < 0.05, (ang, state[i][9:12] - p["true"][9:12])


def test_pose_inertial_uses_gpu_preintegration_end_to_end(opt):
    """PreintegrateIMU -> PoseInertialOptimizationLastKeyFrame on the device without a host round trip."""
    dev = torch.device("cuda", 0)
    probs = [make_inertial_problem(400, seed=20 + s, n_imu=20) for s in range(4)]
    pre = _preintegrate_gpu(opt, probs)
    st = lambda k: torch.from_numpy(np.stack([p[k] for p in probs])).to(dev)
    state = st("state0").clone()
    nin, outl, _ = opt.PoseInertialOptimizationLastKeyFrame(st("hasMP"), st("obs"), st("invSigma2"), st("Xw"), st("close"),
                                                            probs[0]["cam"], probs[0]["Tbc12"], st("kfState"), pre, state,
                                                            want_prior=False)
    torch.cuda.synchronize()
    nga, walk = imu_calib_diagonals()
    for i, p in enumerate(probs):
        pre_o = orc.imu_preintegrate(p["bias"], nga, walk, p["acc"], p["gyro"], p["dt"])
        r, s_o, out_o, _ = orc.pose_inertial_optimization_last_keyframe(p, pre_o)
        assert np.allclose(state[i].cpu().numpy(), s_o, atol=2e-4)
        assert abs(int(nin[i].item()) - r) <= 2
