"""The inequality behind PoseOptimization's chi2 preview (morb_slam_amd/csrc/optimizer.hip, k_pose_opt2): two floating-point sums of the same n
non-negative terms differ by less than 2 n 2^-53 of their value, whatever their order and association (each is within gamma_{n-1} of the exact
sum: Higham, Accuracy and Stability of Numerical Algorithms, section 4.2).  The kernel declares a trial "certainly rejected" when a TREE sum of
its robustified chi2 terms, shrunk by 8 n 2^-53, still exceeds the current state's chi2 — which then also holds for g2o's edge-order sum.
Checked here on adversarial inputs: the bound itself, and that the rule never fires when the edge-order sum would NOT exceed the threshold."""
from fractions import Fraction

import numpy as np
import pytest

U = 2.0 ** -53


def _orders(x, rng):
    n = len(x)
    seq = 0.0
    for v in x:                      # g2o's order: one addition after the other
        seq += v
    rev = 0.0
    for v in x[::-1]:
        rev += v
    y = x.copy()                     # a pairwise tree (what a wave reduction + a fixed cross-wave sum amount to)
    while len(y) > 1:
        if len(y) & 1:
            y = np.append(y, 0.0)
        y = y[0::2] + y[1::2]
    tree = float(y[0])
    p = rng.permutation(n)
    perm = 0.0
    for v in x[p]:
        perm += v
    blocks = float(np.sum([float(np.sum(x[i:i + 64])) for i in range(0, n, 64)]))   # numpy's own blocked pairwise sums
    return seq, (rev, tree, perm, blocks)


@pytest.mark.parametrize("n", [3, 17, 64, 540, 860, 1664, 8192])
def test_sums_of_non_negative_terms_in_any_order_agree_to_2n_ulps(n):
    rng = np.random.default_rng(1000 + n)
    for spread in (0, 8, 30, 60):    # magnitudes over 2^spread: chi2 terms of inliers next to a few outliers
        for _ in range(20):
            x = np.ldexp(rng.random(n) + 0.5, rng.integers(-spread // 2, spread // 2 + 1, n))
            x[rng.random(n) < 0.1] = 0.0
            seq, others = _orders(x, rng)
            exact = float(sum(Fraction(v) for v in x))
            assert abs(seq - exact) <= (n - 1) * U * exact * 1.0000001
            for s in others:
                assert abs(s - seq) <= 2 * n * U * max(s, seq)
                # the kernel's rule with the threshold placed anywhere between the two sums never contradicts the edge-order decision
                for thr in (seq, np.nextafter(seq, np.inf), np.nextafter(seq, 0.0), 0.5 * (s + seq)):
                    certainly_rejected = s * (1.0 - 8.0 * n * U) > thr
                    if certainly_rejected:
                        assert seq > thr


def test_the_rule_is_not_vacuous():
    """... and it does fire as soon as the margin is a few n ulps (the oracle's traces: 90 % of the rejected trials have margins above 1e-12)."""
    rng = np.random.default_rng(7)
    n = 540
    x = rng.random(n) * 5.0
    seq, (rev, tree, perm, blocks) = _orders(x, rng)
    assert tree * (1.0 - 8.0 * n * U) > seq * (1.0 - 1e-12)
    assert not (tree * (1.0 - 8.0 * n * U) > seq)
