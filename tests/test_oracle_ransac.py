"""CPU checks of oracle/ransac_ref.py (the restated cv2.findFundamentalMat): generator constants,
solver identities and the behaviour on synthetic two-view data."""
import numpy as np

import two_view
from oracle import ransac_ref as R


def test_cv_rng_is_the_documented_multiply_with_carry():
    rng = R.CvRNG()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert rng.next() == s & 0xFFFFFFFF
    vals = [R.CvRNG().uniform(0, 100) for _ in range(2)]
    assert vals[0] == vals[1] and 0 <= vals[0] < 100          # fixed seed: deterministic
    assert R.CvRNG().uniform(5, 5) == 5


def test_update_num_iters_matches_the_closed_form():
    assert R.update_num_iters(0.99, 0.45, 7, 1000) == int(np.rint(np.log(0.01) / np.log(1 - 0.55 ** 7)))
    assert R.update_num_iters(0.99, 0.0, 7, 1000) == 0
    assert R.update_num_iters(0.99, 1.0, 7, 1000) == 1000
    assert R.update_num_iters(0.99, 0.9, 7, 1000) == 1000       # needs more than the budget


def test_seven_point_models_satisfy_the_constraints():
    p1, p2, _ = two_view.make_matches(40, outlier_frac=0.0, noise=0.0, seed=3)
    Fs = R.run7point(p1[:7], p2[:7])
    assert 1 <= len(Fs) <= 3
    for F in Fs:
        assert abs(np.linalg.det(F)) < 1e-6 * np.abs(F).max() ** 3 + 1e-12
        assert F[2, 2] in (0.0, 1.0)
        assert R.compute_error(p1[:7], p2[:7], F).max() < 1e-3       # the sample lies on the model
    # one of the solutions explains every noise-free correspondence
    assert min(R.compute_error(p1, p2, F).max() for F in Fs) < 1e-2


def test_cubic_solver():
    for roots in ([1.0, -2.0, 3.5], [0.5, 0.5, -4.0]):
        c = np.poly(roots)
        got = sorted(R.solve_cubic(c))
        assert len(got) >= 2
        for g in got:
            assert min(abs(g - r) for r in roots) < 1e-6
    assert len(R.solve_cubic([1.0, 0.0, 1.0, 0.0])) == 1               # x (x^2 + 1)
    np.testing.assert_allclose(sorted(R.solve_cubic([0.0, 1.0, -3.0, 2.0])), [1.0, 2.0])


def test_ransac_recovers_the_inlier_set():
    p1, p2, truth = two_view.make_matches(400, outlier_frac=0.3, noise=0.3, seed=1)
    F, mask, info = R.find_fundamental_ransac(p1, p2, 1.0, 0.99)
    assert F is not None and not info["lmeds"]
    assert (mask & truth).sum() >= 0.8 * truth.sum()                    # most true matches kept (minimal model, no refit)
    assert (mask & ~truth).sum() <= 0.1 * (~truth).sum() + 3            # few outliers survive (near-epipolar ones)
    assert info["iterations"] < 1000                                    # the budget adapted


def test_lmeds_below_fifteen_points_and_degenerate_input():
    p1, p2, truth = two_view.make_matches(12, outlier_frac=0.17, noise=0.2, seed=5)
    F, mask, info = R.find_fundamental_ransac(p1, p2, 1.0, 0.99)
    assert info["lmeds"] and F is not None and mask.sum() >= 7
    # all points identical: no non-degenerate subset exists
    z = np.zeros((20, 2), np.float32)
    F, mask, info = R.find_fundamental_ransac(z, z, 1.0, 0.99)
    assert F is None and mask is None
