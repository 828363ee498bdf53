"""Trajectory evaluation (host numpy): the viewer's Sim(3) closed form pinned on vectors recorded
from the REFERENCE's own method (tests/golden/make_alignment_golden.py), and ATE-RMSE properties."""
import numpy as np
import pytest

from conftest import ROOT, load_pkg

G = np.load(ROOT / "tests" / "golden" / "trajectory_alignment.npz")


@pytest.fixture(scope="module")
def T():
    return load_pkg("slam.core.trajectory_eval")


def test_alignment_matches_reference_vectors(T):
    for i in range(int(G["n_cases"])):
        s, R, t = T.sim3_align(G[f"gt{i}"], G[f"est{i}"], int(G["Kpairs"][i]))
        np.testing.assert_allclose(s, G["s"][i], rtol=1e-12)
        np.testing.assert_allclose(R, G["R"][i], atol=1e-12)
        np.testing.assert_allclose(t, G["t"][i], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(T.cam_center_from_Tcw(G["Tcw"]), G["centre"], atol=1e-14)
    np.testing.assert_allclose(T.trajectory_centres([G["Tcw"], np.eye(4)])[0], G["centre"], atol=1e-14)
    assert T.sim3_align(G["gt0"][:5], G["est0"][:5]) is None            # < 6 pairs: no alignment


def test_ate_is_zero_under_any_similarity_and_equals_the_noise_otherwise(T):
    rng = np.random.default_rng(3)
    gt = np.cumsum(rng.standard_normal((300, 3)), 0)
    A = rng.standard_normal((3, 3))
    U, _, Vt = np.linalg.svd(A)
    R = U @ Vt * np.sign(np.linalg.det(U @ Vt))
    est = ((gt - np.array([4.0, -2.0, 9.0])) @ R) / 7.3
    assert T.ate_rmse(gt, est) < 1e-9
    assert T.ate_rmse(gt, est, align="none") > 1.0
    s, Rr, t = T.umeyama(gt, est)
    np.testing.assert_allclose(s, 7.3, rtol=1e-9)
    np.testing.assert_allclose(np.linalg.det(Rr), 1.0, atol=1e-12)
    noisy = est + rng.normal(0, 0.01, est.shape)
    ate = T.ate_rmse(gt, noisy)
    assert 0.5 * 7.3 * 0.01 * np.sqrt(3) < ate < 1.2 * 7.3 * 0.01 * np.sqrt(3)
    # a mirrored estimate cannot be aligned by a proper rotation
    assert T.ate_rmse(gt, est * np.array([1.0, 1.0, -1.0])) > 0.1
    with pytest.raises(ValueError):
        T.ate_rmse(gt, est[:-1])
