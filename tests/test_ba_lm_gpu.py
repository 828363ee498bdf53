"""GPU parity of the device-resident LM (csrc/ba_lm.hip, `sslam_ba_solve_host`) and of the host
Schur loop used for global BA (ba_solver.solve_host): both against the single-Jacobian oracle
(oracle/ba_ref.solve_dense_lm; its scipy.sparse form at the C3 size and for > 12 free poses),
plus determinism and edge cases.
fp64 throughout; tolerance 1e-8 relative on parameters (summation order differs), identical
iteration / step counts and termination reason."""
import copy

import numpy as np
import pytest

import ba_scenes
from conftest import load_pkg
from oracle import ba_ref

pytestmark = pytest.mark.gpu


def _snapshot(n_frames=10, window=8, noise=True, n_points=50):
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(n_frames, n_points=n_points, add_noise=noise)
    c = n_frames - 1
    first = max(1, c - window + 1)
    prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(first, c + 1)), list(range(0, first)))
    return prob


def _clone(p):
    return copy.deepcopy(p)


def _close(a, b, tol=1e-8):
    np.testing.assert_allclose(a.q, b.q, rtol=tol, atol=tol)
    np.testing.assert_allclose(a.t, b.t, rtol=tol, atol=tol)
    np.testing.assert_allclose(a.X, b.X, rtol=tol, atol=tol)


@pytest.mark.parametrize("points_const", [False, True])
def test_device_lm_matches_dense_oracle(points_const):
    S = load_pkg("ba_solver")
    prob = _snapshot()
    dev = _clone(prob)
    sd = S.solve_device(dev, 25, 2.0, points_const=points_const)
    q, t, X, info = ba_ref.solve_dense_lm(prob.q, prob.t, prob.pose_const, prob.X, prob.intr, prob.obs_pose,
                                          prob.obs_point, prob.obs_uv, 25, 2.0, points_const)
    assert sd.iterations == info["iterations"] and sd.successful_steps == info["successful_steps"]
    assert sd.termination == info["termination"]
    np.testing.assert_allclose(sd.initial_cost, info["initial_cost"], rtol=1e-12)
    np.testing.assert_allclose(sd.final_cost, info["final_cost"], rtol=1e-9)
    assert sd.final_cost < 0.5 * sd.initial_cost
    np.testing.assert_allclose(dev.q, q, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(dev.t, t, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(dev.X, X, rtol=1e-7, atol=1e-7)
    # constant blocks are bit-for-bit untouched
    np.testing.assert_array_equal(dev.q[prob.pose_const], prob.q[prob.pose_const])
    np.testing.assert_array_equal(dev.t[prob.pose_const], prob.t[prob.pose_const])
    if points_const:
        np.testing.assert_array_equal(dev.X, prob.X)


def _oracle(prob, iters, sparse=True):
    return ba_ref.solve_dense_lm(prob.q, prob.t, prob.pose_const, prob.X, prob.intr, prob.obs_pose,
                                 prob.obs_point, prob.obs_uv, iters, 2.0, False, sparse=sparse)


def test_device_lm_matches_oracle_at_c3_size_and_is_deterministic():
    """SURVEY 8(d) C3 scene (10 opt + 5 fixed KFs, 5000 points, ~30 k observations): the device
    LM against the single-Jacobian oracle (no Schur, SuperLU on the full normal equations), and
    the host Schur loop against the same oracle."""
    S = load_pkg("ba_solver")
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.scaled_scene()
    prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(5, 15)), list(range(0, 5)), 5000)
    assert len(prob.obs_pose) > 20000
    host, dev, dev2 = _clone(prob), _clone(prob), _clone(prob)
    q, t, X, info = _oracle(prob, 12)
    sd = S.solve_device(dev, 12, 2.0)
    sd2 = S.solve_device(dev2, 12, 2.0)
    sh = S.solve_host(host, 12, 2.0)
    for summ, got in ((sd, dev), (sh, host)):
        assert (summ.iterations, summ.successful_steps, summ.termination) == (
            info["iterations"], info["successful_steps"], info["termination"])
        np.testing.assert_allclose(summ.initial_cost, info["initial_cost"], rtol=1e-12)
        np.testing.assert_allclose(summ.final_cost, info["final_cost"], rtol=1e-8)
        np.testing.assert_allclose(got.q, q, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(got.t, t, rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(got.X, X, rtol=1e-6, atol=1e-6)
    assert sd.final_cost < 0.2 * sd.initial_cost
    # order-fixed reductions: bit-identical from run to run
    assert sd2.final_cost == sd.final_cost
    np.testing.assert_array_equal(dev.q, dev2.q)
    np.testing.assert_array_equal(dev.X, dev2.X)


def test_global_ba_shape_on_the_device_and_in_the_host_loop_match_the_oracle(monkeypatch):
    """Global-BA shape: 29 free poses, one gauge keyframe - more than the 12 whose reduced system fits LDS, so the
    device LM factors it in device memory (lm_solve_big_kernel); the host loop (SSLAM_BA_SOLVER=host) accumulates
    the Schur complement per pair of observations of a landmark; the oracle never forms a Schur complement.  All
    three must walk the same trust-region trajectory."""
    S = load_pkg("ba_solver")
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.scaled_scene(n_kf=30, n_points=1500)
    prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(30)), [0], 30000)
    assert 12 < int(np.count_nonzero(~prob.pose_const)) == 29 <= S.MAX_DEVICE_POSES
    q, t, X, info = _oracle(prob, 10)
    for mode in ("auto", "host"):
        monkeypatch.setenv("SSLAM_BA_SOLVER", mode)
        got = _clone(prob)
        summ = S.solve(got, 10, 2.0)
        assert (summ.iterations, summ.successful_steps, summ.termination) == (
            info["iterations"], info["successful_steps"], info["termination"]), mode
        np.testing.assert_allclose(summ.final_cost, info["final_cost"], rtol=1e-8)
        assert summ.final_cost < 0.2 * summ.initial_cost
        np.testing.assert_allclose(got.q, q, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(got.t, t, rtol=1e-7, atol=1e-7)
        # a landmark seen twice under a small baseline is weakly determined along its ray
        np.testing.assert_allclose(got.X, X, rtol=1e-4, atol=1e-3)
    # the device path is bit-reproducible here too
    monkeypatch.setenv("SSLAM_BA_SOLVER", "device")
    a, b = _clone(prob), _clone(prob)
    S.solve(a, 10, 2.0); S.solve(b, 10, 2.0)
    np.testing.assert_array_equal(a.q, b.q)
    np.testing.assert_array_equal(a.X, b.X)


def test_global_bundle_adjustment_through_the_driver_name():
    """`global_bundle_adjustment` as main_revamped.py:81 imports it (reference ba_utils.py:170-214):
    >= 13 keyframes, KF 0 fixed, in-place mutation contract, RMSE strictly decreases (the
    property the reference's own BA test pins, tests/test_ba_utils_T_c_w.py:264-314)."""
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.scaled_scene(n_kf=16, n_points=800)
    pos_ids = {pid: id(mp.position) for pid, mp in wmap.points.items()}
    pose0 = kfs[0].pose.copy()
    old_pose_objs = [kf.pose for kf in kfs]
    before = ba_scenes.reproj_rmse(wmap, kfs, K)
    bau.global_bundle_adjustment(wmap, K, kfs, max_iters=15)
    after = ba_scenes.reproj_rmse(wmap, kfs, K)
    assert after < 0.5 * before and after < 1.6          # 1 px pixel noise
    np.testing.assert_array_equal(kfs[0].pose, pose0)    # gauge keyframe untouched (fix_first)
    for pid, mp in wmap.points.items():                  # landmarks mutated in place
        assert id(mp.position) == pos_ids[pid]
    for k in range(1, len(kfs)):                         # poses replaced + trajectory slot overwritten
        assert kfs[k].pose is not old_pose_objs[k]
        np.testing.assert_array_equal(wmap.poses[k], kfs[k].pose)
    # fix_first=False leaves the gauge free and must still run (Ceres would, too)
    wmap2, kfs2, _ = ba_scenes.scaled_scene(n_kf=14, n_points=400)
    b2 = ba_scenes.reproj_rmse(wmap2, kfs2, K)
    bau.global_bundle_adjustment(wmap2, K, kfs2, fix_first=False, max_iters=5)
    assert ba_scenes.reproj_rmse(wmap2, kfs2, K) < b2
    # fewer than two keyframes: warning + return
    bau.global_bundle_adjustment(wmap2, K, kfs2[:1])


def test_device_lm_perfect_scene_stops_at_once_and_zero_iters():
    S = load_pkg("ba_solver")
    prob = _snapshot(noise=False)
    dev = _clone(prob)
    sd = S.solve_device(dev, 10, 2.0)
    assert sd.final_cost <= sd.initial_cost <= 1e-12 * len(prob.obs_pose) + 1e-9
    _close(dev, prob, 1e-6)
    dev = _clone(_snapshot())
    before = _clone(dev)
    sd = S.solve_device(dev, 0, 2.0)
    assert sd.iterations == 0 and sd.termination == "max iterations" and sd.final_cost == sd.initial_cost
    np.testing.assert_array_equal(dev.X, before.X)


def test_device_lm_rejects_oversized_window_and_bad_index(native):
    S = load_pkg("ba_solver")
    prob = _snapshot(n_frames=10, window=8)
    big = _clone(prob)
    big.pose_const = np.zeros(len(big.q), bool)
    reps = 30                                                # 300 free poses > MAX_DEVICE_POSES
    big.q = np.tile(big.q, (reps, 1)); big.t = np.tile(big.t, (reps, 1)); big.pose_const = np.zeros(len(big.q), bool)
    with pytest.raises(native.NativeError, match="optimised poses"):
        S.solve_device(big, 5, 2.0)
    bad = _clone(prob)
    bad.obs_point = bad.obs_point.copy(); bad.obs_point[3] = len(bad.X)
    with pytest.raises(native.NativeError, match="out of range"):
        S.solve_device(bad, 5, 2.0)


def test_outliers_are_down_weighted_like_the_oracle():
    """Gross outliers (50 px) on 10 % of the observations: Huber keeps the fixed point of the
    dense oracle."""
    S = load_pkg("ba_solver")
    prob = _snapshot()
    rng = np.random.default_rng(7)
    bad = rng.choice(len(prob.obs_uv), len(prob.obs_uv) // 10, replace=False)
    prob.obs_uv[bad] += rng.normal(0, 50.0, (len(bad), 2))
    dev = _clone(prob)
    sd = S.solve_device(dev, 30, 2.0)
    q, t, X, info = ba_ref.solve_dense_lm(prob.q, prob.t, prob.pose_const, prob.X, prob.intr, prob.obs_pose,
                                          prob.obs_point, prob.obs_uv, 30, 2.0)
    np.testing.assert_allclose(sd.final_cost, info["final_cost"], rtol=1e-8)
    assert sd.successful_steps == info["successful_steps"]
    np.testing.assert_allclose(dev.t, t, rtol=1e-6, atol=1e-7)


def test_ill_conditioned_windows_terminate_finite_and_never_increase_the_cost():
    """Where Ceres would fall back inside its linear solver, this LM rejects a step whose reduced
    system fails the Cholesky factorisation and shrinks the trust region.  What a
    caller must be able to rely on either way: the solve terminates, parameters stay finite, the cost
    never goes up.  (a) no constant pose at all - the 7-dimensional gauge freedom makes the reduced
    system singular up to the damping; (b) a free pose that no residual observes - a zero 6 x 6
    diagonal block; (c) a landmark seen once - rank-deficient 3 x 3 point block."""
    S = load_pkg("ba_solver")
    base = _snapshot(n_frames=8, window=6)
    # (a) gauge-free
    a = _clone(base); a.pose_const = np.zeros(len(a.q), bool)
    # (b) drop every observation of one free pose
    b = _clone(base)
    victim = int(np.flatnonzero(~b.pose_const)[0])
    keep = b.obs_pose != victim
    b.obs_pose, b.obs_point, b.obs_uv = b.obs_pose[keep].copy(), b.obs_point[keep].copy(), b.obs_uv[keep].copy()
    # (c) a point with a single observation
    c = _clone(base)
    lone = int(c.obs_point[0])
    first = np.flatnonzero(c.obs_point == lone)[1:]
    keep = np.ones(len(c.obs_point), bool); keep[first] = False
    c.obs_pose, c.obs_point, c.obs_uv = c.obs_pose[keep].copy(), c.obs_point[keep].copy(), c.obs_uv[keep].copy()
    for name, prob in (("gauge-free", a), ("unobserved pose", b), ("single-view point", c)):
        before = _clone(prob)
        sd = S.solve_device(prob, 20, 2.0)
        assert sd.iterations <= 20, name
        assert np.isfinite(sd.final_cost) and sd.final_cost <= sd.initial_cost * (1 + 1e-12), (name, sd)
        for arr in (prob.q, prob.t, prob.X):
            assert np.isfinite(arr).all(), name
        np.testing.assert_allclose(np.linalg.norm(prob.q, axis=1), 1.0, atol=1e-9, err_msg=name)
        if name == "unobserved pose":                      # nothing pulls on it: it must not move
            np.testing.assert_allclose(prob.q[victim], before.q[victim], atol=1e-12)
            np.testing.assert_allclose(prob.t[victim], before.t[victim], atol=1e-12)
