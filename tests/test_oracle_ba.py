"""CPU checks of oracle/ba_ref.py: analytic Jacobians vs central differences,
manifold, Huber, and problem assembly pinned on the trace recorded from the
REFERENCE's own `_core_ba` (tests/golden/ba_assembly.npz)."""
import types

import numpy as np

from conftest import ROOT, load_pkg
from oracle import ba_ref

G = np.load(ROOT / "tests" / "golden" / "ba_assembly.npz")


def _random_problem(n=200, P=6, Q=40, seed=0, unit=True):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((P, 4))
    if unit:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    else:
        q *= rng.uniform(0.5, 1.5, (P, 1))
    t = rng.standard_normal((P, 3))
    X = rng.standard_normal((Q, 3)) + np.array([0, 0, 12.0])
    pi = rng.integers(0, P, n).astype(np.int32)
    xi = rng.integers(0, Q, n).astype(np.int32)
    uv = rng.uniform(0, 1000, (n, 2))
    intr = np.array([718.856, 700.1, 607.19, 185.2])
    return pi, xi, uv, q, t, X, intr


def test_jacobians_match_central_differences():
    for unit in (True, False):          # autodiff differentiates the un-normalised formula
        pi, xi, uv, q, t, X, intr = _random_problem(unit=unit)
        r, Jq, Jt, JX = ba_ref.reproj_residual_jacobian(pi, xi, uv, q, t, X, intr)
        h = 1e-6
        for name, arr, J in (("q", q, Jq), ("t", t, Jt), ("X", X, JX)):
            idx = pi if name != "X" else xi
            for c in range(arr.shape[1]):
                ap, am = arr.copy(), arr.copy()
                ap[:, c] += h; am[:, c] -= h
                args_p = dict(q=q, t=t, X=X); args_m = dict(q=q, t=t, X=X)
                args_p[name] = ap; args_m[name] = am
                rp = ba_ref.reproj_residual_jacobian(pi, xi, uv, args_p["q"], args_p["t"], args_p["X"], intr)[0]
                rm = ba_ref.reproj_residual_jacobian(pi, xi, uv, args_m["q"], args_m["t"], args_m["X"], intr)[0]
                num = (rp - rm) / (2 * h)
                assert np.allclose(J[:, :, c], num, rtol=1e-5, atol=1e-4), (name, c, unit)
        _ = idx


def test_unit_quaternion_transform_is_rotation():
    pu = load_pkg("slam.core.pose_utils")
    rng = np.random.default_rng(1)
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    X = rng.standard_normal(3); t = rng.standard_normal(3)
    T = pu._quat_trans_to_pose(q, t)
    np.testing.assert_allclose(ba_ref.transform_point(q, t, X), T[:3, :3] @ X + t, atol=1e-12)


def test_plus_jacobian_matches_numeric_plus():
    rng = np.random.default_rng(2)
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    J = ba_ref.quat_plus_jacobian(q)
    h = 1e-7
    for c in range(3):
        d = np.zeros(3); d[c] = h
        num = (ba_ref.quat_plus(q, d) - ba_ref.quat_plus(q, -d)) / (2 * h)
        np.testing.assert_allclose(J[:, c], num, atol=1e-7)
    assert abs(np.linalg.norm(ba_ref.quat_plus(q, np.array([0.3, -0.2, 0.1]))) - 1) < 1e-12


def test_huber():
    rho, w = ba_ref.huber_rho(np.array([0.0, 1.0, 4.0, 9.0, 100.0]), 2.0)
    np.testing.assert_allclose(rho, [0, 1, 4, 2 * 2 * 3 - 4, 2 * 2 * 10 - 4])
    np.testing.assert_allclose(w, [1, 1, 1, 2 / 3, 0.2])


def _golden_scene():
    """Rebuild the recorded input scene as plain objects."""
    obs = G["obs_rows"]
    n_kf = len(G["in_poses"])
    kfs = [types.SimpleNamespace(pose=G["in_poses"][k].copy(), kps={}) for k in range(n_kf)]
    points = {}
    for pid, pos in zip(G["in_point_ids"], G["in_points"]):
        points[int(pid)] = types.SimpleNamespace(position=pos.copy(), observations=[])
    for pid, k, kp, u, v in obs:
        kfs[int(k)].kps[int(kp)] = types.SimpleNamespace(pt=(float(u), float(v)))
        points[int(pid)].observations.append((int(k), int(kp), None))
    wmap = types.SimpleNamespace(points=points, poses=[p.copy() for p in G["in_poses"]])
    return wmap, kfs


def test_oracle_assembly_matches_reference_trace():
    wmap, kfs = _golden_scene()
    opt, fix = ba_ref.local_window(int(G["center"]), int(G["window"]))
    asm = ba_ref.assemble_core_ba(
        ((pid, [(f, i) for f, i, _ in mp.observations]) for pid, mp in wmap.points.items()),
        lambda f, i: kfs[f].kps[i].pt, opt, fix, int(G["max_points"]))
    # residual order, owners and pixels exactly as the reference added them
    np.testing.assert_array_equal(asm["obs_kf"], G["res_kf"])
    np.testing.assert_array_equal(np.array(asm["point_keys"])[asm["obs_point"]], G["res_pid"])
    np.testing.assert_array_equal(asm["obs_uv"], G["res_uv"])
    # point block order = first-appearance order in the reference trace
    ref_pts = G["block_owner"][G["block_kind"] == 2]
    np.testing.assert_array_equal(asm["point_keys"], ref_pts)
    assert np.all(G["res_delta"] == 2.0)
    # which pose blocks were constant: fixed KFs (2 blocks each) + intrinsics
    n_fix = len(fix)
    assert int(G["block_const"].sum()) == 2 * n_fix + 1
    assert int((G["block_kind"] == 0).sum()) == len(opt) + n_fix


def test_product_snapshot_matches_reference_trace():
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs = _golden_scene()
    opt, fix = ba_ref.local_window(int(G["center"]), int(G["window"]))
    prob, rows, pts = bau.snapshot_problem(wmap, G["K"], kfs, opt, fix, int(G["max_points"]))
    inv_rows = {v: k for k, v in rows.items()}
    np.testing.assert_array_equal([inv_rows[r] for r in prob.obs_pose], G["res_kf"])
    np.testing.assert_array_equal(prob.obs_uv, G["res_uv"])
    ids = {id(mp): pid for pid, mp in wmap.points.items()}
    np.testing.assert_array_equal([ids[id(pts[j])] for j in prob.obs_point], G["res_pid"])
    assert [bool(prob.pose_const[rows[k]]) for k in opt] == [False] * len(opt)
    assert [bool(prob.pose_const[rows[k]]) for k in fix] == [True] * len(fix)
    np.testing.assert_allclose(prob.intr, [G["K"][0, 0], G["K"][1, 1], G["K"][0, 2], G["K"][1, 2]])
    assert int(G["rec_max_iters"]) == int(G["max_iters"])
    assert int(G["few_n_res"]) < 10 and int(G["few_solved"]) == 0


def _scene_problem(noise):
    import ba_scenes
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(8, n_points=40, add_noise=noise)
    prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(2, 8)), [0, 1])
    return prob


def _rmse(p, q, t, X):
    r = ba_ref.reproj_residual_jacobian(p.obs_pose, p.obs_point, p.obs_uv, q, t, X, p.intr)[0]
    return float(np.sqrt(np.mean(np.sum(r * r, axis=1))))


def test_dense_lm_has_the_reference_tests_property():
    """The one numeric property the reference's BA test pins (tests/test_ba_utils_T_c_w.py:264-314
    on the seeded scene of :116-218): RMSE does not increase on a perfect scene and strictly
    decreases on the noisy one; constant blocks stay bit-identical."""
    p = _scene_problem(False)
    q, t, X, info = ba_ref.solve_dense_lm(p.q, p.t, p.pose_const, p.X, p.intr, p.obs_pose, p.obs_point,
                                          p.obs_uv, 10)
    assert _rmse(p, q, t, X) <= _rmse(p, p.q, p.t, p.X) + 1e-9
    p = _scene_problem(True)
    q, t, X, info = ba_ref.solve_dense_lm(p.q, p.t, p.pose_const, p.X, p.intr, p.obs_pose, p.obs_point,
                                          p.obs_uv, 25)
    assert _rmse(p, q, t, X) < 0.5 * _rmse(p, p.q, p.t, p.X)
    assert info["successful_steps"] >= 3 and info["final_cost"] < info["initial_cost"]
    np.testing.assert_array_equal(q[p.pose_const], p.q[p.pose_const])
    np.testing.assert_array_equal(t[p.pose_const], p.t[p.pose_const])
    # unit quaternions stay unit under the manifold step
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)
    # pose-only variant leaves the landmarks alone
    q2, t2, X2, _ = ba_ref.solve_dense_lm(p.q, p.t, p.pose_const, p.X, p.intr, p.obs_pose, p.obs_point,
                                          p.obs_uv, 10, points_const=True)
    np.testing.assert_array_equal(X2, p.X)
    assert _rmse(p, q2, t2, X2) < _rmse(p, p.q, p.t, p.X)
