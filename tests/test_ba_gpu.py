"""GPU parity tests for the local-BA path, through the C-ABI.

Bar: fp64 residuals/Jacobians equal to the numpy oracle within 1e-11 relative
(the only differences are fma contraction and operation order), plus the
reference's own BA test properties (tests/test_ba_utils_T_c_w.py:264-314:
reprojection RMSE must not increase on a perfect scene and must strictly
decrease on the noisy one)."""
import copy

import numpy as np
import pytest

import ba_scenes
from conftest import load_pkg
from oracle import ba_ref

pytestmark = pytest.mark.gpu


def _problem(n, P=15, Q=5000, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.standard_normal((P, 3))
    X = rng.standard_normal((Q, 3)) * 3 + np.array([0, 0, 20.0])
    pi = rng.integers(0, P, n).astype(np.int32)
    xi = rng.integers(0, Q, n).astype(np.int32)
    uv = rng.uniform(0, 1241, (n, 2))
    intr = np.array([718.856, 718.856, 607.1928, 185.2157])
    return pi, xi, uv, q, t, X, intr


def _run_host(native, ctx, pi, xi, uv, q, t, X, intr, jac=True):
    n = len(pi)
    r = np.full((n, 2), np.nan); Jq = np.full((n, 2, 4), np.nan)
    Jt = np.full((n, 2, 3), np.nan); JX = np.full((n, 2, 3), np.nan)
    P = native.ptr
    native.check(native.lib().sslam_ba_residual_jacobian_host(
        ctx.handle, n, P(pi), P(xi), P(uv), len(q), P(q), P(t), len(X), P(X), P(intr), P(r),
        P(Jq) if jac else None, P(Jt) if jac else None, P(JX) if jac else None))
    return r, Jq, Jt, JX


@pytest.mark.parametrize("n", [1, 255, 256, 257, 30011])
def test_kernel_matches_oracle(native, gpu_ctx, n):
    args = _problem(n, seed=n)
    r, Jq, Jt, JX = _run_host(native, gpu_ctx, *args)
    ro, Jqo, Jto, JXo = ba_ref.reproj_residual_jacobian(*args)
    for a, b in ((r, ro), (Jq, Jqo), (Jt, Jto), (JX, JXo)):
        scale = np.abs(b).max() + 1e-300
        assert np.abs(a - b).max() / scale < 1e-11


def test_residual_only_and_empty(native, gpu_ctx):
    args = _problem(1000, seed=3)
    r, Jq, _, _ = _run_host(native, gpu_ctx, *args, jac=False)
    assert np.isnan(Jq).all()
    np.testing.assert_allclose(r, ba_ref.reproj_residual_jacobian(*args)[0], rtol=1e-11, atol=1e-9)
    pi, xi, uv, q, t, X, intr = args
    _run_host(native, gpu_ctx, pi[:0], xi[:0], uv[:0], q, t, X, intr)      # n_obs == 0 is legal


def test_bad_index_is_rejected_on_host(native, gpu_ctx):
    pi, xi, uv, q, t, X, intr = _problem(10, seed=4)
    pi[3] = len(q)                     # out of range: must never reach the GPU
    with pytest.raises(native.NativeError, match="out of range"):
        _run_host(native, gpu_ctx, pi, xi, uv, q, t, X, intr)


def test_device_pointer_entry_at_scale(native, gpu_ctx):
    """C3-sized and 64x larger problem through the _dev entry; checks against the
    oracle on a strided sample plus a size-independent property: r is linear in uv."""
    ctx, L, P = gpu_ctx, native.lib(), native.ptr
    n = 2_000_000
    pi, xi, uv, q, t, X, intr = _problem(n, seed=5)
    d = {k: ctx.upload(v) for k, v in dict(pi=pi, xi=xi, uv=uv, q=q, t=t, X=X, intr=intr).items()}
    outs = {k: ctx.malloc(n * w * 8) for k, w in dict(r=2, Jq=8, Jt=6, JX=6).items()}
    native.check(L.sslam_ba_residual_jacobian_dev(
        ctx.handle, n, P(d["pi"]), P(d["xi"]), P(d["uv"]), len(q), P(d["q"]), P(d["t"]), len(X),
        P(d["X"]), P(d["intr"]), P(outs["r"]), P(outs["Jq"]), P(outs["Jt"]), P(outs["JX"])))
    ctx.sync()
    r = np.empty((n, 2)); Jq = np.empty((n, 2, 4)); JX = np.empty((n, 2, 3))
    ctx.d2h(r, outs["r"]); ctx.d2h(Jq, outs["Jq"]); ctx.d2h(JX, outs["JX"])
    s = slice(0, n, 997)
    ro, Jqo, _, JXo = ba_ref.reproj_residual_jacobian(pi[s], xi[s], uv[s], q, t, X, intr)
    np.testing.assert_allclose(r[s], ro, rtol=1e-11, atol=1e-8)
    np.testing.assert_allclose(Jq[s], Jqo, rtol=1e-10, atol=1e-7)
    np.testing.assert_allclose(JX[s], JXo, rtol=1e-10, atol=1e-7)
    # property: shifting every observation by (du,dv) shifts every residual by -(du,dv)
    uv2 = uv + np.array([3.0, -7.0]); ctx.h2d(d["uv"], uv2)
    native.check(L.sslam_ba_residual_jacobian_dev(
        ctx.handle, n, P(d["pi"]), P(d["xi"]), P(d["uv"]), len(q), P(d["q"]), P(d["t"]), len(X),
        P(d["X"]), P(d["intr"]), P(outs["r"]), None, None, None))
    ctx.sync()
    r2 = np.empty((n, 2)); ctx.d2h(r2, outs["r"])
    np.testing.assert_allclose(r2 - r, np.broadcast_to([-3.0, 7.0], r.shape), atol=1e-8)
    for p in list(d.values()) + list(outs.values()):
        ctx.free(p)


# ----------------------------------------------------------------- drop-in API
def test_local_ba_perfect_scene_does_not_increase_rmse():
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(10, add_noise=False)
    before = ba_scenes.reproj_rmse(wmap, kfs, K)
    bau.local_bundle_adjustment(wmap, K, kfs, center_kf_idx=9, window_size=8, max_iters=25)
    assert ba_scenes.reproj_rmse(wmap, kfs, K) <= before + 1e-6


def test_local_ba_noisy_scene_reduces_rmse_and_mutates_like_reference():
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(10, add_noise=True)
    before = ba_scenes.reproj_rmse(wmap, kfs, K)
    pos_ids = {pid: id(mp.position) for pid, mp in wmap.points.items()}
    pose_ids = [id(k.pose) for k in kfs]
    map_pose_ids = [id(p) for p in wmap.poses]
    fixed_before = [kfs[k].pose.copy() for k in (0, 1)]
    bau.local_bundle_adjustment(wmap, K, kfs, center_kf_idx=9, window_size=8, max_iters=25)
    after = ba_scenes.reproj_rmse(wmap, kfs, K)
    assert after < before and after < 0.5 * before
    # landmarks optimised in place (identity preserved), ba_utils.py:269
    assert all(id(mp.position) == pos_ids[pid] for pid, mp in wmap.points.items())
    # opt keyframes (2..9) got NEW pose arrays; map trajectory slots overwritten in place
    assert all(id(kfs[k].pose) != pose_ids[k] for k in range(2, 10))
    assert [id(p) for p in wmap.poses] == map_pose_ids
    for k in range(2, 10):
        np.testing.assert_array_equal(wmap.poses[k], kfs[k].pose)
    # fixed keyframes (0,1) untouched
    for k in (0, 1):
        np.testing.assert_array_equal(kfs[k].pose, fixed_before[k])
        assert id(kfs[k].pose) == pose_ids[k]


def test_two_view_and_pose_only_ba():
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(2, add_noise=True)
    b = ba_scenes.reproj_rmse(wmap, kfs, K)
    bau.two_view_ba(wmap, K, kfs, max_iters=25)
    assert ba_scenes.reproj_rmse(wmap, kfs, K) < b
    wmap, kfs, K = ba_scenes.reference_test_scene(3, add_noise=True)
    b = ba_scenes.reproj_rmse(wmap, kfs, K, frames=[2])
    pts_before = {pid: mp.position.copy() for pid, mp in wmap.points.items()}
    bau.pose_only_ba(wmap, K, kfs, kf_idx=2, max_iters=15)
    assert ba_scenes.reproj_rmse(wmap, kfs, K, frames=[2]) < b
    for pid, mp in wmap.points.items():            # landmarks constant in pose-only BA
        np.testing.assert_array_equal(mp.position, pts_before[pid])


def test_too_few_residuals_is_a_noop(caplog):
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.reference_test_scene(3, n_points=2, add_noise=True)
    snap = copy.deepcopy([k.pose for k in kfs])
    assert bau.local_bundle_adjustment(wmap, K, kfs, center_kf_idx=2, window_size=2) is None
    for k, p in zip(kfs, snap):
        np.testing.assert_array_equal(k.pose, p)
    assert any("not enough residuals" in r.message for r in caplog.records)


def test_scaled_c3_scene_converges():
    """SURVEY 8(d) C3 scene: 10 opt + 5 fixed KFs, 5000 points, CLI defaults
    (window 10, max_points 5000, max_iters 12; main_revamped.py:242-248)."""
    bau = load_pkg("slam.core.ba_utils")
    wmap, kfs, K = ba_scenes.scaled_scene()
    before = ba_scenes.reproj_rmse(wmap, kfs, K)
    bau.local_bundle_adjustment(wmap, K, kfs, center_kf_idx=14, window_size=10,
                                max_points=5000, max_iters=12)
    after = ba_scenes.reproj_rmse(wmap, kfs, K)
    assert after < before and after < 2.0       # ~1 px pixel noise floor (sqrt(2) px RMSE)
