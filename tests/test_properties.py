"""Property tests (hypothesis) of the host-side logic: shard plan arithmetic, pose conversions,
the RANSAC iteration budget, Sim(3) alignment - no GPU."""
import numpy as np
from hypothesis import given, settings, strategies as st

from conftest import load_pkg
from oracle import ransac_ref


@settings(max_examples=200, deadline=None)
@given(world=st.integers(1, 16), B=st.integers(1, 64), frame=st.integers(0, 100000))
def test_shard_plan_owner_is_the_inverse_of_frames(world, B, frame):
    fs = load_pkg("frame_shard")
    rnd, rank, slot = fs.ShardPlan(world, 0, B).owner(frame)
    assert 0 <= rank < world and 0 <= slot < B
    plan = fs.ShardPlan(world, rank, B)
    assert plan.frames(rnd)[slot] == frame
    # chunks of one round are contiguous and disjoint across ranks
    starts = [fs.ShardPlan(world, r, B).frames(rnd)[0] for r in range(world)]
    assert starts == [starts[0] + r * B for r in range(world)]
    halo = plan.halo(rnd)
    assert halo is None if plan.frames(rnd)[0] == 0 else halo == plan.frames(rnd)[0] - 1


@settings(max_examples=200, deadline=None)
@given(v=st.lists(st.floats(-3.0, 3.0), min_size=3, max_size=3), t=st.lists(st.floats(-50, 50), min_size=3, max_size=3))
def test_pose_quaternion_roundtrip(v, t):
    P = load_pkg("slam.core.pose_utils")
    rv = np.array(v)
    ang = np.linalg.norm(rv)
    if ang < 1e-9:
        R = np.eye(3)
    else:
        k = rv / ang
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    q, tt = P._pose_to_quat_trans(T)
    assert abs(np.linalg.norm(q) - 1.0) < 1e-12 and q[3] >= 0.0            # unit, w >= 0 (pose_utils.py:95-105)
    np.testing.assert_allclose(P._quat_trans_to_pose(q, tt), T, atol=1e-9)
    np.testing.assert_allclose(P._pose_inverse(T) @ T, np.eye(4), atol=1e-9)


@settings(max_examples=300, deadline=None)
@given(p=st.floats(0.5, 0.9999), ep=st.floats(0.0, 1.0), cap=st.integers(1, 5000))
def test_ransac_budget_is_bounded_and_monotone(p, ep, cap):
    n = ransac_ref.update_num_iters(p, ep, 7, cap)
    assert 0 <= n <= cap
    assert ransac_ref.update_num_iters(p, min(1.0, ep + 0.05), 7, cap) >= n      # more outliers never need fewer samples


@settings(max_examples=100, deadline=None)
@given(seed=st.integers(0, 10000), s=st.floats(0.05, 50.0))
def test_umeyama_recovers_any_similarity(seed, s):
    T = load_pkg("slam.core.trajectory_eval")
    rng = np.random.default_rng(seed)
    est = rng.standard_normal((30, 3)) * 5
    U, _, Vt = np.linalg.svd(rng.standard_normal((3, 3)))
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = -R
    t = rng.standard_normal(3) * 20
    gt = (s * (R @ est.T)).T + t
    s2, R2, t2 = T.umeyama(gt, est)
    np.testing.assert_allclose(s2, s, rtol=1e-8)
    np.testing.assert_allclose(R2, R, atol=1e-8)
    assert T.ate_rmse(gt, est) < 1e-7 * (1 + s * 20)


@settings(max_examples=100, deadline=None)
@given(n=st.integers(1, 300), seed=st.integers(0, 10000), op=st.sampled_from(["setitem", "pt", "sort", "swap", "pop", "none"]))
def test_keypoint_coordinates_are_rebuilt_after_any_in_place_edit(n, seed, op):
    """The keypoint list handed back by feature_extractor is caller-owned; feature_matcher must see
    whatever is in it NOW (the reference rebuilds the tensor from .pt on every call,
    features_utils.py:65-77, :143-144)."""
    T = load_pkg("slam.core.types")
    rng = np.random.default_rng(seed)
    xy = (rng.random((n, 2)) * 1000).astype(np.float32)
    kps = T.keypoints_from_xy(xy)
    np.testing.assert_array_equal(T.xy_from_keypoints(kps), xy)
    i = int(rng.integers(0, n))
    if op == "setitem":
        kps[i] = T.KeyPoint(1.5, -2.5, 1)
    elif op == "pt" and not T.HAVE_CV2:
        kps[i].pt = (7.25, 8.5)
    elif op == "sort":
        kps.sort(key=lambda k: k.pt[1])
    elif op == "swap":
        j = int(rng.integers(0, n))
        kps[i], kps[j] = kps[j], kps[i]
    elif op == "pop":
        kps.pop(i)
    want = np.array([k.pt for k in kps], np.float32).reshape(-1, 2)
    got = T.xy_from_keypoints(kps)
    assert got.dtype == np.float32 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)
