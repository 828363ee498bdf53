"""Bundle-adjustment entry points backed by the HIP residual/Jacobian kernel.

Drop-in for the reference's slam/core/ba_utils.py: same function names,
arguments, defaults, in-place mutation and logging behaviour

    two_view_ba              (ba_utils.py:74)
    pose_only_ba             (ba_utils.py:89)
    local_bundle_adjustment  (ba_utils.py:146)   <- the hot path
    global_bundle_adjustment (ba_utils.py:170)
    _core_ba                 (ba_utils.py:220)

Poses are T_cw (camera-from-world) 4x4 matrices in `kfs[k].pose`; pixels come
from `kfs[k].kps[i].pt`; `world_map.points` is a dict of objects with
`.position` (float64[3], optimised IN PLACE) and `.observations`
[(kf_idx, kp_idx, descriptor)].

Instead of one pybind residual block per observation, the map is snapshotted
once into SoA index arrays (the two Python loops of ba_utils.py:262-282 with
set lookups instead of list scans), the residual/Jacobian arithmetic of
`ReprojErrorCost(PINHOLE)` runs batched on the GPU, and
`opencv-simpleslam_amd/ba_solver.py` plays the role of `pyceres.solve`.
"""
from __future__ import annotations

import logging

import numpy as np

from .pose_utils import _pose_inverse, _pose_to_quat_trans, _quat_trans_to_pose  # noqa: F401
from ... import ba_solver

logger = logging.getLogger("ba")

HUBER_DELTA = 2.0      # pyceres.HuberLoss(2.0), ba_utils.py:236
MIN_RESIDUALS = 10     # ba_utils.py:284


def two_view_ba(world_map, K, kfs, max_iters: int = 20):
    assert len(world_map.poses) >= 2, "two_view_ba expects at least 2 poses"
    _core_ba(world_map, K, kfs, opt_kf_idx=[0, 1], fix_kf_idx=[], max_iters=max_iters,
             info_tag="[2-view BA]")


def pose_only_ba(world_map, K, kfs, kf_idx: int, max_iters: int = 8, huber_thr: float = 2.0):
    """One keyframe pose free, every landmark constant (ba_utils.py:89-140)."""
    intr = _intrinsics(K)
    q0, t0 = _pose_to_quat_trans(kfs[kf_idx].pose)
    pts, obs_point, obs_uv = [], [], []
    for mp in world_map.points.values():
        hits = [(f, i) for f, i, _ in mp.observations if f == kf_idx]
        if not hits:
            continue
        pts.append(np.asarray(mp.position, np.float64))
        for f, i in hits:
            u, v = kfs[f].kps[i].pt
            obs_point.append(len(pts) - 1)
            obs_uv.append((float(u), float(v)))
    if len(obs_point) < MIN_RESIDUALS:
        logger.warning("[Pose-only BA] skipped – not enough residuals")
        return
    prob = ba_solver.BAProblem(
        q=q0[None].copy(), t=t0[None].copy(), pose_const=np.zeros(1, bool),
        X=np.array(pts, np.float64), intr=intr,
        obs_pose=np.zeros(len(obs_point), np.int32),
        obs_point=np.asarray(obs_point, np.int32),
        obs_uv=np.asarray(obs_uv, np.float64))
    summ = ba_solver.solve_pose_only(prob, max_iters, huber_thr)
    new_Tcw = _quat_trans_to_pose(prob.q[0], prob.t[0])
    kfs[kf_idx].pose = new_Tcw
    if len(world_map.poses) > kf_idx:
        world_map.poses[kf_idx][:] = new_Tcw
    logger.debug("[Pose-only BA] iters=%s residuals=%d", summ.successful_steps, len(obs_point))


def local_bundle_adjustment(world_map, K, kfs, center_kf_idx: int, window_size: int = 6,
                            max_points: int = 10000, max_iters: int = 15):
    """Sliding-window BA: keyframes [max(1, c-w+1) .. c] are optimised, every
    earlier keyframe is held fixed as gauge (ba_utils.py:155-157)."""
    first_opt = max(1, center_kf_idx - window_size + 1)
    opt_kf = list(range(first_opt, center_kf_idx + 1))
    fix_kf = list(range(0, first_opt))
    logger.debug("[Local BA window] | opt_kf=%s fix_kf=%s center=%d", opt_kf, fix_kf, center_kf_idx)
    _core_ba(world_map, K, kfs, opt_kf_idx=opt_kf, fix_kf_idx=fix_kf, max_points=max_points,
             max_iters=max_iters, info_tag=f"[Local BA @ KF {center_kf_idx}]")


def global_bundle_adjustment(world_map, K, kfs, *, fix_first: bool = True,
                             max_points: int | None = 30000, max_iters: int = 30):
    if len(kfs) < 2:
        logger.warning("[Global BA] skipped – need at least 2 keyframes")
        return
    opt_kf_idx = list(range(len(kfs)))
    fix_kf_idx = [0] if (fix_first and len(kfs) > 0) else []
    _core_ba(world_map, K, kfs, opt_kf_idx=opt_kf_idx, fix_kf_idx=fix_kf_idx,
             max_points=max_points, max_iters=max_iters,
             info_tag=f"[Global BA] | KFs={len(kfs)} pts≤{max_points}]")


def _intrinsics(K):
    return np.array([float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])], np.float64)


def snapshot_problem(world_map, K, kfs, opt_kf_idx, fix_kf_idx, max_points=None):
    """Map -> SoA, following `_core_ba`'s selection rules exactly:
    a point enters iff one of its observations is in an optimisable keyframe
    (ba_utils.py:264) and fewer than max_points were taken before it (:266, the
    loop keeps scanning); of its observations those in opt or fixed keyframes
    become residuals (:273), in observation order.  A keyframe listed in both
    sets ends up constant, as Ceres' set_parameter_block_constant would make it
    (:250-257).  Returns (BAProblem, kf_rows, point_objs)."""
    opt, fix = list(opt_kf_idx), list(fix_kf_idx)
    opt_set, fix_set = set(opt), set(fix)
    rows = {}
    q, t, const = [], [], []
    for k in opt + [f for f in fix if f not in opt_set]:
        qq, tt = _pose_to_quat_trans(kfs[k].pose)
        rows[k] = len(q)
        q.append(qq); t.append(tt); const.append(k in fix_set)

    point_objs, obs_pose, obs_point, obs_uv = [], [], [], []
    for mp in world_map.points.values():
        observations = mp.observations
        if not any(f in opt_set for f, _, _ in observations):
            continue
        if max_points and len(point_objs) >= max_points:
            continue
        point_objs.append(mp)
        j = len(point_objs) - 1
        for f_idx, kp_idx, _d in observations:
            if f_idx not in opt_set and f_idx not in fix_set:
                continue
            u, v = kfs[f_idx].kps[kp_idx].pt
            obs_pose.append(rows[f_idx]); obs_point.append(j); obs_uv.append((float(u), float(v)))

    prob = ba_solver.BAProblem(
        q=np.array(q, np.float64).reshape(-1, 4), t=np.array(t, np.float64).reshape(-1, 3),
        pose_const=np.array(const, bool),
        X=np.array([np.asarray(mp.position, np.float64) for mp in point_objs]).reshape(-1, 3),
        intr=_intrinsics(K),
        obs_pose=np.asarray(obs_pose, np.int32), obs_point=np.asarray(obs_point, np.int32),
        obs_uv=np.asarray(obs_uv, np.float64).reshape(-1, 2))
    return prob, rows, point_objs


def _core_ba(world_map, K, kfs, *, opt_kf_idx, fix_kf_idx, max_points=None, max_iters: int = 20,
             info_tag: str = ""):
    prob, rows, point_objs = snapshot_problem(world_map, K, kfs, opt_kf_idx, fix_kf_idx, max_points)
    n_res = len(prob.obs_pose)
    if n_res < MIN_RESIDUALS:
        logger.warning("%s skipped – not enough residuals", info_tag)
        return

    summ = ba_solver.solve(prob, max_iters, HUBER_DELTA)

    # landmarks: in place, identity-preserving (the reference hands mp.position
    # itself to Ceres, ba_utils.py:269)
    for mp, Xn in zip(point_objs, prob.X):
        mp.position[:] = Xn
    # poses: new matrix on the keyframe, overwrite slot k of the map trajectory
    # (ba_utils.py:296-300 - indexed by KEYFRAME index, as the reference does)
    for k in opt_kf_idx:
        new_Tcw = _quat_trans_to_pose(prob.q[rows[k]], prob.t[rows[k]])
        kfs[k].pose = new_Tcw
        if len(world_map.poses) > k:
            world_map.poses[k][:] = new_Tcw

    logger.info("%s iters=%s  chi2=%.2f  residuals=%d", info_tag, str(summ.iterations),
                float(summ.final_cost), n_res)
    return None
