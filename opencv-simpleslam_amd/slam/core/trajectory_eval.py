"""Trajectory evaluation: Sim(3) alignment of the estimated camera centres to ground truth and
ATE-RMSE (the second half of BASELINE.json's metric).

The reference holds the alignment in its 2-D trajectory viewer
(slam/core/visualization_utils.py:337-358: camera centre -R^T t, Umeyama's closed form on the
last `Kpairs` (gt, est) pairs) with the call disabled (:364) and never reduces it to a number;
`sim3_align` reproduces that closed form (pinned on vectors recorded from the reference's own
method, tests/golden/trajectory_alignment.npz) and `ate_rmse` is the usual RMSE of the aligned
centres.  Host numpy: a few hundred 3-vectors.
"""
from __future__ import annotations

import numpy as np


def cam_center_from_Tcw(Tcw) -> np.ndarray:
    """World-frame camera centre of a camera-from-world pose (visualization_utils.py:337-340)."""
    Tcw = np.asarray(Tcw, np.float64)
    return -Tcw[:3, :3].T @ Tcw[:3, 3]


def trajectory_centres(poses_cw) -> np.ndarray:
    """[n,3] centres of a list / array of 4x4 camera-from-world poses (SoA snapshot of Map.poses)."""
    P = np.asarray(poses_cw, np.float64).reshape(-1, 4, 4)
    return -np.einsum("nji,nj->ni", P[:, :3, :3], P[:, :3, 3])


def sim3_align(gt_xyz, est_xyz, Kpairs: int | None = 100):
    """(s, R, t) minimising sum |s R est + t - gt|^2 over the last `Kpairs` pairs (None = all);
    returns None with fewer than 6 pairs, as the viewer refuses to align then (:344-345)."""
    gt = np.asarray(gt_xyz, np.float64).reshape(-1, 3)
    est = np.asarray(est_xyz, np.float64).reshape(-1, 3)
    if len(gt) < 6 or len(est) < 6:
        return None
    if Kpairs is not None:
        gt, est = gt[-Kpairs:], est[-Kpairs:]
    n = len(gt)
    mu_g, mu_e = gt.mean(axis=0), est.mean(axis=0)
    g0, e0 = gt - mu_g, est - mu_e
    U, S, Vt = np.linalg.svd((e0.T @ g0) / n)          # covariance with the estimate on the left
    d = np.array([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = (U * d) @ Vt
    var_e = np.sum(e0 * e0) / n
    s = float(np.sum(S * d) / (var_e + 1e-12))
    # NOTE: with the covariance taken as est^T gt, U D V^T is the TRANSPOSE of the least-squares
    # rotation est -> gt, yet the viewer applies it as s * (R @ E.T) (:392).  The closed form is
    # reproduced here as the reference wrote it (that is what the golden vectors pin); `umeyama`
    # below is the conventional solution and is what `ate_rmse` uses.
    t = mu_g - s * (R @ mu_e)
    return s, R, t


def apply_sim3(est_xyz, s, R, t) -> np.ndarray:
    est = np.asarray(est_xyz, np.float64).reshape(-1, 3)
    return (s * (R @ est.T)).T + t


def umeyama(gt_xyz, est_xyz):
    """Least-squares Sim(3) (Umeyama 1991) with the covariance in the conventional order, so that
    s R est + t ~ gt for ANY rotation between the frames - what ATE needs."""
    gt = np.asarray(gt_xyz, np.float64).reshape(-1, 3)
    est = np.asarray(est_xyz, np.float64).reshape(-1, 3)
    n = len(gt)
    mu_g, mu_e = gt.mean(axis=0), est.mean(axis=0)
    g0, e0 = gt - mu_g, est - mu_e
    U, S, Vt = np.linalg.svd((g0.T @ e0) / n)
    d = np.array([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = (U * d) @ Vt
    s = float(np.sum(S * d) / (np.sum(e0 * e0) / n + 1e-12))
    return s, R, mu_g - s * (R @ mu_e)


def ate_rmse(gt_xyz, est_xyz, align: str = "sim3") -> float:
    """Absolute trajectory error: RMSE of |aligned est - gt| over all pairs.  align: 'sim3'
    (monocular: scale is unobservable), 'none'."""
    gt = np.asarray(gt_xyz, np.float64).reshape(-1, 3)
    est = np.asarray(est_xyz, np.float64).reshape(-1, 3)
    if len(gt) != len(est) or len(gt) == 0:
        raise ValueError("ate_rmse needs two equally long, non-empty trajectories")
    if align == "sim3":
        if len(gt) < 3:
            raise ValueError("Sim(3) alignment needs at least 3 poses")
        est = apply_sim3(est, *umeyama(gt, est))
    elif align != "none":
        raise ValueError(f"unknown alignment {align!r}")
    return float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1))))
