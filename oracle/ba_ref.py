"""CPU oracle: local-BA residual / Jacobian evaluation and problem assembly.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy, float64.

Restates, for the path `slam/core/ba_utils.py:146-165 -> :220-306 -> :56-68`
of the reference:

* `assemble_local_ba` - which keyframes / points / observations enter the
  problem and in which order (ba_utils.py:155-157 window rule, :243-257 pose
  blocks, :262-282 the two nested loops, `continue` quirk at :266-267).
  PINNED: tests/golden/ba_assembly.npz was produced by running the reference's
  own `_core_ba` against recording stubs.
* `reproj_residual_jacobian` - COLMAP 3.10
  `ReprojErrorCostFunction<PinholeCameraModel>` (src/colmap/estimators/
  cost_functions.h; sensor/models.h `PinholeCameraModel::ImgFromCam`) as built
  by `cost_functions.ReprojErrorCost(CameraModelId.PINHOLE, uv)` at
  ba_utils.py:61-64, differentiated the way `ceres::AutoDiffCostFunction<...,
  2, 4, 3, 3, 4>` does (exact derivative of the un-normalised Eigen
  quaternion sandwich).  PARITY UNPINNED (pycolmap/pyceres absent); checked
  against central differences in tests instead.
* `quat_plus_jacobian` - Ceres 2.x `EigenQuaternionManifold::PlusJacobian`
  (ba_utils.py:247).
* `huber_rho` - Ceres `HuberLoss(2.0)` (ba_utils.py:236).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------- #
#  Residual + Jacobian (reference: ba_utils.py:56-68 -> pycolmap ReprojErrorCost)
# --------------------------------------------------------------------------- #
def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _skew(v):
    z = np.zeros_like(v[..., 0])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def transform_point(q_xyzw, t, X):
    """Eigen `Quaternion::_transformVector` (no normalisation) + translation."""
    a = q_xyzw[..., :3]
    w = q_xyzw[..., 3:4]
    uv = 2.0 * _cross(a, X)
    return X + w * uv + _cross(a, uv) + t


def reproj_residual_jacobian(pose_idx, point_idx, uv, q, t, X, intr):
    """Per-observation residual r[n,2] and Jacobians Jq[n,2,4] (ambient,
    xyzw order), Jt[n,2,3], JX[n,2,3].

    pose_idx/point_idx: int arrays [n]; uv [n,2]; q [P,4] xyzw; t [P,3];
    X [Q,3]; intr = (fx, fy, cx, cy).
    """
    pose_idx = np.asarray(pose_idx, np.int64)
    point_idx = np.asarray(point_idx, np.int64)
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    q = np.asarray(q, np.float64).reshape(-1, 4)
    t = np.asarray(t, np.float64).reshape(-1, 3)
    X = np.asarray(X, np.float64).reshape(-1, 3)
    fx, fy, cx, cy = [float(v) for v in intr]

    qq = q[pose_idx]
    tt = t[pose_idx]
    XX = X[point_idx]
    a = qq[:, :3]
    w = qq[:, 3]

    p = transform_point(qq, tt, XX)
    d = 1.0 / p[:, 2]
    r = np.stack([fx * p[:, 0] * d + cx - uv[:, 0],
                  fy * p[:, 1] * d + cy - uv[:, 1]], axis=-1)

    # d r / d p
    n = len(pose_idx)
    Jp = np.zeros((n, 2, 3))
    Jp[:, 0, 0] = fx * d
    Jp[:, 0, 2] = -fx * p[:, 0] * d * d
    Jp[:, 1, 1] = fy * d
    Jp[:, 1, 2] = -fy * p[:, 1] * d * d

    # d p / d X = I + 2 w [a]x + 2 (a a^T - |a|^2 I)
    I3 = np.eye(3)[None]
    aa = a[:, :, None] * a[:, None, :]
    a2 = np.sum(a * a, axis=-1)[:, None, None]
    dp_dX = I3 + 2.0 * w[:, None, None] * _skew(a) + 2.0 * (aa - a2 * I3)

    # d p / d a = -2 w [X]x + 2 ((a.X) I + a X^T - 2 X a^T) ; d p / d w = 2 a x X
    aX = np.sum(a * XX, axis=-1)[:, None, None]
    dp_da = (-2.0 * w[:, None, None] * _skew(XX)
             + 2.0 * (aX * I3 + a[:, :, None] * XX[:, None, :]
                      - 2.0 * XX[:, :, None] * a[:, None, :]))
    dp_dw = 2.0 * _cross(a, XX)
    dp_dq = np.concatenate([dp_da, dp_dw[:, :, None]], axis=-1)   # [n,3,4]

    Jq = Jp @ dp_dq
    Jt = Jp.copy()
    JX = Jp @ dp_dX
    return r, Jq, Jt, JX


def quat_plus_jacobian(q_xyzw):
    """Ceres `EigenQuaternionManifold::PlusJacobian`, 4x3, storage x,y,z,w."""
    x, y, z, w = [q_xyzw[..., i] for i in range(4)]
    J = np.stack([np.stack([w, z, -y], -1),
                  np.stack([-z, w, x], -1),
                  np.stack([y, -x, w], -1),
                  np.stack([-x, -y, -z], -1)], -2)
    return J


def quat_plus(q_xyzw, delta):
    """Ceres `EigenQuaternionManifold::Plus`: q_new = exp(delta) (x) q."""
    q = np.asarray(q_xyzw, np.float64)
    delta = np.asarray(delta, np.float64)
    nd = np.linalg.norm(delta)
    if nd == 0.0:
        return q.copy()
    s = np.sin(nd) / nd
    dq = np.array([s * delta[0], s * delta[1], s * delta[2], np.cos(nd)])
    x1, y1, z1, w1 = dq
    x2, y2, z2, w2 = q
    return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                     w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])


def huber_rho(s, delta=2.0):
    """Ceres HuberLoss on s = |r|^2: returns (rho, rho')."""
    s = np.asarray(s, np.float64)
    b = delta * delta
    r = np.sqrt(np.maximum(s, 1e-300))
    rho = np.where(s > b, 2.0 * delta * r - b, s)
    rho1 = np.where(s > b, delta / r, 1.0)
    return rho, np.maximum(rho1, np.finfo(np.float64).tiny)


# --------------------------------------------------------------------------- #
#  Problem assembly (reference: ba_utils.py:146-165, :220-286)
# --------------------------------------------------------------------------- #
def local_window(center_kf_idx, window_size):
    """ba_utils.py:155-157."""
    first_opt = max(1, center_kf_idx - window_size + 1)
    opt_kf = list(range(first_opt, center_kf_idx + 1))
    fix_kf = list(range(0, first_opt))
    return opt_kf, fix_kf


def assemble_core_ba(points, kf_uv, opt_kf_idx, fix_kf_idx, max_points=None):
    """Walk the map exactly like `_core_ba` (ba_utils.py:259-282).

    points : iterable of (point_key, observations) in dict order, where
             observations = list of (kf_idx, kp_idx)
    kf_uv  : callable (kf_idx, kp_idx) -> (u, v)
    Returns dict with point_keys (in block order), obs_point (index into
    point_keys), obs_kf, obs_uv.
    """
    opt = list(opt_kf_idx)
    fix = list(fix_kf_idx)
    point_keys, obs_point, obs_kf, obs_uv = [], [], [], []
    added_pts = 0
    for key, observations in points:
        if not any(f in opt for f, _ in observations):
            continue
        if max_points and added_pts >= max_points:
            continue
        point_keys.append(key)
        added_pts += 1
        for f_idx, kp_idx in observations:
            if (f_idx not in opt) and (f_idx not in fix):
                continue
            u, v = kf_uv(f_idx, kp_idx)
            obs_point.append(added_pts - 1)
            obs_kf.append(f_idx)
            obs_uv.append((float(u), float(v)))
    return {
        "point_keys": point_keys,
        "obs_point": np.asarray(obs_point, np.int32),
        "obs_kf": np.asarray(obs_kf, np.int32),
        "obs_uv": np.asarray(obs_uv, np.float64).reshape(-1, 2),
    }
