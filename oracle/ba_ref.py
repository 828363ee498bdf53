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


# --------------------------------------------------------------------------- #
#  Dense trust-region solve (reference: ba_utils.py:288-293 -> pyceres.solve)
# --------------------------------------------------------------------------- #
def solve_dense_lm(q, t, pose_const, X, intr, obs_pose, obs_point, obs_uv, max_iters,
                   huber_delta=2.0, points_const=False, sparse=False):
    """What `pyceres.solve(opts, problem, summary)` does to the `_core_ba` problem, restated
    with ONE dense Jacobian and dense normal equations (no Schur elimination, no block
    structure) - small problems only.  Ceres 2.x defaults the reference leaves untouched
    (ba_utils.py:289-292 sets only max_num_iterations / linear solver / threads):
    LEVENBERG_MARQUARDT, initial_trust_region_radius 1e4, min/max_lm_diagonal 1e-6 / 1e32,
    min_relative_decrease 1e-3, radius update r / max(1/3, 1 - (2 rho - 1)^3) on success,
    r / decrease (decrease doubling) on failure, function / gradient / parameter tolerances
    1e-6 / 1e-10 / 1e-8; HuberLoss(delta) applied as sqrt(rho') scaling (corrector with
    rho'' <= 0); EigenQuaternionManifold on every quaternion.  PARITY UNPINNED (pyceres
    absent): Ceres' exact iterates depend on its linear solver and are not reproduced, the
    policy and the fixed point are.  `sparse=True` stores the SAME single Jacobian in
    scipy.sparse CSR form and factorises the full normal equations with SuperLU (still no Schur
    elimination): that is what lets the C3-size problem (~48 k rows x 14 k columns) be checked
    against this formulation.  Returns (q, t, X, info)."""
    q, t, X = np.array(q, np.float64), np.array(t, np.float64), np.array(X, np.float64)
    pose_const = np.asarray(pose_const, bool)
    opt_rows = np.flatnonzero(~pose_const)
    Po, Q, n = len(opt_rows), len(X), len(obs_pose)
    slot = -np.ones(len(q), int)
    slot[opt_rows] = np.arange(Po)
    nX = 0 if points_const else 3 * Q
    dim = 6 * Po + nX

    def cost_of(qq, tt, XX):
        r = reproj_residual_jacobian(obs_pose, obs_point, obs_uv, qq, tt, XX, intr)[0]
        if not np.all(np.isfinite(r)):
            return np.inf
        return 0.5 * float(np.sum(huber_rho(np.sum(r * r, axis=1), huber_delta)[0]))

    cost = cost_of(q, t, X)
    info = {"initial_cost": cost, "iterations": 0, "successful_steps": 0, "termination": "max iterations"}
    radius, decrease = 1e4, 2.0
    for it in range(int(max_iters)):
        info["iterations"] = it + 1
        r, Jq, Jt, JX = reproj_residual_jacobian(obs_pose, obs_point, obs_uv, q, t, X, intr)
        sw = np.sqrt(huber_rho(np.sum(r * r, axis=1), huber_delta)[1])
        pj = quat_plus_jacobian(q)
        if sparse:
            import scipy.sparse as sp
            from scipy.sparse.linalg import spsolve
            rows_, cols_, vals_ = [], [], []
            s_of = slot[np.asarray(obs_pose)]
            op = np.flatnonzero(s_of >= 0)
            Jp = np.concatenate([Jq[op] @ pj[np.asarray(obs_pose)[op]], Jt[op]], axis=2) * sw[op, None, None]
            rr = (2 * op[:, None, None] + np.arange(2)[None, :, None]) + np.zeros((1, 1, 6), int)
            cc = (6 * s_of[op][:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 2, 1), int)
            rows_.append(rr.ravel()); cols_.append(cc.ravel()); vals_.append(Jp.ravel())
            if not points_const:
                JXs = JX * sw[:, None, None]
                rr = (2 * np.arange(n)[:, None, None] + np.arange(2)[None, :, None]) + np.zeros((1, 1, 3), int)
                cc = (6 * Po + 3 * np.asarray(obs_point)[:, None, None] + np.arange(3)[None, None, :]) + np.zeros((1, 2, 1), int)
                rows_.append(rr.ravel()); cols_.append(cc.ravel()); vals_.append(JXs.ravel())
            J = sp.csr_matrix((np.concatenate(vals_), (np.concatenate(rows_), np.concatenate(cols_))),
                              shape=(2 * n, dim))
        else:
            J = np.zeros((2 * n, dim))
            for i in range(n):
                s = slot[obs_pose[i]]
                if s >= 0:
                    J[2 * i:2 * i + 2, 6 * s:6 * s + 3] = sw[i] * (Jq[i] @ pj[obs_pose[i]])
                    J[2 * i:2 * i + 2, 6 * s + 3:6 * s + 6] = sw[i] * Jt[i]
                if not points_const:
                    c = 6 * Po + 3 * obs_point[i]
                    J[2 * i:2 * i + 2, c:c + 3] = sw[i] * JX[i]
        f = (r * sw[:, None]).reshape(-1)
        g = J.T @ f
        if np.max(np.abs(g), initial=0.0) < 1e-10:
            info["termination"] = "gradient tolerance"
            break
        H = J.T @ J
        if sparse:
            D = np.clip(H.diagonal(), 1e-6, 1e32) / radius
            d = spsolve((H + sp.diags(D)).tocsc(), -g)
        else:
            D = np.clip(np.diag(H), 1e-6, 1e32) / radius
            d = np.linalg.solve(H + np.diag(D), -g)
        Jd = J @ d
        model_change = -float(Jd @ (f + 0.5 * Jd))
        x_norm = np.sqrt(np.sum(X * X) + np.sum(q[opt_rows] ** 2) + np.sum(t[opt_rows] ** 2))
        if np.linalg.norm(d) <= 1e-8 * (x_norm + 1e-8):
            info["termination"] = "parameter tolerance"
            break
        qn, tn, Xn = q.copy(), t.copy(), X.copy()
        for s, row in enumerate(opt_rows):
            qn[row] = quat_plus(q[row], d[6 * s:6 * s + 3])
            tn[row] = t[row] + d[6 * s + 3:6 * s + 6]
        if not points_const:
            Xn = X + d[6 * Po:].reshape(Q, 3)
        new_cost = cost_of(qn, tn, Xn)
        rel = (cost - new_cost) / model_change if model_change > 0 else -1.0
        if rel > 1e-3 and np.isfinite(new_cost):
            change = cost - new_cost
            q, t, X, cost = qn, tn, Xn, new_cost
            info["successful_steps"] += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            decrease = 2.0
            if abs(change) < 1e-6 * cost:
                info["termination"] = "function tolerance"
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                info["termination"] = "trust region collapsed"
                break
    info["final_cost"] = cost
    return q, t, X, info
