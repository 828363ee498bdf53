"""SE(3) helpers that define the BA parameterisation (numpy only).

Same names, argument meaning and return conventions as the reference's
slam/core/pose_utils.py (`project_to_SO3` :5, `_pose_inverse` :17,
`_pose_rt_to_homogenous` :52, `_pose_to_quat_trans` :63, `_quat_trans_to_pose`
:109): poses are 4x4 camera-from-world matrices, quaternions are unit, stored
(x, y, z, w) by default with w >= 0.  Implemented without SciPy; parity with the
reference's outputs is pinned by tests/golden/pose_utils.npz.
"""
from __future__ import annotations

import numpy as np

__all__ = ["project_to_SO3", "_pose_inverse", "_pose_rt_to_homogenous",
           "_pose_to_quat_trans", "_quat_trans_to_pose"]


def project_to_SO3(M):
    """Nearest rotation (Frobenius) to a 3x3 matrix, det = +1."""
    U, _, Vt = np.linalg.svd(np.asarray(M, dtype=float))
    if np.linalg.det(U @ Vt) < 0:
        U = U.copy()
        U[:, 2] = -U[:, 2]
    return U @ Vt


def _pose_inverse(T, validate=True):
    T = np.asarray(T, dtype=float)
    if T.shape != (4, 4):
        raise ValueError("T must be 4x4.")
    R = project_to_SO3(T[:3, :3]) if validate else T[:3, :3]
    out = np.eye(4)
    out[:3, :3] = R.T
    out[:3, 3] = -(R.T @ T[:3, 3])
    return out


def _pose_rt_to_homogenous(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t, dtype=float).ravel()
    return T


def _rot_to_quat_xyzw(R):
    # Shepperd: pivot on the largest of (R00, R11, R22, trace) for stability
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    k = int(np.argmax([R[0, 0], R[1, 1], R[2, 2], tr]))
    q = np.empty(4)
    if k == 3:
        q[:] = (R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1], 1.0 + tr)
    else:
        a, b, c = k, (k + 1) % 3, (k + 2) % 3
        q[a] = 1.0 - tr + 2.0 * R[a, a]
        q[b] = R[b, a] + R[a, b]
        q[c] = R[c, a] + R[a, c]
        q[3] = R[c, b] - R[b, c]
    return q / np.linalg.norm(q)


def _pose_to_quat_trans(T, ordering="xyzw"):
    """4x4 T_cw -> (unit quaternion, translation).  `ordering` is "xyzw"
    (what the BA kernel and Eigen/COLMAP expect) or "wxyz"."""
    T = np.asarray(T, dtype=float)
    assert T.shape == (4, 4)
    q = _rot_to_quat_xyzw(project_to_SO3(T[:3, :3]))
    if q[3] < 0:
        q = -q
    t = T[:3, 3].copy()
    if ordering.lower() == "xyzw":
        return q, t
    return np.array([q[3], q[0], q[1], q[2]]), t


def _quat_trans_to_pose(q, t, ordering="xyzw"):
    q = np.asarray(q, dtype=float)
    if ordering.lower() == "wxyz":
        q = np.array([q[1], q[2], q[3], q[0]])
    x, y, z, w = q / np.linalg.norm(q)
    T = np.eye(4)
    T[:3, :3] = [
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ]
    T[:3, 3] = np.asarray(t, dtype=float).reshape(3)
    return T
