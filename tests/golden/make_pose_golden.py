"""Generate tests/golden/pose_utils.npz from the REFERENCE's own pose_utils.

Run in the build container only (needs /root/reference on disk):
    python tests/golden/make_pose_golden.py
The reference module `slam.core.pose_utils` imports only numpy + scipy, so it
runs as-is.  The .npz holds inputs and the reference's outputs (data only).
"""
import sys
import numpy as np

sys.path.insert(0, "/root/reference")
from slam.core.pose_utils import (_pose_inverse, _pose_to_quat_trans,   # noqa: E402
                                  _quat_trans_to_pose)


def main(out="tests/golden/pose_utils.npz"):
    rng = np.random.default_rng(7)
    Ts, qs, ts, Tinv, Tback = [], [], [], [], []
    for i in range(64):
        A = rng.standard_normal((3, 3))
        U, _, Vt = np.linalg.svd(A)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            U[:, -1] *= -1
            R = U @ Vt
        if i % 4 == 1:                       # mild non-orthonormal drift
            R = R + 1e-3 * rng.standard_normal((3, 3))
        if i == 2:                           # 180 degree rotation (w == 0 edge)
            R = np.diag([1.0, -1.0, -1.0])
        if i == 3:
            R = np.eye(3)
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = rng.standard_normal(3) * 5
        q, t = _pose_to_quat_trans(T)
        Ts.append(T); qs.append(q); ts.append(t)
        Tinv.append(_pose_inverse(T))
        Tback.append(_quat_trans_to_pose(q, t))
    np.savez(out, T=np.array(Ts), q=np.array(qs), t=np.array(ts),
             T_inv=np.array(Tinv), T_back=np.array(Tback))
    print("wrote", out)


if __name__ == "__main__":
    main()
