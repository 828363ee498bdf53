"""Generate tests/golden/trajectory_alignment.npz from the REFERENCE's own Sim(3) alignment.

Run in the build container only (needs /root/reference on disk):
    python tests/golden/make_alignment_golden.py
`slam.core.visualization_utils` imports cv2 / open3d / matplotlib at module scope; cv2 and the
map module are replaced by empty stubs (none of them is touched by the two methods used here:
Trajectory2D._cam_center_from_Tcw, visualization_utils.py:337-340, and
Trajectory2D._maybe_update_alignment, :342-358), matplotlib is the real package.  The methods
run unmodified on a plain namespace standing in for `self`.  The .npz holds inputs and the
reference's outputs (data only).
"""
import sys
import types

import numpy as np

sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
lm = types.ModuleType("slam.core.landmark_utils")
lm.Map = type("Map", (), {})
sys.modules.setdefault("slam.core.landmark_utils", lm)
import matplotlib                                                   # noqa: E402
matplotlib.use("Agg")
from slam.core.visualization_utils import Trajectory2D              # noqa: E402


def _traj(rng, n, kind):
    tt = np.linspace(0, 1, n)
    if kind == 0:      # gentle curve
        return np.stack([40 * np.sin(2 * tt), 0.5 * tt, 120 * tt], 1)
    if kind == 1:      # loop
        return np.stack([30 * np.cos(6 * tt), 0.2 * np.sin(9 * tt), 30 * np.sin(6 * tt)], 1)
    return np.cumsum(rng.standard_normal((n, 3)), 0)


def main(out="tests/golden/trajectory_alignment.npz"):
    rng = np.random.default_rng(11)
    gts, ests, ss, Rs, ts, Ks = [], [], [], [], [], []
    for case in range(9):
        n = [6, 20, 150, 150, 400, 37, 100, 100, 12][case]
        gt = _traj(rng, n, case % 3)
        A = rng.standard_normal((3, 3))
        U, _, Vt = np.linalg.svd(A)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            U[:, -1] *= -1
            R = U @ Vt
        s = float(rng.uniform(0.05, 20.0))
        t = rng.standard_normal(3) * 10
        est = ((gt - t) @ R) / s                                   # gt = s R est + t
        if case >= 3:
            est = est + rng.normal(0, 0.02, est.shape)
        if case == 8:
            est[:, 1] = 0.0                                        # planar estimate (rank-deficient)
        K = [100, 100, 100, 60, 100, 100, 30, 400, 100][case]
        me = types.SimpleNamespace(gt_xyz=[g for g in gt], est_xyz=[e for e in est],
                                   s=1.0, R=np.eye(3), t=np.zeros(3), align_ok=False)
        Trajectory2D._maybe_update_alignment(me, Kpairs=K)
        assert me.align_ok
        gts.append(gt); ests.append(est); ss.append(me.s); Rs.append(me.R); ts.append(me.t); Ks.append(K)
    T = np.eye(4)
    T[:3, :3] = Rs[2]
    T[:3, 3] = [1.0, -2.0, 3.0]
    centre = Trajectory2D._cam_center_from_Tcw(T)
    np.savez(out, n_cases=len(gts), Kpairs=np.array(Ks), s=np.array(ss), R=np.array(Rs), t=np.array(ts),
             Tcw=T, centre=centre, **{f"gt{i}": g for i, g in enumerate(gts)},
             **{f"est{i}": e for i, e in enumerate(ests)})
    print("wrote", out)


if __name__ == "__main__":
    main()
