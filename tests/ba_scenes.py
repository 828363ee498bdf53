"""Synthetic BA scenes shared by CPU and GPU tests.

`reference_test_scene` regenerates the seeded scene of the reference's own
tests/test_ba_utils_T_c_w.py:116-218 (default_rng(42), 50 points in
x[-1,1] y[-0.7,0.7] z[4,8], cameras translating 0.10/frame and yawing 2 deg/frame,
fx=fy=800, 1280x960; noise 5 px / 0.5 m / 15 deg / 0.05 m) - same draws in the
same order, with a Rodrigues formula in place of cv2.Rodrigues.
`scaled_scene` is the SURVEY.md section 8(d) "C3" scene: 10 opt + 5 fixed KFs,
5000 points seen by 2-10 consecutive keyframes, KITTI intrinsics.
"""
import math
import types

import numpy as np

W, H = 1280, 960
K_TEST = np.array([[800.0, 0, W / 2.0], [0, 800.0, H / 2.0], [0, 0, 1.0]])
K_KITTI = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


class KP:
    def __init__(self, x, y):
        self.pt = (float(x), float(y))


class MapPoint:
    def __init__(self, pos):
        self.position = np.asarray(pos, np.float64).copy()
        self.observations = []


class WorldMap:
    def __init__(self):
        self.points = {}
        self.poses = []


def _rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


def _yaw(deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _inv(T):
    R, t = T[:3, :3], T[:3, 3]
    Ti = np.eye(4); Ti[:3, :3] = R.T; Ti[:3, 3] = -R.T @ t
    return Ti


def reference_test_scene(n_frames, n_points=50, add_noise=True, pix_noise=5.0,
                         pose_trans_noise=0.5, pose_rot_noise_deg=15.0, point_noise=0.05):
    if not add_noise:
        pix_noise = pose_trans_noise = pose_rot_noise_deg = point_noise = 0.0
    rng = np.random.default_rng(42)
    pts_gt = np.column_stack((rng.uniform(-1.0, 1.0, n_points), rng.uniform(-0.7, 0.7, n_points),
                              rng.uniform(4.0, 8.0, n_points)))
    poses_gt = []
    for i in range(n_frames):
        T = np.eye(4); T[:3, :3] = _yaw(i * 2.0); T[:3, 3] = [i * 0.10, 0, 0]
        poses_gt.append(T)
    wmap = WorldMap()
    kfs = []
    for T_wc_gt in poses_gt:
        t_noise = rng.normal(0.0, pose_trans_noise, 3)
        axis = rng.normal(0.0, 1.0, 3); axis /= np.linalg.norm(axis)
        angle = math.radians(pose_rot_noise_deg) * rng.normal()
        Tn = np.eye(4)
        Tn[:3, :3] = _rodrigues(axis * angle) @ T_wc_gt[:3, :3]
        Tn[:3, 3] = T_wc_gt[:3, 3] + t_noise
        T_cw = _inv(Tn)
        wmap.poses.append(T_cw)
        kfs.append(types.SimpleNamespace(pose=T_cw.copy(), kps=[]))
    for pid, Xw in enumerate(pts_gt):
        mp = MapPoint(Xw + rng.normal(0.0, point_noise, 3))
        wmap.points[pid] = mp
        for f, T_wc in enumerate(poses_gt):
            Xc = T_wc[:3, :3].T @ (Xw - T_wc[:3, 3])
            if Xc[2] <= 0:
                continue
            u = K_TEST[0, 0] * Xc[0] / Xc[2] + K_TEST[0, 2]
            v = K_TEST[1, 1] * Xc[1] / Xc[2] + K_TEST[1, 2]
            if not (0.0 <= u < W and 0.0 <= v < H):
                continue
            u += rng.normal(0.0, pix_noise); v += rng.normal(0.0, pix_noise)
            kfs[f].kps.append(KP(u, v))
            mp.observations.append((f, len(kfs[f].kps) - 1, None))
    return wmap, kfs, K_TEST


def scaled_scene(n_kf=15, n_points=5000, seed=42, pix_noise=1.0, point_noise=0.05,
                 rot_noise_deg=0.5, trans_noise=0.05, n_fixed_clean=5):
    rng = np.random.default_rng(seed)
    K = K_KITTI
    poses_gt = []
    for i in range(n_kf):
        T = np.eye(4); T[:3, :3] = _yaw(i * 1.0); T[:3, 3] = [0.02 * i, 0, 0.8 * i]
        poses_gt.append(_inv(T))            # T_cw
    wmap = WorldMap(); kfs = []
    for i, T_cw in enumerate(poses_gt):
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        Tn = T_cw.copy()
        amp = 0.0 if i < n_fixed_clean else 1.0      # gauge keyframes: already converged
        Tn[:3, :3] = _rodrigues(axis * math.radians(rot_noise_deg) * rng.normal() * amp) @ T_cw[:3, :3]
        Tn[:3, 3] += rng.normal(0, trans_noise, 3) * amp
        wmap.poses.append(Tn); kfs.append(types.SimpleNamespace(pose=Tn.copy(), kps=[]))
    for pid in range(n_points):
        first = int(rng.integers(0, n_kf - 1))
        n_obs = int(rng.integers(2, 11))
        last = min(n_kf, first + n_obs)
        # a point in front of the middle camera of its track
        Tm = _inv(poses_gt[(first + last - 1) // 2])
        Xc = np.array([rng.uniform(-8, 8), rng.uniform(-1.5, 1.5), rng.uniform(8, 40)])
        Xw = Tm[:3, :3] @ Xc + Tm[:3, 3]
        mp = MapPoint(Xw + rng.normal(0, point_noise, 3))
        for f in range(first, last):
            Xf = poses_gt[f][:3, :3] @ Xw + poses_gt[f][:3, 3]
            if Xf[2] <= 0.5:
                continue
            u = K[0, 0] * Xf[0] / Xf[2] + K[0, 2] + rng.normal(0, pix_noise)
            v = K[1, 1] * Xf[1] / Xf[2] + K[1, 2] + rng.normal(0, pix_noise)
            kfs[f].kps.append(KP(u, v))
            mp.observations.append((f, len(kfs[f].kps) - 1, None))
        if mp.observations:
            wmap.points[pid] = mp
    return wmap, kfs, K


def reproj_rmse(wmap, kfs, K, frames=None):
    sq, n = 0.0, 0
    for mp in wmap.points.values():
        for f, i, _ in mp.observations:
            if frames is not None and f not in frames:
                continue
            T = kfs[f].pose
            Xc = T[:3, :3] @ mp.position + T[:3, 3]
            u = K[0, 0] * Xc[0] / Xc[2] + K[0, 2]
            v = K[1, 1] * Xc[1] / Xc[2] + K[1, 2]
            du, dv = u - kfs[f].kps[i].pt[0], v - kfs[f].kps[i].pt[1]
            sq += du * du + dv * dv; n += 1
    return math.sqrt(sq / max(n, 1))
