"""Seeded scenes for `reproject_and_match_2d3d` (map points with observation descriptors, a predicted
pose, current keypoints + descriptors).  Shared by the golden generator (which feeds them to the
REFERENCE's function) and the tests (which feed the same arrays to the oracle / HIP path)."""
import types

import numpy as np

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
W, H = 1241, 376
CASES = [  # seed, n_pts, n_kp, radius, max_l2, use_cosine, pix_noise
    (0, 400, 300, 12.0, 0.8, False, 2.0), (1, 1500, 1000, 12.0, 0.8, False, 2.0), (2, 800, 600, 25.0, 0.6, False, 6.0),
    (3, 600, 500, 12.0, 0.5, True, 2.0), (4, 50, 40, 3.0, 0.8, False, 2.0)]


def unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def make_case(seed, n_pts, n_kp, radius, max_l2, use_cosine, pix_noise=2.0, desc_noise=0.25):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-25, 25, n_pts), rng.uniform(-5, 5, n_pts), rng.uniform(-10, 70, n_pts)], 1)
    ang = rng.uniform(-0.1, 0.1)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    Tcw = np.eye(4)
    Tcw[:3, :3] = R
    Tcw[:3, 3] = rng.normal(0, 0.3, 3)
    Xc = X @ R.T + Tcw[:3, 3]
    proj = (K @ (Xc / Xc[:, 2:3]).T).T[:, :2]
    base = unit(rng.standard_normal((n_pts, 128)))
    n_obs = rng.integers(0, 10, n_pts)                        # 0..9 observations, the last six count
    obs_desc, obs_valid = np.zeros((n_pts, 10, 128), np.float32), np.zeros((n_pts, 10), bool)
    for i in range(n_pts):
        for j in range(n_obs[i]):
            if rng.random() < 0.1:                            # an observation stored without descriptor
                continue
            obs_desc[i, j] = unit(base[i] + desc_noise * rng.standard_normal(128) / np.sqrt(128))
            obs_valid[i, j] = True
    vis = np.flatnonzero((Xc[:, 2] > 0.5) & (proj[:, 0] > 0) & (proj[:, 0] < W) & (proj[:, 1] > 0) & (proj[:, 1] < H))
    pick = rng.choice(vis, min(len(vis), n_kp * 2 // 3), replace=False)
    kp = [proj[pick] + rng.normal(0, pix_noise, (len(pick), 2))]
    des = [unit(base[pick] + desc_noise * rng.standard_normal((len(pick), 128)) / np.sqrt(128))]
    n_cl = n_kp - len(pick)
    kp.append(np.stack([rng.uniform(0, W, n_cl), rng.uniform(0, H, n_cl)], 1))
    des.append(unit(rng.standard_normal((n_cl, 128))))
    kp = np.concatenate(kp).astype(np.float32)
    des = np.concatenate(des).astype(np.float32)
    perm = rng.permutation(len(kp))
    kp, des = kp[perm], des[perm]
    ids = rng.permutation(10 * n_pts)[:n_pts]                 # arbitrary, non-contiguous map ids
    wmap = types.SimpleNamespace(points={})
    for i in range(n_pts):
        obs = [(int(j), int(j), obs_desc[i, j].copy() if obs_valid[i, j] else None) for j in range(n_obs[i])]
        wmap.points[int(ids[i])] = types.SimpleNamespace(position=X[i].copy(), observations=obs)
    digest = float(X.sum() + kp.astype(np.float64).sum() + des.astype(np.float64).sum()
                   + obs_desc.astype(np.float64).sum() + ids.sum())
    return dict(wmap=wmap, K=K, Tcw=Tcw, kp=kp, des=des, W=W, H=H, radius=radius, max_l2=max_l2,
                use_cosine=use_cosine, digest=digest)
