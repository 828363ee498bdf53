"""Synthetic two-view correspondences (KITTI intrinsics) with outliers, for the RANSAC tests."""
import numpy as np

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])


def make_matches(n, outlier_frac=0.3, noise=0.3, seed=0, planar=False):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-15, 15, n), rng.uniform(-4, 4, n), rng.uniform(8, 60, n)], 1)
    if planar:
        X[:, 2] = 20.0 + 0.1 * X[:, 0]
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.3, -0.05, 1.1])
    x1 = (K @ X.T).T
    x2 = (K @ (R @ X.T + t[:, None])).T
    p1 = x1[:, :2] / x1[:, 2:]
    p2 = x2[:, :2] / x2[:, 2:]
    p1 += rng.normal(0, noise, p1.shape)
    p2 += rng.normal(0, noise, p2.shape)
    n_out = int(round(outlier_frac * n))
    out = rng.choice(n, n_out, replace=False)
    p2[out] = np.stack([rng.uniform(0, 1241, n_out), rng.uniform(0, 376, n_out)], 1)
    truth = np.ones(n, bool)
    truth[out] = False
    return p1.astype(np.float32), p2.astype(np.float32), truth
