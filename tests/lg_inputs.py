"""Seeded synthetic matcher inputs: keypoints uniform in
[2, W-3] x [2, H-3] of a 1241x376 frame, descriptors = row-normalised normal
draws; image 1 = shuffled, jittered, partly replaced copy of image 0 so that a
non-trivial set of true correspondences exists."""
import numpy as np

W_IMG, H_IMG = 1241, 376


def make_pair(m, n=None, seed=0, noise=0.05, drop=0.2):
    n = m if n is None else n
    rng = np.random.default_rng(seed)
    k0 = np.column_stack([rng.uniform(2, W_IMG - 3, m), rng.uniform(2, H_IMG - 3, m)]).astype(np.float32)
    d0 = rng.standard_normal((m, 128)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    src = rng.permutation(m)
    src = np.resize(src, n)
    k1 = (k0[src] + np.float32([3.0, 1.0]) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    d1 = d0[src] + noise * rng.standard_normal((n, 128)).astype(np.float32)
    nd = int(drop * n)
    d1[:nd] = rng.standard_normal((nd, 128))
    k1[:nd] = np.column_stack([rng.uniform(2, W_IMG - 3, nd), rng.uniform(2, H_IMG - 3, nd)])
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    return k0, d0, k1.astype(np.float32), d1.astype(np.float32)
