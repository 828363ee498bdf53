"""Fundamental-matrix RANSAC / LMedS outlier filter on the HIP backend.

Stands in for `cv2.findFundamentalMat(pts1, pts2, cv2.FM_RANSAC, thresh, 0.99)` as
`filter_matches_ransac` calls it (slam/core/features_utils.py:185-200).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native


def find_fundamental_ransac(pts1, pts2, thresh: float = 1.0, confidence: float = 0.99,
                            max_iters: int = 1000, ctx=None):
    """pts1, pts2: [n,2] matched pixels (cast to float32 as the reference does), n >= 8.
    Returns (F [3,3] float64 or None, mask [n] bool or None, info dict) - (None, None) where
    cv2 returns (None, None)."""
    ctx = ctx or _native.default_context()
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    if len(p1) != len(p2):
        raise ValueError("pts1 / pts2 length mismatch")
    n = len(p1)
    mask = np.zeros(n, np.uint8)
    F = np.zeros(9, np.float64)
    info = (C.c_int * 4)()
    P = _native.ptr
    _native.check(_native.lib().sslam_fmat_ransac_host(
        ctx.handle, n, P(p1), P(p2), float(thresh), float(confidence), int(max_iters), P(mask), P(F), info),
        "sslam_fmat_ransac_host")
    meta = {"inliers": int(info[0]), "iterations": int(info[1]), "lmeds": bool(info[2]), "sample": int(info[3])}
    # cv2: RANSAC succeeds with any accepted model (>= 7 inliers by construction); LMedS needs >= 7
    if info[0] < 0 or (meta["lmeds"] and info[0] < 7):
        return None, None, meta
    return F.reshape(3, 3), mask.astype(bool), meta


def filter_matches_dev(ctx, n_max: int, n_dev, xy1_dev, xy2_dev, ij_dev, ij_out_dev, info_out_dev,
                       thresh: float = 1.0, confidence: float = 0.99, max_iters: int = 1000,
                       mask_out_dev=None, F_out_dev=None):
    """Device-resident `filter_matches_ransac` (slam/core/features_utils.py:185-200): all arguments are
    device pointers (ints); consumes the matcher's `(ij, info[0])` and leaves the kept pairs and their
    count on the device.  Enqueued on ctx's stream; nothing is read back here."""
    _native.check(_native.lib().sslam_fmat_ransac_dev(
        ctx.handle, int(n_max), _native.ptr(n_dev), _native.ptr(xy1_dev), _native.ptr(xy2_dev), _native.ptr(ij_dev),
        float(thresh), float(confidence), int(max_iters), _native.ptr(mask_out_dev), _native.ptr(ij_out_dev),
        _native.ptr(F_out_dev), _native.ptr(info_out_dev)), "sslam_fmat_ransac_dev")
