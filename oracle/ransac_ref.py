"""CPU oracle: `cv2.findFundamentalMat(pts1, pts2, cv2.FM_RANSAC, thresh, conf)` restated.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy, sequential, follows the reference's
only call site slam/core/features_utils.py:185-200.

PARITY UNPINNED: `opencv_python==4.11.0.86` (requirements.txt:4) is absent here.  Restated from
OpenCV 4.x's published classic (non-USAC) path:
  modules/calib3d/src/fundam.cpp   findFundamentalMat (method dispatch: RANSAC iff >= 15 points,
                                   else LMedS), FMEstimatorCallback::{runKernel -> run7Point,
                                   computeError, checkSubset}
  modules/calib3d/src/ptsetreg.cpp RANSACPointSetRegistrator::run, LMeDSPointSetRegistrator::run,
                                   getSubset, findInliers, RANSACUpdateNumIters
  modules/core: cv::RNG (multiply-with-carry), cv::solveCubic
The 7-point null space is taken from numpy's SVD (OpenCV: its own Jacobi SVD); the two spans
agree, the roots of the cubic are the same matrices up to rounding.
"""
from __future__ import annotations

import math

import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)
DBL_MIN = float(np.finfo(np.float64).tiny)
MODEL_POINTS = 7


class CvRNG:
    """cv::RNG: state = (uint32)state * 4164903690 + (state >> 32)."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def update_num_iters(p, ep, model_points, max_iters):
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, DBL_MIN)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < DBL_MIN:
        return 0
    num, denom = math.log(num), math.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))


def _last_point_collinear(pts, idx):
    i = len(idx) - 1
    xi, yi = pts[idx[i]]
    for j in range(i):
        dx1, dy1 = float(pts[idx[j]][0]) - float(xi), float(pts[idx[j]][1]) - float(yi)
        for k in range(j):
            dx2, dy2 = float(pts[idx[k]][0]) - float(xi), float(pts[idx[k]][1]) - float(yi)
            if abs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def get_subset(p1, p2, rng, max_attempts=10000):
    n = len(p1)
    for _ in range(max_attempts):
        idx = []
        for _i in range(MODEL_POINTS):
            v = rng.uniform(0, n)
            while v in idx:
                v = rng.uniform(0, n)
            idx.append(v)
        if not _last_point_collinear(p1, idx) and not _last_point_collinear(p2, idx):
            return idx
    return None


def solve_cubic(c):
    a0, a1, a2, a3 = (float(v) for v in c)
    if a0 == 0:
        if a1 == 0:
            if a2 == 0:
                return []
            return [-a3 / a2]
        d = a2 * a2 - 4 * a1 * a3
        if d < 0:
            return []
        d = math.sqrt(d)
        q1, q2 = (-a2 + d) * 0.5, (a2 + d) * -0.5
        if abs(q1) > abs(q2):
            x0, x1 = q1 / a1, a3 / q1
        else:
            x0, x1 = q2 / a1, a3 / q2
        return [x0, x1] if d > 0 else [x0]
    a0 = 1.0 / a0
    a1, a2, a3 = a1 * a0, a2 * a0, a3 * a0
    Q = (a1 * a1 - 3 * a2) * (1.0 / 9)
    R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1.0 / 54)
    Qc = Q * Q * Q
    d = Qc - R * R
    if d > 0:
        theta = math.acos(R / math.sqrt(Qc))
        t0, t1, t2 = -2 * math.sqrt(Q), theta / 3, a1 / 3
        return [t0 * math.cos(t1) - t2, t0 * math.cos(t1 + 2 * math.pi / 3) - t2,
                t0 * math.cos(t1 + 4 * math.pi / 3) - t2]
    if d == 0:
        if R >= 0:
            x0, x1 = -2 * R ** (1 / 3) - a1 / 3, R ** (1 / 3) - a1 / 3
        else:
            x0, x1 = 2 * (-R) ** (1 / 3) - a1 / 3, -((-R) ** (1 / 3)) - a1 / 3
        return [x0] if x0 == x1 else [x0, x1]
    d = math.sqrt(-d)
    e = (d + abs(R)) ** (1 / 3)
    if R > 0:
        e = -e
    return [(e + Q / e) - a1 / 3]


def run7point(m1, m2):
    """FMEstimatorCallback::runKernel for 7 points -> list of 1..3 F (3x3, F[2,2] = 1 or 0)."""
    x0, y0 = m1[:, 0].astype(np.float64), m1[:, 1].astype(np.float64)
    x1, y1 = m2[:, 0].astype(np.float64), m2[:, 1].astype(np.float64)
    A = np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones(7)], axis=1)
    _, _, Vt = np.linalg.svd(A, full_matrices=True)
    f1, f2 = Vt[7].copy(), Vt[8].copy()
    g = f1 - f2
    G, F2 = g.reshape(3, 3), f2.reshape(3, 3)
    c = np.zeros(4)
    c[0], c[3] = np.linalg.det(G), np.linalg.det(F2)
    for row in range(3):
        m = G.copy(); m[row] = F2[row]; c[1] += np.linalg.det(m)
        m = F2.copy(); m[row] = G[row]; c[2] += np.linalg.det(m)
    out = []
    for r in solve_cubic(c):
        lam, mu = r, 1.0
        s = g[8] * r + f2[8]
        F = np.zeros(9)
        if abs(s) > DBL_EPSILON:
            mu = 1.0 / s
            lam *= mu
            F[8] = 1.0
        F[:8] = g[:8] * lam + f2[:8] * mu
        out.append(F.reshape(3, 3))
    return out


def compute_error(p1, p2, F):
    """FMEstimatorCallback::computeError: float32 of max(d1^2 s1, d2^2 s2), doubles inside."""
    F = np.asarray(F, np.float64).reshape(9)
    x1, y1 = p1[:, 0].astype(np.float64), p1[:, 1].astype(np.float64)
    x2, y2 = p2[:, 0].astype(np.float64), p2[:, 1].astype(np.float64)
    with np.errstate(all="ignore"):
        a = F[0] * x1 + F[1] * y1 + F[2]
        b = F[3] * x1 + F[4] * y1 + F[5]
        c = F[6] * x1 + F[7] * y1 + F[8]
        s2 = 1.0 / (a * a + b * b)
        d2 = x2 * a + y2 * b + c
        a = F[0] * x2 + F[3] * y2 + F[6]
        b = F[1] * x2 + F[4] * y2 + F[7]
        c = F[2] * x2 + F[5] * y2 + F[8]
        s1 = 1.0 / (a * a + b * b)
        d1 = x1 * a + y1 * b + c
        return np.maximum(d1 * d1 * s1, d2 * d2 * s2).astype(np.float32)


def find_fundamental_ransac(pts1, pts2, thresh=1.0, confidence=0.99, max_iters=1000):
    """Returns (F or None, mask bool[n] or None, info)."""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    n = len(p1)
    assert n >= 8
    if thresh <= 0:
        thresh = 3
    if confidence < DBL_EPSILON or confidence > 1 - DBL_EPSILON:
        confidence = 0.99
    rng = CvRNG()
    info = {"lmeds": n < 15, "iterations": 0, "sample": -1}
    if n >= 15:
        niters, max_good, best = max(max_iters, 1), 0, None
        t = np.float32(thresh * thresh)
        it = 0
        while it < niters:
            idx = get_subset(p1, p2, rng)
            if idx is None:
                break
            for F in run7point(p1[idx], p2[idx]):
                good = int(np.count_nonzero(compute_error(p1, p2, F) <= t))
                if good > max(max_good, MODEL_POINTS - 1):
                    max_good, best = good, F
                    info["sample"] = it
                    niters = update_num_iters(confidence, (n - good) / n, MODEL_POINTS, niters)
            it += 1
        info["iterations"] = it
        if best is None:
            return None, None, info
        mask = compute_error(p1, p2, best) <= t
        info["inliers"] = int(mask.sum())
        return best, mask, info
    niters = max(update_num_iters(confidence, 0.45, MODEL_POINTS, max_iters), 1)
    min_median, best = np.finfo(np.float64).max, None
    for it in range(niters):
        idx = get_subset(p1, p2, rng)
        if idx is None:
            break
        info["iterations"] = it + 1
        for F in run7point(p1[idx], p2[idx]):
            e = np.sort(compute_error(p1, p2, F))
            med = float(e[n // 2]) if n % 2 else float(np.float32((e[n // 2 - 1] + e[n // 2]) * np.float32(0.5)))
            if med < min_median:
                min_median, best = med, F
                info["sample"] = it
    if best is None:
        return None, None, info
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - MODEL_POINTS)) * math.sqrt(min_median), 0.001)
    mask = compute_error(p1, p2, best) <= np.float32(sigma * sigma)
    info["inliers"] = int(mask.sum())
    if info["inliers"] < MODEL_POINTS:
        return None, None, info
    return best, mask, info
