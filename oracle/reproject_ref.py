"""CPU oracle: `reproject_and_match_2d3d` (slam/core/pnp_utils.py:224-304) restated for float
descriptors (the ALIKED path; the uint8 / Hamming branch needs cv2 and is out of scope).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy; brute-force radius search instead of
scipy's cKDTree (same set: squared Euclidean distance in float64 <= r^2).

PINNED: tests/golden/reproject_match.npz holds the outputs of the REFERENCE's own function on the
seeded scenes of tests/reproject_scenes.py (generator: tests/golden/make_reproject_golden.py).

Behaviour that is easy to miss (all reproduced):
* `use_cosine` only selects which threshold NAME is used - the distance is always L2, because
  `_best_mp_distance_to_cur_desc` calls `_desc_distance(..., metric="auto")` (:118);
* a point whose LAST observation carries no descriptor is skipped (`_choose_mp_descriptor`, :46-50,
  :270-272) even when earlier observations have one; otherwise the minimum runs over the last six
  observations that do have a descriptor (:107-120);
* greedy, in `world_map.points` order: a keypoint taken by an earlier point is gone (:260, :276).
"""
from __future__ import annotations

import numpy as np


def project_points(K, Tcw, pts_w):
    """`_project_points` (:127-141): float64 camera coordinates, float32 pixels, -1 where z <= 1e-8."""
    pts_w = np.asarray(pts_w, np.float64)
    Xc = pts_w @ Tcw[:3, :3].T + Tcw[:3, 3]
    z = Xc[:, 2]
    uv = np.full((len(pts_w), 2), -1.0, np.float32)
    valid = z > 1e-8
    if np.any(valid):
        proj = (K @ (Xc[valid] / z[valid, None]).T).T
        uv[valid] = proj[:, :2].astype(np.float32, copy=False)
    return uv, z


def reproject_and_match_2d3d(world_map, K, Tcw_pred, kps_cur, des_cur, img_w, img_h, radius_px=12.0,
                             max_l2=0.8, use_cosine=False):
    """Returns (pts3d f32 [M,3], pts2d f32 [M,2], kp_indices, mp_ids)."""
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), [], [])
    if des_cur is None or len(des_cur) == 0 or not world_map.points:
        return empty
    pts2d = np.asarray(kps_cur, np.float32).reshape(-1, 2)
    if len(pts2d) == 0:
        return empty
    des_cur = np.asarray(des_cur)
    assert des_cur.dtype != np.uint8, "binary descriptors: the Hamming branch is not restated"
    items = list(world_map.points.items())
    pts3d_all = np.asarray([mp.position for _, mp in items], np.float64)
    uv_all, z_all = project_points(np.asarray(K, np.float64), np.asarray(Tcw_pred, np.float64), pts3d_all)
    cand = np.flatnonzero((z_all > 0.0) & (uv_all[:, 0] >= 0.0) & (uv_all[:, 0] < float(img_w))
                          & (uv_all[:, 1] >= 0.0) & (uv_all[:, 1] < float(img_h)))
    kp64 = pts2d.astype(np.float64)
    used = np.zeros(len(pts2d), bool)
    thr = max_l2                                   # use_cosine picks the same number (see module doc)
    out3, out2, kpids, mpids = [], [], [], []
    r2 = float(radius_px) ** 2
    for a in cand:
        mp_id, mp = items[a]
        d2 = np.sum((kp64 - uv_all[a].astype(np.float64)) ** 2, axis=1)
        near = np.flatnonzero(d2 <= r2)
        if len(near) == 0 or not mp.observations or mp.observations[-1][2] is None:
            continue
        obs = [np.asarray(d).reshape(-1).astype(np.float32, copy=False) for _, _, d in mp.observations[-6:] if d is not None]
        best_i, best_d = -1, 1e9
        for i in near:
            if used[i]:
                continue
            cur = des_cur[i].reshape(-1).astype(np.float32, copy=False)
            d = min((float(np.linalg.norm(o - cur)) for o in obs if o.shape[0] == cur.shape[0]), default=float("inf"))
            if d < best_d:
                best_d, best_i = d, int(i)
        if best_i < 0 or best_d > thr:
            continue
        used[best_i] = True
        out3.append(pts3d_all[a].astype(np.float32)); out2.append(pts2d[best_i]); kpids.append(best_i); mpids.append(mp_id)
    if not out3:
        return empty
    return np.asarray(out3, np.float32), np.asarray(out2, np.float32), kpids, mpids
