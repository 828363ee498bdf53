"""Drop-in for the descriptor-matching half of the reference's `slam/core/pnp_utils.py`:
`reproject_and_match_2d3d` (pnp_utils.py:224-304) on the HIP backend, plus the cv2-free helpers
around it (`Matches2D3D`, `_project_points`, `predict_pose_const_vel`).  The PnP solvers of that
module (`solve_pnp_ransac`, `refine_pose_pnp`) call cv2 and stay with the reference; a maintainer
patches in just this function:

    import slam.core.pnp_utils as ref
    ref.reproject_and_match_2d3d = amd_pnp_utils.reproject_and_match_2d3d
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List

import numpy as np

from ... import _native

DESC_DIM = 128
MAX_OBS_CHECK = 6


@dataclass
class Matches2D3D:
    pts3d: np.ndarray          # (N,3) world
    pts2d: np.ndarray          # (N,2) image
    kp_indices: List[int]      # indices into current frame keypoints
    mp_ids: List[int]          # matched map point ids


def _pose_inverse(T):
    R, t = T[:3, :3], T[:3, 3]
    Ti = np.eye(4, dtype=T.dtype)
    Ti[:3, :3] = R.T
    Ti[:3, 3] = -R.T @ t
    return Ti


def predict_pose_const_vel(Tcw_prevprev, Tcw_prev):
    """T_pred = T_prev * inv(T_prevprev) * T_prev (pnp_utils.py:26-30)."""
    return Tcw_prev @ _pose_inverse(Tcw_prevprev) @ Tcw_prev


def _kp_coords(kps):
    if isinstance(kps, (list, tuple)):
        if len(kps) == 0:
            return np.empty((0, 2), np.float32)
        if hasattr(kps[0], "pt"):
            return np.float32([kp.pt for kp in kps])
    kps = np.asarray(kps)
    if kps.size == 0:
        return np.empty((0, 2), np.float32)
    assert kps.ndim == 2 and kps.shape[1] >= 2, "kps must be (N,2)"
    return np.ascontiguousarray(kps[:, :2], np.float32)


def snapshot_map_points(world_map):
    """SoA view of `world_map.points` for the kernel: ids, positions, and per point the descriptors
    of its last six observations (valid ones first; count 0 when the LAST observation has none,
    which is the reference's skip rule, pnp_utils.py:46-50 / :270-272)."""
    items = list(world_map.points.items())
    Q = len(items)
    ids = np.fromiter((k for k, _ in items), np.int64, Q)
    pts = np.empty((Q, 3), np.float64)
    cnt = np.zeros(Q, np.int32)
    desc = np.zeros((Q, MAX_OBS_CHECK, DESC_DIM), np.float32)
    for q, (_, mp) in enumerate(items):
        pts[q] = mp.position
        obs = mp.observations
        if not obs or obs[-1][2] is None:
            continue
        n = 0
        for _, _, d in obs[-MAX_OBS_CHECK:]:
            if d is None:
                continue
            d = np.asarray(d).reshape(-1)
            if d.shape[0] != DESC_DIM:
                continue
            desc[q, n] = d
            n += 1
        cnt[q] = n
    return ids, pts, cnt, desc


def reproject_and_match_2d3d(world_map, K, Tcw_pred, kps_cur, des_cur, img_w, img_h, radius_px: float = 12.0,
                             max_hamm: int = 64, max_l2: float = 0.8, use_cosine: bool = False, ctx=None):
    """Same signature and result as the reference (float descriptors).  `use_cosine` changes nothing
    there either: the distance is always L2, only the threshold's name differs (pnp_utils.py:118)."""
    empty = Matches2D3D(np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), [], [])
    if des_cur is None or len(des_cur) == 0 or not world_map.points:
        return empty
    pts2d = _kp_coords(kps_cur)
    if len(pts2d) == 0:
        return empty
    des = np.asarray(des_cur)
    if des.dtype == np.uint8:
        raise NotImplementedError("binary (ORB) descriptors: the Hamming branch is outside this backend's scope")
    des = np.ascontiguousarray(des, np.float32).reshape(len(pts2d), -1)
    if des.shape[1] != DESC_DIM:
        raise ValueError("reproject_and_match_2d3d expects 128-d descriptors")
    ctx = ctx or _native.default_context()
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    Td = np.ascontiguousarray(Tcw_pred, np.float64).reshape(16)
    P = _native.ptr
    if hasattr(world_map, "device_arrays"):
        # SoA map (slam/core/landmark_utils.py of this overlay): its arrays already live on the GPU,
        # only the rows touched since the last call travel; no walk over the dict of objects
        ids, pts, _, _ = world_map.soa()
        Q = len(ids)
        d_pos, d_cnt, d_desc = world_map.device_arrays(ctx)
        scr = _dev_scratch(ctx, Q, len(pts2d))
        ctx.h2d(scr["kp"], pts2d); ctx.h2d(scr["des"], des)
        _native.check(_native.lib().sslam_reproject_match_dev(
            ctx.handle, Q, P(d_pos), P(d_cnt), P(d_desc), P(Kd), P(Td), len(pts2d), P(scr["kp"]), P(scr["des"]),
            int(img_w), int(img_h), float(radius_px), float(max_l2), P(scr["out"]), None, P(scr["info"])),
            "sslam_reproject_match_dev")
        out = np.empty(Q, np.int32); info4 = np.empty(4, np.int32)
        ctx.d2h(out, scr["out"]); ctx.d2h(info4, scr["info"])
        if info4[1]:
            raise _native.NativeError(f"more than the candidate capacity of keypoints within {radius_px} px of one projection")
    else:
        ids, pts, cnt, desc = snapshot_map_points(world_map)
        out = np.full(len(ids), -1, np.int32)
        info = (C.c_int32 * 2)()
        _native.check(_native.lib().sslam_reproject_match_host(
            ctx.handle, len(ids), P(pts), P(cnt), P(desc), P(Kd), P(Td), len(pts2d), P(pts2d), P(des), int(img_w), int(img_h),
            float(radius_px), float(max_l2), P(out), None, info), "sslam_reproject_match_host")
    hit = np.flatnonzero(out >= 0)
    if len(hit) == 0:
        return empty
    return Matches2D3D(pts[hit].astype(np.float32), pts2d[out[hit]].copy(), [int(i) for i in out[hit]],
                       [int(i) for i in ids[hit]])


def _dev_scratch(ctx, Q, N):
    """Device buffers of the current-frame inputs / outputs of the association, grown on demand.  They
    live ON the context object (`ctx.scratch`, released by `Context.close`): a module-level table keyed
    by `id(ctx)` would hand the pointers of a collected context to a new one that reuses the id."""
    cur = ctx.scratch.get("reproject")
    if cur is None or cur["Q"] < Q or cur["N"] < N:
        if cur is not None:
            ctx.sync()
            for p_ in cur["_ptrs"]:
                ctx.free(p_)
        Qc, Nc = max(Q, 2 * (cur["Q"] if cur else 0), 1024), max(N, (cur["N"] if cur else 0), 1024)
        cur = {"Q": Qc, "N": Nc, "kp": ctx.malloc(Nc * 8), "des": ctx.malloc(Nc * DESC_DIM * 4),
               "out": ctx.malloc(Qc * 4), "info": ctx.malloc(16)}
        cur["_ptrs"] = (cur["kp"], cur["des"], cur["out"], cur["info"])
        ctx.scratch["reproject"] = cur
    return cur
