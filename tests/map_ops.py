"""A scripted sequence of map mutations (the calls the reference's pipeline makes:
two_view_bootstrap.py:399-408, triangulation_utils.py:84-102, ba_utils.py:269, landmark_utils.py:138)
applied to ANY map implementing the reference's container API, and a dump of the resulting state."""
import numpy as np


def run(m):
    rng = np.random.default_rng(7)
    m.add_pose(np.eye(4), True)
    T = np.eye(4); T[0, 3] = 0.5
    m.add_pose(T, False); m.add_pose(T @ T, True)
    ids = m.add_points(rng.uniform(-5, 5, (40, 3)), rng.uniform(0, 1, (40, 3)), keyframe_idx=0)
    for j, pid in enumerate(ids):
        m.points[pid].add_observation(0, j, rng.standard_normal(128).astype(np.float32))
        if j % 3:
            m.points[pid].add_observation(1, 2 * j, rng.standard_normal(128))          # float64 in, float32 unit row stored
    ids2 = m.add_points(rng.uniform(-5, 5, (25, 3)))                                   # default colour, keyframe -1
    for j, pid in enumerate(ids2):
        for f in range(j % 9):                                                         # up to 8 observations: only the last six count
            m.points[pid].add_observation(f, j + f, rng.standard_normal(128).astype(np.float32))
    # BA-style in-place update of every position (ba_utils.py:269 hands mp.position to the solver)
    for pid in list(m.points)[::2]:
        m.points[pid].position[:] = m.points[pid].position * 1.01 + 0.001
    # near-duplicates, then the merge
    base = m.points[ids[3]].position.copy()
    dup = m.add_points(np.stack([base + 0.01, base + 0.02, m.points[ids2[4]].position + 0.015]))
    for pid in dup:
        m.points[pid].add_observation(2, 7, rng.standard_normal(128).astype(np.float32))
    m.fuse_closeby_duplicate_landmarks(radius=0.05)
    ids3 = m.add_points(rng.uniform(-1, 1, (5, 3)))                                    # ids continue after a merge
    m.points[ids3[0]].add_observation(2, 1, rng.standard_normal(128).astype(np.float32))
    return m


def state(m):
    pids = m.point_ids()
    obs_len = np.array([len(m.points[p].observations) for p in pids], np.int64)
    last_desc = np.stack([np.asarray(m.points[p].observations[-1][2], np.float32) if m.points[p].observations
                          else np.zeros(128, np.float32) for p in pids])
    return {"ids": np.array(pids, np.int64), "positions": m.get_point_array(), "colours": m.get_color_array(),
            "obs_len": obs_len, "last_desc": last_desc, "n_poses": len(m.poses),
            "keyframe_indices": np.array(m.keyframe_indices, np.int64), "len": len(m),
            "kf_idx": np.array([m.points[p].keyframe_idx for p in pids], np.int64)}
