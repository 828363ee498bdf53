"""Map containers with a structure-of-arrays core kept in step with every mutation, mirrored on the GPU.

Drop-in for the container half of the reference's slam/core/landmark_utils.py (`MapPoint`
:47-74, `Map` :80-161; `triangulate_points` is cv2 and stays with the reference): same attributes
and methods (`points` dict in insertion order, `poses`, `keyframe_indices`, `add_pose`,
`add_points`, `get_point_array`, `get_color_array`, `point_ids`, `__len__`,
`fuse_closeby_duplicate_landmarks`, `MapPoint.add_observation`, `.position`, `.observations`).

Why: every per-frame consumer of the map on the hot path - `reproject_and_match_2d3d`
(pnp_utils.py:224-304) and `_core_ba` (ba_utils.py:220-306) - walks the dict of objects in Python
to rebuild the same arrays (9.7 ms of a 14 ms association call at 5000 points).  Here the arrays
ARE the storage:

    _ids [Q] int64, _pos [Q,3] float64, _col [Q,3] float32          one row per landmark, dict order
    _dcnt [Q] int32, _desc [Q,6,128] float32                         descriptors of the last six
                                                                     observations (valid ones first;
                                                                     count 0 when the LAST has none:
                                                                     the reference's skip rule)

`points` is a dict subclass that tells the map about every DIRECT mutation - `points.pop(pid)`
(triangulation_utils.py:105), `points[pid] = MapPoint(id=.., position=..)` (the reference's tests),
`del`, `clear`, `update` - so the arrays never go out of step with the dict: a removed landmark is
detached (it keeps its own copy of position / colour), its row becomes a hole that the next array
consumer compacts away; an assigned landmark is attached to a fresh row at once.

`MapPoint.position` is a property over row `_pos[row]`: `mp.position[:] = X` (what BA does,
ba_utils.py:269) and `mp.position = X` (what the duplicate merge does, :157) both write the array
in place.  `add_observation` refreshes the point's descriptor rows.  `device_arrays(ctx)` returns
device pointers of (positions, counts, descriptors), uploading positions / counts whole (a few
tens of KB) and only the descriptor rows touched since the last call.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

DESC_DIM = 128
MAX_OBS_CHECK = 6


def _canon_desc(desc):
    """Reference landmark_utils.py:25-41: torch -> numpy, binary descriptors flattened, float
    descriptors as unit-norm float32 rows."""
    if hasattr(desc, "detach"):
        desc = desc.detach().to("cpu")
        desc = desc.contiguous().numpy() if str(desc.dtype) == "torch.uint8" else desc.float().contiguous().numpy()
    d = np.asarray(desc)
    if d.dtype == np.uint8:
        return d.reshape(-1)
    d = d.astype(np.float32, copy=False)
    n = np.linalg.norm(d) + 1e-8
    return (d / n).reshape(-1)


class MapPoint:
    """One landmark.  Attached to a map it is a row of that map's arrays plus its observation list;
    built on its own - `MapPoint(id=0, position=np.zeros(3))`, the reference's dataclass signature
    (landmark_utils.py:47-70) - it carries its own position / colour until a map adopts it."""
    __slots__ = ("id", "keyframe_idx", "observations", "_map", "_row", "_own_pos", "_own_col")

    def __init__(self, id: int, position, keyframe_idx: int = -1, colour=None, observations=None):
        self.id = int(id)
        self.keyframe_idx = int(keyframe_idx)
        self.observations: List[Tuple[int, int, np.ndarray]] = [] if observations is None else observations
        self._map = None
        self._row = -1
        self._own_pos = np.asarray(position, np.float64).reshape(3)
        self._own_col = np.ones(3, np.float32) if colour is None else np.asarray(colour, np.float32).reshape(3)

    @classmethod
    def _attached(cls, pid: int, owner: "Map", row: int, keyframe_idx: int = -1) -> "MapPoint":
        mp = cls.__new__(cls)
        mp.id = int(pid)
        mp.keyframe_idx = int(keyframe_idx)
        mp.observations = []
        mp._map = owner
        mp._row = row
        mp._own_pos = mp._own_col = None
        return mp

    def _detach(self) -> None:
        """Leave the map keeping the current values (the caller of `points.pop` may still use the object)."""
        if self._map is not None:
            self._own_pos = self._map._pos[self._row].copy()
            self._own_col = self._map._col[self._row].copy()
            self._map, self._row = None, -1

    @property
    def position(self) -> np.ndarray:
        if self._map is None:
            return self._own_pos
        return self._map._pos[self._row]                 # a view: in-place writes land in the array

    @position.setter
    def position(self, value) -> None:
        if self._map is None:
            self._own_pos = np.asarray(value, np.float64).reshape(3)
        else:
            self._map._pos[self._row] = np.asarray(value, np.float64).reshape(3)

    @property
    def colour(self) -> np.ndarray:
        if self._map is None:
            return self._own_col
        return self._map._col[self._row]

    @colour.setter
    def colour(self, value) -> None:
        if self._map is None:
            self._own_col = np.asarray(value, np.float32).reshape(3)
        else:
            self._map._col[self._row] = np.asarray(value, np.float32).reshape(3)

    def add_observation(self, keyframe_idx: int, kp_idx: int, descriptor) -> None:
        """Register that *kp_idx* in *keyframe_idx* observes this landmark (reference :72-74)."""
        self.observations.append((keyframe_idx, kp_idx, _canon_desc(descriptor)))
        if self._map is not None:
            self._map._refresh_desc(self)

    def __repr__(self) -> str:
        return f"MapPoint(id={self.id}, position={self.position!r}, keyframe_idx={self.keyframe_idx}, " \
               f"observations={len(self.observations)})"


class _PointDict(dict):
    """`Map.points`: a dict in insertion order whose direct mutations keep the owning map's arrays in
    step (the reference mutates it directly: triangulation_utils.py:105, tests/test_landmark_utils.py)."""
    __slots__ = ("_owner",)

    def __init__(self, owner: "Map"):
        super().__init__()
        self._owner = owner

    def __setitem__(self, pid, mp) -> None:
        dict.__setitem__(self, pid, self._owner._adopt(pid, mp, dict.get(self, pid)))

    def __delitem__(self, pid) -> None:
        mp = dict.__getitem__(self, pid)
        dict.__delitem__(self, pid)
        self._owner._release(mp)

    def pop(self, pid, *default):
        if pid in self:
            mp = dict.pop(self, pid)
            self._owner._release(mp)
            return mp
        if default:
            return default[0]
        raise KeyError(pid)

    def popitem(self):
        pid, mp = dict.popitem(self)
        self._owner._release(mp)
        return pid, mp

    def clear(self) -> None:
        for mp in list(self.values()):
            self._owner._release(mp)
        dict.clear(self)

    def update(self, *args, **kwargs) -> None:
        for pid, mp in dict(*args, **kwargs).items():
            self[pid] = mp

    def setdefault(self, pid, default=None):
        if pid not in self:
            self[pid] = default
        return dict.__getitem__(self, pid)

    def __ior__(self, other):
        self.update(other)
        return self


class Map:
    """3-D points + camera trajectory (reference :80-161), SoA inside."""

    def __init__(self) -> None:
        self.points: Dict[int, MapPoint] = _PointDict(self)
        self.keyframe_indices: List[int] = []
        self.poses: List[np.ndarray] = []
        self._next_pid = 0
        self._n = 0                                      # rows in use (holes of removed landmarks included)
        self._holes = False                              # a landmark was removed directly: compact before the arrays are read
        cap = 1024
        self._ids = np.zeros(cap, np.int64)
        self._pos = np.zeros((cap, 3), np.float64)
        self._col = np.ones((cap, 3), np.float32)
        self._dcnt = np.zeros(cap, np.int32)
        self._desc = np.zeros((cap, MAX_OBS_CHECK, DESC_DIM), np.float32)
        self._dirty_lo, self._dirty_hi = 0, 0            # descriptor rows [lo, hi) changed since the last device sync
        self._dev = None                                 # (ctx, cap, pts_ptr, cnt_ptr, desc_ptr)

    @classmethod
    def from_reference(cls, ref_map) -> "Map":
        """Copy of a dict-of-objects map (the reference's `Map`, or anything with `.points` {id: obj
        with .position / .observations [/ .colour / .keyframe_idx]}, `.poses`, `.keyframe_indices`):
        same ids, same order, same observation lists."""
        m = cls()
        items = list(ref_map.points.items())
        k = len(items)
        m._grow(k)
        for r, (pid, src) in enumerate(items):
            m._ids[r] = int(pid)
            m._pos[r] = np.asarray(src.position, np.float64).reshape(3)
            col = getattr(src, "colour", None)
            if col is not None:
                m._col[r] = np.asarray(col, np.float32).reshape(3)
            mp = MapPoint._attached(int(pid), m, r, getattr(src, "keyframe_idx", -1))
            mp.observations = list(src.observations)
            dict.__setitem__(m.points, int(pid), mp)
        m._n = k
        m._next_pid = getattr(ref_map, "_next_pid", (max((int(p) for p, _ in items), default=-1) + 1))
        m.poses = [np.array(p, copy=True) for p in getattr(ref_map, "poses", [])]
        m.keyframe_indices = list(getattr(ref_map, "keyframe_indices", []))
        m.resync()
        m._touch(0, k)
        return m

    # ---------------- Camera trajectory ---------------- #
    def add_pose(self, pose_c_w: np.ndarray, is_keyframe: bool) -> None:
        assert pose_c_w.shape == (4, 4), "Pose must be 4×4 homogeneous matrix"
        self.poses.append(pose_c_w.copy())
        if is_keyframe:
            self.keyframe_indices.append(len(self.poses) - 1)

    # ---------------- Landmarks ------------------------ #
    def _grow(self, need: int) -> None:
        cap = len(self._ids)
        if need <= cap:
            return
        new = max(need, 2 * cap)
        for name in ("_ids", "_pos", "_col", "_dcnt", "_desc"):
            old = getattr(self, name)
            arr = np.zeros((new,) + old.shape[1:], old.dtype)
            arr[:cap] = old
            setattr(self, name, arr)
        self._col[cap:] = 1.0

    def _touch(self, lo: int, hi: int) -> None:
        if self._dirty_lo == self._dirty_hi:
            self._dirty_lo, self._dirty_hi = lo, hi
        else:
            self._dirty_lo, self._dirty_hi = min(self._dirty_lo, lo), max(self._dirty_hi, hi)

    def add_points(self, pts3d: np.ndarray, colours: Optional[np.ndarray] = None, keyframe_idx: int = -1) -> List[int]:
        """Add a set of 3-D points and return the list of newly assigned ids (reference :99-117)."""
        pts3d = np.asarray(pts3d)
        if pts3d.ndim != 2 or pts3d.shape[1] != 3:
            raise ValueError("pts3d must be (N,3)")
        k = len(pts3d)
        self._compact()
        self._grow(self._n + k)
        r0 = self._n
        self._pos[r0:r0 + k] = pts3d.astype(np.float64)
        self._col[r0:r0 + k] = 1.0 if colours is None else np.asarray(colours, np.float32)
        self._dcnt[r0:r0 + k] = 0
        new_ids = list(range(self._next_pid, self._next_pid + k))
        self._ids[r0:r0 + k] = new_ids
        for j, pid in enumerate(new_ids):
            dict.__setitem__(self.points, pid, MapPoint._attached(pid, self, r0 + j, keyframe_idx))
        self._next_pid += k
        self._n += k
        self._touch(r0, r0 + k)
        return new_ids

    # ---- direct mutations of `points` (called by _PointDict)
    def _adopt(self, pid, mp, old):
        """`points[pid] = mp`: the landmark becomes a row of the arrays.  A new key is appended (dict
        insertion order = row order); an existing key keeps its place, so its row is reused."""
        if not hasattr(mp, "observations") or not hasattr(mp, "position"):
            raise TypeError("Map.points values must be MapPoint-like (position, observations)")
        if old is mp and getattr(mp, "_map", None) is self:
            return mp
        if not isinstance(mp, MapPoint):                 # a foreign landmark object (e.g. the reference's dataclass):
            src = mp                                     # the map stores its own MapPoint with the same values
            mp = MapPoint(getattr(src, "id", pid), src.position, getattr(src, "keyframe_idx", -1),
                          getattr(src, "colour", None), list(src.observations))
        if mp._map is not None:                          # attached elsewhere (or to another key): move a copy of its values
            mp._detach()
        if old is not None:
            row = old._row
            old._detach()
        else:
            self._grow(self._n + 1)
            row = self._n
            self._n += 1
        self._ids[row] = int(pid)
        self._pos[row] = mp._own_pos
        self._col[row] = mp._own_col
        mp._map, mp._row = self, row
        mp._own_pos = mp._own_col = None
        self._refresh_desc(mp)
        return mp

    def _release(self, mp) -> None:
        """`points.pop(pid)` / `del points[pid]`: the row becomes a hole until the next compaction."""
        if isinstance(mp, MapPoint) and mp._map is self:
            mp._detach()
            self._holes = True

    def _compact(self) -> None:
        """Close the holes left by directly removed landmarks: rows back in dict order, `_n == len(points)`."""
        if not self._holes:
            return
        keep = np.fromiter((mp._row for mp in self.points.values()), np.int64, len(self.points))
        n = len(keep)
        for name in ("_ids", "_pos", "_col", "_dcnt", "_desc"):
            arr = getattr(self, name)
            arr[:n] = arr[keep]                          # keep is increasing: fancy indexing copies first
        for r, mp in enumerate(self.points.values()):
            mp._row = r
        self._n = n
        self._holes = False
        self._touch(0, n)

    def _refresh_desc(self, mp: MapPoint) -> None:
        """Descriptor rows of one landmark from its observation list (pnp_utils.py:46-50, :107-120,
        :270-272: last six observations, those with a 128-d float descriptor, none at all when the
        LAST observation has no descriptor)."""
        r = mp._row
        obs = mp.observations
        n = 0
        if obs and obs[-1][2] is not None:
            for _, _, d in obs[-MAX_OBS_CHECK:]:
                if d is None:
                    continue
                d = np.asarray(d).reshape(-1)
                if d.shape[0] != DESC_DIM or d.dtype == np.uint8:
                    continue
                self._desc[r, n] = d
                n += 1
        self._dcnt[r] = n
        self._touch(r, r + 1)

    def resync(self) -> None:
        """Rebuild every descriptor row from the observation lists (only needed after code appended to
        `mp.observations` directly instead of calling `add_observation`)."""
        for mp in self.points.values():
            self._refresh_desc(mp)

    # ---------------- Convenience accessors ------------ #
    def get_point_array(self) -> np.ndarray:
        self._compact()
        return self._pos[:self._n].copy() if self._n else np.empty((0, 3))

    def get_color_array(self) -> np.ndarray:
        self._compact()
        return self._col[:self._n].copy() if self._n else np.empty((0, 3), np.float32)

    def point_ids(self) -> List[int]:
        return list(self.points.keys())

    def __len__(self) -> int:
        return len(self.points)

    # ---------------- SoA views (host) ------------------ #
    def soa(self):
        """(ids [Q], positions [Q,3], descriptor counts [Q], descriptors [Q,6,128]) in dict order - views."""
        self._compact()
        n = self._n
        return self._ids[:n], self._pos[:n], self._dcnt[:n], self._desc[:n]

    # ---------------- Merging landmarks ---------------- #
    def fuse_closeby_duplicate_landmarks(self, radius: float = 0.05) -> None:
        """Average-merge landmarks whose centres are closer than ``radius`` (reference :138-161);
        the surviving rows are compacted so the arrays stay in dict order."""
        if len(self.points) < 2:
            return
        from scipy.spatial import cKDTree
        self._compact()
        ids = list(self.points.keys())
        pts = self._pos[:self._n]
        pairs = sorted(cKDTree(pts).query_pairs(radius))
        removed = set()
        for i, j in pairs:
            ida, idb = ids[i], ids[j]
            if idb in removed or ida in removed:
                continue
            self.points[ida].position = (self.points[ida].position + self.points[idb].position) * 0.5
            removed.add(idb)
        for idx in removed:
            self.points.pop(idx, None)                   # detaches the landmark, leaves a hole
        self._compact()

    # ---------------- device mirror --------------------- #
    def device_arrays(self, ctx):
        """Device pointers (positions f64 [Q,3], counts i32 [Q], descriptors f32 [Q,6,128]) of the current
        map on `ctx`'s GPU, brought up to date: positions and counts are uploaded whole (BA rewrites
        positions in place without telling anyone; 28 bytes per landmark), descriptors only for the
        rows touched since the last call."""
        self._compact()
        n = self._n
        cap = len(self._ids)
        row_bytes = MAX_OBS_CHECK * DESC_DIM * 4
        if self._dev is None or self._dev[0] is not ctx or self._dev[1] != cap:
            if self._dev is not None:
                old = self._dev
                old[0].sync()
                for p in old[2:]:
                    old[0].free(p)
            self._dev = (ctx, cap, ctx.malloc(cap * 24), ctx.malloc(cap * 4), ctx.malloc(cap * row_bytes))
            self._dirty_lo, self._dirty_hi = 0, n
        _, _, d_pos, d_cnt, d_desc = self._dev
        if n:
            ctx.h2d(d_pos, self._pos[:n])
            ctx.h2d(d_cnt, self._dcnt[:n])
            lo, hi = self._dirty_lo, min(self._dirty_hi, n)
            if hi > lo:
                ctx.h2d(d_desc + lo * row_bytes, self._desc[lo:hi])
        self._dirty_lo = self._dirty_hi = 0
        return d_pos, d_cnt, d_desc
