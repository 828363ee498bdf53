"""KeyPoint / DMatch carriers for environments without OpenCV.

Downstream of the hot path the reference only ever reads `.pt` on keypoints
and `.queryIdx` / `.trainIdx` (and `.distance` on the OpenCV-BF path) on
matches; when `cv2` is importable its own classes are
used so the rest of the reference pipeline keeps working unchanged.
"""
from __future__ import annotations

from collections import deque as _deque
from itertools import repeat as _repeat
from operator import attrgetter as _attrgetter

import numpy as _np

try:                                       # (cv2 is absent from the build image: tests/cv2_stub.py stands in for it)
    import cv2 as _cv2
    KeyPoint = _cv2.KeyPoint
    DMatch = _cv2.DMatch
    HAVE_CV2 = True
    _kp_convert = getattr(_cv2, "KeyPoint_convert", None)      # both overloads: keypoints -> [N,2] float32, points2f -> keypoints
    keypoint_shells = None                 # cv2.KeyPoint stores its fields at construction; KeyPoint_convert builds a frame's in one C pass

    def match_shells(n, src=None, start=0):
        """n - start cv2.DMatch(0, 0, 0, 0.0) objects to be filled by `bind_matches` once the pair array has arrived (cv2's fields
        are writable): the constructions happen while the GPU is still matching, two attribute stores per match are left for
        afterwards.  `src` is unused with cv2's class (the duck type resolves lazily against it)."""
        return list(map(DMatch, _repeat(0, max(0, n - start)), _repeat(0), _repeat(0), _repeat(0.0))), None
except Exception:                          # noqa: BLE001
    HAVE_CV2 = False
    _kp_convert = None

    class _KeyPointDefaults:
        __slots__ = ()
        size = 1.0
        angle = -1.0
        response = 0.0
        octave = 0
        class_id = -1

    class KeyPoint(_KeyPointDefaults):
        """Duck type of cv2.KeyPoint(x, y, size).  Only `pt` is stored per instance unless another
        field is set: the reference reads nothing else downstream, and a frame builds thousands.
        A keypoint made by `keypoint_shells` reads its `pt` from the frame's coordinate array on first use
        (then keeps it), so the objects can be built while the GPU is still extracting."""
        __slots__ = ("_pt", "_src", "_i", "__dict__")
        edits = 0          # bumped by every `pt` assignment on any instance (see keypoint_edit_epoch)

        def __init__(self, x=0.0, y=0.0, size=1.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
            self._pt = (float(x), float(y))
            if size != 1.0:
                self.size = float(size)
            if angle != -1.0:
                self.angle = float(angle)
            if response != 0.0:
                self.response = float(response)
            if octave != 0:
                self.octave = int(octave)
            if class_id != -1:
                self.class_id = int(class_id)

        @property
        def pt(self):
            try:
                return self._pt
            except AttributeError:
                p = self._pt = self._src.point(self._i)
                return p

        @pt.setter
        def pt(self, v):
            x, y = v
            self._pt = (float(x), float(y))
            KeyPoint.edits += 1

        def __repr__(self):
            return f"KeyPoint(pt={self.pt})"

    class _PointSource:
        """The [N,2] coordinates a frame's keypoint shells resolve their `pt` against (set once the
        extraction has finished).  The first few reads convert their own element (the matcher's spot check
        of 8 keypoints must not pay for the frame); from the 17th on, the whole array goes to python
        floats in one C pass."""
        __slots__ = ("xy", "pts", "reads")

        def __init__(self):
            self.xy = None
            self.pts = None
            self.reads = 0

        def point(self, i):
            pts = self.pts
            if pts is None:
                self.reads += 1
                if self.reads <= 16:
                    x, y = self.xy[i]
                    return (float(x), float(y))
                pts = self.pts = self.xy.tolist()
            x, y = pts[i]
            return (x, y)

    def keypoint_shells(n):
        """n KeyPoint objects without coordinates yet + the source to hand the [>= n, 2] array to
        (`src.xy = xy`) before anyone reads a `pt`."""
        src = _PointSource()
        new = KeyPoint.__new__
        out = []
        add = out.append
        for i in range(n):
            k = new(KeyPoint)
            k._src = src
            k._i = i
            add(k)
        return out, src

    class DMatch:
        """Duck type of cv2.DMatch(queryIdx, trainIdx, imgIdx, distance).  A match made by `match_shells` reads
        its two indices from the call's [K,2] pair array on first use (then keeps them), so the objects can be
        built while the GPU is still matching."""
        __slots__ = ("_q", "_t", "_im", "_d", "_src", "_i")
        edits = 0          # bumped by every attribute assignment on any instance (see dmatch_edit_epoch)

        def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=0.0):
            self._q = int(queryIdx)
            self._t = int(trainIdx)
            if imgIdx != 0:
                self._im = int(imgIdx)
            if distance != 0.0:
                self._d = float(distance)

        @property
        def imgIdx(self):
            try:
                return self._im
            except AttributeError:
                return 0

        @imgIdx.setter
        def imgIdx(self, v):
            self._im = int(v)
            DMatch.edits += 1

        @property
        def distance(self):
            try:
                return self._d
            except AttributeError:
                return 0.0

        @distance.setter
        def distance(self, v):
            self._d = float(v)
            DMatch.edits += 1

        def _resolve(self):
            q, t = self._src.pair(self._i)
            self._q, self._t = q, t

        @property
        def queryIdx(self):
            try:
                return self._q
            except AttributeError:
                self._resolve()
                return self._q

        @queryIdx.setter
        def queryIdx(self, v):
            if not hasattr(self, "_t"):
                self._resolve()
            self._q = int(v)
            DMatch.edits += 1

        @property
        def trainIdx(self):
            try:
                return self._t
            except AttributeError:
                self._resolve()
                return self._t

        @trainIdx.setter
        def trainIdx(self, v):
            if not hasattr(self, "_q"):
                self._resolve()
            self._t = int(v)
            DMatch.edits += 1

        def __repr__(self):
            return f"DMatch({self.queryIdx}->{self.trainIdx})"

    class _PairSource:
        """The [K,2] index pairs a call's match shells resolve against.  The first few reads convert their own row (the
        filter's spot check of eight matches must not pay for the list); from the 17th on, the whole array goes to python
        ints in one C pass - as `_PointSource` does for the keypoints."""
        __slots__ = ("ij", "pairs", "reads")

        def __init__(self):
            self.ij = None
            self.pairs = None
            self.reads = 0

        def pair(self, i):
            pairs = self.pairs
            if pairs is None:
                self.reads += 1
                if self.reads <= 16:
                    q, t = self.ij[i]
                    return (int(q), int(t))
                pairs = self.pairs = self.ij.tolist()
            return pairs[i]

    def match_shells(n, src=None, start=0):
        """n DMatch(-, -, 0, 0.0) objects without indices yet + the source to hand the [>= n, 2] int array to
        (`src.ij = ij`) before anyone reads an index.  With `src` and `start`: shells start .. n - 1 of an existing source
        (more matches came back than shells had been prepared)."""
        src = _PairSource() if src is None else src
        new = DMatch.__new__
        out = []
        add = out.append
        for i in range(start, n):
            m = new(DMatch)
            m._src = src
            m._i = i
            add(m)
        return out, src


def keypoint_edit_epoch():
    """A counter that moves whenever the `pt` of ANY KeyPoint of this module's duck type is assigned (None with cv2's own class,
    which cannot be watched): a KeyPointList made at an epoch still holds what it was built from while the counter stands still."""
    return None if HAVE_CV2 else KeyPoint.edits


def dmatch_edit_epoch():
    """A counter that moves whenever the indices of ANY DMatch of this module's duck type are assigned (None with cv2's own
    class, which cannot be watched): match objects built before an edit may only be handed out again while it stands still."""
    return None if HAVE_CV2 else DMatch.edits


def keypoints_from_xy(xy):
    """[N,2] float array -> list of KeyPoint (size 1), the bulk form of the reference's
    `[cv2.KeyPoint(x, y, 1) for x, y in kps]` (features_utils.py:62)."""
    if HAVE_CV2:
        if _kp_convert is not None and len(xy):
            # (the convert overload's own defaults are size 1, response 1: the constructor the reference calls leaves response 0)
            return list(_kp_convert(_np.ascontiguousarray(xy, _np.float32), size=1, response=0))
        return [KeyPoint(x, y, 1) for x, y in xy.tolist()]
    pts = xy.tolist()                       # python floats in one C pass
    out = []
    new = KeyPoint.__new__
    for x, y in pts:
        k = new(KeyPoint)
        k._pt = (x, y)
        out.append(k)
    return out


_pt_of = None


def xy_from_keypoints(kps):
    """list of KeyPoint -> [N,2] float32, rebuilt from `.pt` on EVERY call like the reference does
    (features_utils.py:65-77): the list is caller-owned and may have been edited in place
    (kps[i] = ..., sort(), a new .pt), so nothing about it is cached.  One C-level pass
    (np.fromiter over a chained attrgetter), 0.26 ms for 2048 keypoints."""
    global _pt_of
    import itertools
    import operator
    if _pt_of is None:
        _pt_of = operator.attrgetter("pt")
    n = len(kps)
    if n == 0:
        return _np.empty((0, 2), _np.float32)
    if _kp_convert is not None:
        try:                                # cv2's own pass over its own class (anything else in the list: the generic pass below)
            xy = _np.asarray(_kp_convert(kps), _np.float32)
            if xy.shape == (n, 2):
                return xy
        except Exception:                   # noqa: BLE001
            pass
    return _np.fromiter(itertools.chain.from_iterable(map(_pt_of, kps)), _np.float32, 2 * n).reshape(n, 2)


def matches_from_ij(ij):
    """[K,2] int array -> list of DMatch(queryIdx, trainIdx, 0, 0.0) (features_utils.py:80-83)."""
    if len(ij) == 0:
        return []
    q, t = _np.asarray(ij).T.tolist()
    return list(map(DMatch, q, t, _repeat(0), _repeat(0.0)))


_set = setattr


def bind_matches(shells, src, ij):
    """The objects `match_shells` prepared while the GPU was matching -> the K matches of the [K,2] pair array that has arrived:
    surplus objects dropped, missing ones made, the indices bound (the duck type: lazily, through `src`; cv2's class: two
    attribute stores per match in C-level passes)."""
    k = len(ij)
    if k < len(shells):
        del shells[k:]
    if HAVE_CV2:
        if k:
            q, t = _np.asarray(ij).T.tolist()
            n = len(shells)
            _deque(map(_set, shells, _repeat("queryIdx"), q), maxlen=0)          # (map stops at the shorter: the prepared ones)
            _deque(map(_set, shells, _repeat("trainIdx"), t), maxlen=0)
            if k > n:                                                            # more matches than had been prepared
                shells.extend(map(DMatch, q[n:], t[n:], _repeat(0), _repeat(0.0)))
        return shells
    src.ij = ij
    if k > len(shells):
        shells.extend(match_shells(k, src, len(shells))[0])
    return shells


class KeyPointList(list):
    """The list `feature_extractor` returns: an ordinary list of KeyPoint that also remembers the float32 [N,2]
    array it was built from and whether it has been edited since (every list mutator sets `_dirty`), so that
    `feature_matcher` can use the copy of the keypoints that is still on the GPU instead of rebuilding and
    re-uploading them.  A copy (`list(kps)`), a slice or an edited list is an ordinary / dirty list and takes
    the rebuilding path.  (An in-place edit of an ELEMENT - `kps[i].pt = ...` - cannot be seen by the list: with this module's
    duck type every `pt` assignment anywhere moves a counter the list compares with its own; with cv2's class EVERY element is
    read back through cv2.KeyPoint_convert and compared with the remembered array.)"""
    __slots__ = ("_xy", "_dirty", "_epoch")

    def __init__(self, items=(), xy=None):
        super().__init__(items)
        self._xy = xy
        self._dirty = xy is None
        self._epoch = keypoint_edit_epoch()

    def _touch(self):
        self._dirty = True

    def __setitem__(self, i, v): self._touch(); super().__setitem__(i, v)
    def __delitem__(self, i): self._touch(); super().__delitem__(i)
    def __iadd__(self, o): self._touch(); return super().__iadd__(o)
    def __imul__(self, o): self._touch(); return super().__imul__(o)
    def append(self, v): self._touch(); super().append(v)
    def extend(self, v): self._touch(); super().extend(v)
    def insert(self, i, v): self._touch(); super().insert(i, v)
    def pop(self, *a): self._touch(); return super().pop(*a)
    def remove(self, v): self._touch(); super().remove(v)
    def reverse(self): self._touch(); super().reverse()
    def sort(self, **kw): self._touch(); super().sort(**kw)
    def clear(self): self._touch(); super().clear()

    def pristine_xy(self):
        """The array this list was built from if it is provably still what the list holds, else None."""
        if self._dirty or self._xy is None or len(self) != len(self._xy):
            return None
        if self._epoch is not None:                        # duck type: no `pt` has been assigned anywhere since the list was made
            if self._epoch == KeyPoint.edits:
                return self._xy
            self._dirty = True
            return None
        # cv2's class cannot be watched: read EVERY keypoint back (cv2.KeyPoint_convert: one C pass, microseconds) - an element
        # edited in place, or replaced by something that is not a cv2.KeyPoint, makes the list an ordinary one
        try:
            cur = xy_from_keypoints(self)
        except Exception:                                  # noqa: BLE001
            cur = None
        if cur is None or cur.shape != self._xy.shape or not _np.array_equal(cur, self._xy):
            self._dirty = True
            return None
        return self._xy


_q_of, _t_of = _attrgetter("queryIdx"), _attrgetter("trainIdx")


class MatchList(list):
    """The list `feature_matcher` returns on the device-resident path: an ordinary list of DMatch that also remembers the
    int32 [K,2] (queryIdx, trainIdx) array it was built from and whether it has been edited since (every list mutator sets
    `_dirty`), so that `filter_matches_ransac` may apply the inlier mask the device computed for exactly these pairs in
    exactly this order.  A sorted / filtered / copied list is an ordinary or dirty list and is filtered from scratch.  (An
    in-place edit of an ELEMENT - `m.queryIdx = ...` - cannot be seen by the list: the duck type's edit counter is compared,
    see dmatch_edit_epoch; with cv2's class every element's indices are read back and compared.)"""
    __slots__ = ("_ij", "_dirty", "_epoch", "_qt")

    def __init__(self, items=(), ij=None):
        super().__init__(items)
        self._ij = ij
        self._dirty = ij is None
        self._epoch = dmatch_edit_epoch()
        self._qt = None                 # cv2's class: the two index columns as python lists (made on the first check)

    def _touch(self):
        self._dirty = True

    def __setitem__(self, i, v): self._touch(); super().__setitem__(i, v)
    def __delitem__(self, i): self._touch(); super().__delitem__(i)
    def __iadd__(self, o): self._touch(); return super().__iadd__(o)
    def __imul__(self, o): self._touch(); return super().__imul__(o)
    def append(self, v): self._touch(); super().append(v)
    def extend(self, v): self._touch(); super().extend(v)
    def insert(self, i, v): self._touch(); super().insert(i, v)
    def pop(self, *a): self._touch(); return super().pop(*a)
    def remove(self, v): self._touch(); super().remove(v)
    def reverse(self): self._touch(); super().reverse()
    def sort(self, **kw): self._touch(); super().sort(**kw)
    def clear(self): self._touch(); super().clear()

    def pristine_ij(self):
        """The pair array this list was built from if it is provably still what the list holds, else None."""
        if self._dirty or self._ij is None or len(self) != len(self._ij):
            return None
        if self._epoch is not None:                        # duck type: no index has been assigned anywhere since the list was made
            if self._epoch == DMatch.edits:
                return self._ij
            self._dirty = True
            return None
        # cv2's class cannot be watched: read EVERY match's two indices back (two C-level attribute passes)
        if len(self):
            try:
                if self._qt is None:
                    self._qt = _np.asarray(self._ij).T.tolist()
                same = list(map(_q_of, self)) == self._qt[0] and list(map(_t_of, self)) == self._qt[1]
            except Exception:                              # noqa: BLE001
                same = False
            if not same:
                self._dirty = True
                return None
        return self._ij
