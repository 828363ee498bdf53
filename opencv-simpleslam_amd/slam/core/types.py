"""KeyPoint / DMatch carriers for environments without OpenCV.

Downstream of the hot path the reference only ever reads `.pt` on keypoints
and `.queryIdx` / `.trainIdx` (and `.distance` on the OpenCV-BF path) on
matches; when `cv2` is importable its own classes are
used so the rest of the reference pipeline keeps working unchanged.
"""
from __future__ import annotations

try:                                       # pragma: no cover - cv2 absent in the build image
    import cv2 as _cv2
    KeyPoint = _cv2.KeyPoint
    DMatch = _cv2.DMatch
    HAVE_CV2 = True
except Exception:                          # noqa: BLE001
    HAVE_CV2 = False

    class KeyPoint:
        """Duck type of cv2.KeyPoint(x, y, size)."""
        __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

        def __init__(self, x=0.0, y=0.0, size=1.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
            self.pt = (float(x), float(y))
            self.size = float(size)
            self.angle = float(angle)
            self.response = float(response)
            self.octave = int(octave)
            self.class_id = int(class_id)

        def __repr__(self):
            return f"KeyPoint(pt={self.pt})"

    class DMatch:
        """Duck type of cv2.DMatch(queryIdx, trainIdx, imgIdx, distance)."""
        __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

        def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=0.0):
            self.queryIdx = int(queryIdx)
            self.trainIdx = int(trainIdx)
            self.imgIdx = int(imgIdx)
            self.distance = float(distance)

        def __repr__(self):
            return f"DMatch({self.queryIdx}->{self.trainIdx})"
