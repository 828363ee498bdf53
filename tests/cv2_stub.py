"""A minimal stand-in for the `cv2` module (absent from the image), for the CPU plumbing test of the overlay's OpenCV
branch - BASELINE config C1 "ORB + BF matcher" (reference slam/core/features_utils.py:28-29, :33-55, :104-107, :177-178).
Only what that branch touches: ORB_create / SIFT_create / AKAZE_create -> detectAndCompute, BFMatcher(norm, crossCheck) /
FlannBasedMatcher -> match, KeyPoint, DMatch, the two norm constants.  The "detector" is a deterministic toy (corner-ish
pixels, 32-byte binary descriptors of their neighbourhood); it exists so that the calls have something to carry."""
import importlib.util
import subprocess
import sys
import sysconfig
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent


def build_cv2like():
    """gcc-build tests/cv2like/cv2like.c (KeyPoint / DMatch as C structs behind python objects, KeyPoint_convert in one C
    pass: the COST of the wheel's value classes, which python classes cannot stand in for) -> the loaded module."""
    src = HERE / "cv2like" / "cv2like.c"
    so = HERE / "cv2like" / "_cv2like.so"
    if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-I", sysconfig.get_paths()["include"], str(src), "-o", str(so)],
                       check=True, capture_output=True, text=True)
    spec = importlib.util.spec_from_file_location("_cv2like", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install(native_classes=False):
    """Install the stand-in as `cv2`.  native_classes: KeyPoint / DMatch / KeyPoint_convert from the C module (the LightGlue
    drop-in path with cv2's OWN classes present - the only environment main_revamped.py runs in - is tested and timed against
    these); otherwise plain python classes."""
    cv2 = types.ModuleType("cv2")
    cv2.NORM_L2, cv2.NORM_HAMMING = 4, 6
    calls = cv2._calls = []

    class KeyPoint:
        def __init__(self, x=0.0, y=0.0, size=1.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
            self.pt, self.size, self.angle, self.response, self.octave, self.class_id = (float(x), float(y)), size, angle, response, octave, class_id

    class DMatch:
        def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=0.0):
            self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(queryIdx), int(trainIdx), int(imgIdx), float(distance)

    class _Detector:
        def __init__(self, kind, nfeatures):
            self.kind, self.nfeatures = kind, nfeatures

        def detectAndCompute(self, img, mask):
            calls.append(("detectAndCompute", self.kind))
            g = img if img.ndim == 2 else img.mean(axis=2)
            g = g.astype(np.float32)
            r = np.abs(g[1:-1, 1:-1] * 4 - g[:-2, 1:-1] - g[2:, 1:-1] - g[1:-1, :-2] - g[1:-1, 2:])
            r[:8] = 0; r[-8:] = 0; r[:, :8] = 0; r[:, -8:] = 0
            flat = np.argsort(-r, axis=None, kind="stable")[:self.nfeatures]
            flat = flat[r.reshape(-1)[flat] > 0]
            if len(flat) == 0:
                return (), None                              # what cv2 returns on a blank image
            ys, xs = np.unravel_index(flat, r.shape)
            ys, xs = ys + 1, xs + 1
            kps = tuple(KeyPoint(float(x), float(y), 31.0, response=float(r[y - 1, x - 1])) for x, y in zip(xs, ys))
            rng = np.random.default_rng(0)
            off = rng.integers(-7, 8, (256, 4))
            des = np.zeros((len(kps), 32), np.uint8)
            for i, (x, y) in enumerate(zip(xs, ys)):
                bits = g[y + off[:, 0], x + off[:, 1]] < g[y + off[:, 2], x + off[:, 3]]
                des[i] = np.packbits(bits)
            return kps, des

    class BFMatcher:
        def __init__(self, normType=4, crossCheck=False):
            self.normType, self.crossCheck = normType, crossCheck
            calls.append(("BFMatcher", normType, crossCheck))

        def match(self, d0, d1):
            calls.append(("match", len(d0), len(d1)))
            if self.normType == cv2.NORM_HAMMING:
                dist = np.unpackbits(d0[:, None, :] ^ d1[None, :, :], axis=2).sum(2).astype(np.float32)
            else:
                dist = np.linalg.norm(d0[:, None, :].astype(np.float32) - d1[None, :, :].astype(np.float32), axis=2)
            j = dist.argmin(1)
            out = []
            for i, jj in enumerate(j):
                if not self.crossCheck or dist[:, jj].argmin() == i:
                    out.append(DMatch(i, int(jj), 0, float(dist[i, jj])))
            return out                                        # query order, NOT distance order

    class FlannBasedMatcher(BFMatcher):
        def __init__(self, index_params=None, search_params=None):
            calls.append(("FlannBasedMatcher", index_params, search_params))
            self.normType, self.crossCheck = cv2.NORM_L2, False

    if native_classes:
        like = build_cv2like()
        KeyPoint, DMatch = like.KeyPoint, like.DMatch

        def KeyPoint_convert(arg, *a, **kw):
            """Both overloads of cv2.KeyPoint_convert: keypoints -> [N,2] float32; points2f[, size[, response[, octave[,
            class_id]]]] -> tuple of KeyPoint (response defaults to 1, unlike the KeyPoint constructor's 0)."""
            if isinstance(arg, np.ndarray):
                order = ("size", "response", "octave", "class_id")
                vals = dict(zip(order, a)); vals.update(kw)
                pts = np.ascontiguousarray(arg, np.float32).reshape(-1, 2)
                return like._xy_to_kp(pts, float(vals.get("size", 1)), float(vals.get("response", 1)), int(vals.get("octave", 0)),
                                      int(vals.get("class_id", -1)))
            out = np.empty((len(arg), 2), np.float32)
            like._kp_to_xy(arg, out)
            return out
        cv2.KeyPoint_convert = KeyPoint_convert
    cv2.KeyPoint, cv2.DMatch, cv2.BFMatcher, cv2.FlannBasedMatcher = KeyPoint, DMatch, BFMatcher, FlannBasedMatcher
    cv2.ORB_create = lambda nfeatures=500: (calls.append(("ORB_create", nfeatures)), _Detector("orb", nfeatures))[1]
    cv2.SIFT_create = lambda nfeatures=0: (calls.append(("SIFT_create", nfeatures)), _Detector("sift", nfeatures or 500))[1]
    cv2.AKAZE_create = lambda: (calls.append(("AKAZE_create",)), _Detector("akaze", 500))[1]
    sys.modules["cv2"] = cv2
    return cv2
