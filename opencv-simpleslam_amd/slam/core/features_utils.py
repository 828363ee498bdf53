"""Feature extraction + matching entry points backed by the HIP ALIKED /
LightGlue kernels.

Drop-in for the LightGlue path of the reference's slam/core/features_utils.py:
same names, arguments, return types and error behaviour

    init_feature_pipeline(args)                      features_utils.py:18
    feature_extractor(args, img, detector)           features_utils.py:85
    feature_matcher(args, kp0, kp1, des0, des1, m)   features_utils.py:109
    filter_matches_ransac(kp1, kp2, matches, thresh) features_utils.py:185
    detect_and_match(img1, img2, det, m, args)       features_utils.py:250   (legacy pair entry)

`args` is the CLI namespace of slam/monocular/main_revamped.py (fields read:
use_lightglue, max_features, min_conf).  Keypoints are `cv2.KeyPoint` when
OpenCV is importable, otherwise the duck type in .types (only `.pt` is read
downstream); matches likewise `cv2.DMatch` / duck type.

The OpenCV ORB/SIFT/AKAZE + BF/FLANN branch of the reference (cv2 C++, config 1
"plumbing") is delegated to cv2 when it is installed and raises otherwise: this
package accelerates the LightGlue path only.  There is no CPU fallback for the
LightGlue path: without the HIP library / an MI355X it raises.
"""
from __future__ import annotations

import logging
import os
from typing import List

import numpy as np

import ctypes as _C
import weakref

from .types import DMatch, KeyPoint, HAVE_CV2, keypoints_from_xy, matches_from_ij, xy_from_keypoints
from ... import _native, weights as _weights
from ...aliked import AlikedHIP
from ...feature_ring import DeviceFeatureRing
from ...lightglue import LightGlueHIP

_log = logging.getLogger("opencv_simpleslam_amd")

# Optional upstream checkpoints (no network here): set these to .pth paths
ENV_ALIKED = "SSLAM_ALIKED_WEIGHTS"          # aliked-n16.pth
ENV_LIGHTGLUE = "SSLAM_LIGHTGLUE_WEIGHTS"    # aliked_lightglue.pth
MAX_IMAGE_H = int(os.environ.get("SSLAM_MAX_IMAGE_H", 2160))
MAX_IMAGE_W = int(os.environ.get("SSLAM_MAX_IMAGE_W", 4096))


ENV_ALLOW_RANDOM = "SSLAM_ALLOW_RANDOM_WEIGHTS"
ENV_RANDOM_LG_ARGS = "SSLAM_RANDOM_LIGHTGLUE_ARGS"   # e.g. "seed=1,match_gain=4.0,match_bias=3.0": arguments of the random fallback


def _state_dict(env_name, random_fn, what):
    """Checkpoint named by `env_name`, else - ONLY when SSLAM_ALLOW_RANDOM_WEIGHTS=1 - seeded random
    weights of the same architecture, with a WARNING.  The reference never runs untrained: it loads
    `aliked-n16` / `aliked_lightglue` through torch.hub (features_utils.py:25-26) or fails; dropped into
    main_revamped.py a silent random fallback would produce garbage tracks without a word."""
    path = os.environ.get(env_name)
    if path:
        return _weights.load_state_dict(path)
    if os.environ.get(ENV_ALLOW_RANDOM, "") != "1":
        raise RuntimeError(
            f"init_feature_pipeline: no {what} checkpoint - set {env_name} to the upstream .pth file (there is no "
            f"network here to download it as the reference does, features_utils.py:25-26), or set "
            f"{ENV_ALLOW_RANDOM}=1 to run seeded RANDOM weights (synthetic benchmarking / parity tests only)")
    _log.warning("%s: %s unset - running seeded RANDOM-INIT weights (%s=1): matches are meaningless on real imagery",
                 what, env_name, ENV_ALLOW_RANDOM)
    kw = {}
    if env_name == ENV_LIGHTGLUE and os.environ.get(ENV_RANDOM_LG_ARGS):
        # synthetic benchmarking only: a random init whose assignment head is sharp enough to produce matches
        for item in os.environ[ENV_RANDOM_LG_ARGS].split(","):
            k, v = item.split("=")
            kw[k.strip()] = int(v) if k.strip() == "seed" else float(v)
    return random_fn(kw.pop("seed", 0), **kw)


def init_feature_pipeline(args):
    """Instantiate detector & matcher according to CLI arguments -> (detector, matcher)."""
    if args.use_lightglue:
        max_kpts = int(getattr(args, "max_features", 4000))
        ctx = _native.default_context(int(os.environ.get("LOCAL_RANK", 0)) % max(1, _native.device_count()))
        detector = AlikedHIP(_state_dict(ENV_ALIKED, _weights.random_aliked_state_dict, "ALIKED (aliked-n16)"),
                             max_num_keypoints=max_kpts, max_h=MAX_IMAGE_H, max_w=MAX_IMAGE_W, ctx=ctx)
        # the matcher runs on a stream of its own (same device): it reads the extractor's device records directly, ordered
        # behind the extraction by an event, so a match can run while a frame's results are still on their way to the host
        mctx = _native.Context(ctx.device)
        # (two pairs per launch: prev -> cur and keyframe -> cur of one frame go out together, feature_ring.py)
        matcher = LightGlueHIP(_state_dict(ENV_LIGHTGLUE, _weights.random_lightglue_state_dict, "LightGlue (aliked_lightglue)"),
                               max_kpts=max_kpts, ctx=mctx, max_pairs=DeviceFeatureRing.PAIRS)
        matcher._feature_ring = _ring_of(detector)
        matcher._feature_ring.attach_matcher(matcher)
        return detector, matcher
    if not HAVE_CV2:
        raise ImportError("the OpenCV detector/matcher branch needs cv2; this backend accelerates "
                          "the --use_lightglue path only")
    import cv2
    det = getattr(args, "detector", "orb")
    nfeat = int(getattr(args, "max_features", 6000))
    if det == "orb":
        detector = cv2.ORB_create(nfeat)
    elif det == "sift":
        detector = cv2.SIFT_create(nfeatures=nfeat)
    elif det == "akaze":
        detector = cv2.AKAZE_create()
    else:
        raise ValueError(f"Unsupported detector: {det}")
    if getattr(args, "matcher", "bf") == "flann":
        matcher = cv2.FlannBasedMatcher(dict(algorithm=1, trees=5), dict(checks=50))
    else:
        norm = cv2.NORM_HAMMING if det in ("orb", "akaze") else cv2.NORM_L2
        matcher = cv2.BFMatcher(norm, crossCheck=True)
    return detector, matcher


def _convert_lg_kps_to_opencv(xy: np.ndarray) -> List[KeyPoint]:
    return keypoints_from_xy(np.asarray(xy))


def _convert_opencv_to_lg_kps(kps) -> np.ndarray:
    return xy_from_keypoints(kps)


def _convert_lg_matches_to_opencv(ij: np.ndarray) -> List[DMatch]:
    return matches_from_ij(np.asarray(ij))


def _ring_of(detector):
    """The detector's device-resident frame store (feature_ring.DeviceFeatureRing: residency by use, memo, look-ahead)."""
    ring = getattr(detector, "_feature_ring", None)
    if ring is None:
        ring = detector._feature_ring = DeviceFeatureRing(detector)
    return ring


def feature_extractor(args, img: np.ndarray, detector):
    """One image -> (list[KeyPoint], descriptors [N,128] float32 unit rows)."""
    if args.use_lightglue:
        # includes the reference's second L2 normalisation (:100); the features also stay on the GPU for the matcher
        return _ring_of(detector).extract(img)
    kp0, des0 = detector.detectAndCompute(img, None)
    if des0 is None:
        return [], []
    return kp0, des0


def _as_numpy_f32(x):
    if hasattr(x, "detach"):                    # torch tensor (the reference accepts both)
        x = x.detach().to("cpu").numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


_last_ring = None        # weak reference to the ring of the last resident match (filter_matches_ransac has no matcher argument)


def feature_matcher(args, kp0, kp1, des0, des1, matcher):
    """Two frames -> list[DMatch] (LightGlue: ascending queryIdx, score > args.min_conf)."""
    if (des0 is None or des1 is None or kp0 is None or kp1 is None
            or len(kp0) == 0 or len(kp1) == 0 or len(des0) == 0 or len(des1) == 0):
        return []
    if args.use_lightglue:
        thr = float(getattr(args, "min_conf", 0.7))
        ring = getattr(matcher, "_feature_ring", None)
        if ring is not None and isinstance(des0, np.ndarray) and isinstance(des1, np.ndarray):
            a = ring.lookup(des0, kp0, 0)
            b = ring.lookup(des1, kp1, 1, keep=a[5]) if a is not None else None
            if a is not None and b is not None:
                # both frames are `feature_extractor`'s own and (still, or again) on the GPU
                global _last_ring
                _last_ring = weakref.ref(ring)
                return ring.match(a, b, thr)
        ij, _scores, _stop = matcher.match(_convert_opencv_to_lg_kps(kp0), _as_numpy_f32(des0),
                                           _convert_opencv_to_lg_kps(kp1), _as_numpy_f32(des1), min_conf=thr)
        return _convert_lg_matches_to_opencv(ij)
    matches = matcher.match(des0, des1)
    return sorted(matches, key=lambda m: m.distance)


def _lightglue_detect_and_match(img1, img2, extractor, matcher):
    """Legacy pair entry (reference features_utils.py:233-247; no longer on its main path, still what its own
    tests/test_lightglue_vs_manual.py drives): extract both images, match the feature dicts as they come out of
    `extractor.extract` - they carry 'image_size', so LightGlue normalises the keypoints by the IMAGE size, not by
    their bounding box - and return every match LightGlue keeps (filter_threshold only, no min_conf cut)."""
    ring = _ring_of(extractor)
    kp0, des0 = ring.extract(img1)
    kp1, des1 = ring.extract(img2)
    if len(kp0) == 0 or len(kp1) == 0:
        return kp0, kp1, des0, des1, []
    h0, w0 = np.asarray(img1).shape[:2]
    h1, w1 = np.asarray(img2).shape[:2]
    ij, _scores, _stop = matcher.match(kp0._xy, des0, kp1._xy, des1, min_conf=0.0, size0=(w0, h0), size1=(w1, h1))
    return kp0, kp1, des0, des1, _convert_lg_matches_to_opencv(ij)


def _opencv_detect_and_match(img1, img2, detector, matcher):
    kp1, des1 = detector.detectAndCompute(img1, None)
    kp2, des2 = detector.detectAndCompute(img2, None)
    if des1 is None or des2 is None:
        return [], [], [], [], []        # gracefully handle empty images
    return kp1, kp2, des1, des2, sorted(matcher.match(des1, des2), key=lambda m: m.distance)


def detect_and_match(img1, img2, detector, matcher, args):
    """Front-end entry of the reference (features_utils.py:250-256): OpenCV or LightGlue depending on the CLI flag."""
    if args.use_lightglue:
        return _lightglue_detect_and_match(img1, img2, detector, matcher)
    return _opencv_detect_and_match(img1, img2, detector, matcher)


def filter_matches_ransac(kp1, kp2, matches, thresh=1.0):
    """Drop outliers with fundamental-matrix RANSAC, as the reference does at
    features_utils.py:185-200 - scored on the GPU (`sslam_fmat_ransac_host`: OpenCV's classic
    7-point RANSAC / LMedS restated, no cv2 needed)."""
    if len(matches) < 8:
        return matches
    ring = _last_ring() if _last_ring is not None else None
    if ring is not None:
        # the list feature_matcher just returned for exactly these frames: the filter already ran on the device behind the
        # match (same matches, same pixels, same threshold) and its mask came back with them - or it will from now on
        kept = ring.filtered(kp1, kp2, matches, thresh)
        if kept is not None:
            return kept
    from ... import epipolar
    pts1 = np.float32([kp1[m.queryIdx].pt for m in matches])
    pts2 = np.float32([kp2[m.trainIdx].pt for m in matches])
    _, mask, _ = epipolar.find_fundamental_ransac(pts1, pts2, thresh, 0.99)
    if mask is None:
        return []
    return [m for m, ok in zip(matches, mask) if ok]
