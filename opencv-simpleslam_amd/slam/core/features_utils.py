"""Feature extraction + matching entry points backed by the HIP ALIKED /
LightGlue kernels.

Drop-in for the LightGlue path of the reference's slam/core/features_utils.py:
same names, arguments, return types and error behaviour

    init_feature_pipeline(args)                      features_utils.py:18
    feature_extractor(args, img, detector)           features_utils.py:85
    feature_matcher(args, kp0, kp1, des0, des1, m)   features_utils.py:109
    filter_matches_ransac(kp1, kp2, matches, thresh) features_utils.py:185

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

from .types import DMatch, KeyPoint, HAVE_CV2, keypoints_from_xy, matches_from_ij, xy_from_keypoints
from ... import _native, weights as _weights
from ...aliked import AlikedHIP
from ...lightglue import LightGlueHIP

_log = logging.getLogger("opencv_simpleslam_amd")

# Optional upstream checkpoints (no network here): set these to .pth paths
ENV_ALIKED = "SSLAM_ALIKED_WEIGHTS"          # aliked-n16.pth
ENV_LIGHTGLUE = "SSLAM_LIGHTGLUE_WEIGHTS"    # aliked_lightglue.pth
MAX_IMAGE_H = int(os.environ.get("SSLAM_MAX_IMAGE_H", 2160))
MAX_IMAGE_W = int(os.environ.get("SSLAM_MAX_IMAGE_W", 4096))


ENV_ALLOW_RANDOM = "SSLAM_ALLOW_RANDOM_WEIGHTS"


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
    return random_fn(0)


def init_feature_pipeline(args):
    """Instantiate detector & matcher according to CLI arguments -> (detector, matcher)."""
    if args.use_lightglue:
        max_kpts = int(getattr(args, "max_features", 4000))
        ctx = _native.default_context(int(os.environ.get("LOCAL_RANK", 0)) % max(1, _native.device_count()))
        detector = AlikedHIP(_state_dict(ENV_ALIKED, _weights.random_aliked_state_dict, "ALIKED (aliked-n16)"),
                             max_num_keypoints=max_kpts, max_h=MAX_IMAGE_H, max_w=MAX_IMAGE_W, ctx=ctx)
        matcher = LightGlueHIP(_state_dict(ENV_LIGHTGLUE, _weights.random_lightglue_state_dict, "LightGlue (aliked_lightglue)"),
                               max_kpts=max_kpts, ctx=ctx)
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


def feature_extractor(args, img: np.ndarray, detector):
    """One image -> (list[KeyPoint], descriptors [N,128] float32 unit rows)."""
    if args.use_lightglue:
        xy, desc = detector.extract(img)        # includes the reference's second L2 normalisation (:100)
        return _convert_lg_kps_to_opencv(xy), desc
    kp0, des0 = detector.detectAndCompute(img, None)
    if des0 is None:
        return [], []
    return kp0, des0


def _as_numpy_f32(x):
    if hasattr(x, "detach"):                    # torch tensor (the reference accepts both)
        x = x.detach().to("cpu").numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def feature_matcher(args, kp0, kp1, des0, des1, matcher):
    """Two frames -> list[DMatch] (LightGlue: ascending queryIdx, score > args.min_conf)."""
    if (des0 is None or des1 is None or kp0 is None or kp1 is None
            or len(kp0) == 0 or len(kp1) == 0 or len(des0) == 0 or len(des1) == 0):
        return []
    if args.use_lightglue:
        thr = float(getattr(args, "min_conf", 0.7))
        ij, _scores, _stop = matcher.match(_convert_opencv_to_lg_kps(kp0), _as_numpy_f32(des0),
                                           _convert_opencv_to_lg_kps(kp1), _as_numpy_f32(des1), min_conf=thr)
        return _convert_lg_matches_to_opencv(ij)
    matches = matcher.match(des0, des1)
    return sorted(matches, key=lambda m: m.distance)


def filter_matches_ransac(kp1, kp2, matches, thresh=1.0):
    """Drop outliers with fundamental-matrix RANSAC, as the reference does at
    features_utils.py:185-200 - scored on the GPU (`sslam_fmat_ransac_host`: OpenCV's classic
    7-point RANSAC / LMedS restated, no cv2 needed)."""
    if len(matches) < 8:
        return matches
    from ... import epipolar
    pts1 = np.float32([kp1[m.queryIdx].pt for m in matches])
    pts2 = np.float32([kp2[m.trainIdx].pt for m in matches])
    _, mask, _ = epipolar.find_fundamental_ransac(pts1, pts2, thresh, 0.99)
    if mask is None:
        return []
    return [m for m, ok in zip(matches, mask) if ok]
