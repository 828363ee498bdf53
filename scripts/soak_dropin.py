"""Soak of the drop-in frame loop (look-ahead match, device ring, page-locked records): N frames as main_revamped.py
drives them; every 25th pair is re-matched through the host path and must give the same matches."""
import importlib, os, sys, time
os.environ.setdefault("SSLAM_ALLOW_RANDOM_WEIGHTS", "1")
from pathlib import Path
from types import SimpleNamespace
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fu = importlib.import_module("opencv-simpleslam_amd.slam.core.features_utils")
W = importlib.import_module("opencv-simpleslam_amd.weights")
sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)          # weights that produce matches
fu._weights.random_lightglue_state_dict = lambda seed=0: sd
args = SimpleNamespace(use_lightglue=True, max_features=2048, min_conf=0.05, detector="aliked", matcher="lightglue")
det, mat = fu.init_feature_pipeline(args)
imgs = [frames.structured_frame(i) for i in range(16)]
pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
kp_prev, des_prev = fu.feature_extractor(args, imgs[0], det)
t0 = time.perf_counter(); checked = 0; total = 0
for i in range(1, N):
    kp, des = fu.feature_extractor(args, imgs[i % 16], det)
    m = fu.feature_matcher(args, kp_prev, kp, des_prev, des, mat)
    total += len(m)
    if i % 25 == 0:
        ref = fu.feature_matcher(args, list(kp_prev), list(kp), des_prev.copy(), des.copy(), mat)
        assert pairs(m) == pairs(ref), f"frame {i}: look-ahead / resident matches differ from the host path"
        checked += 1
    if i % 97 == 0:                                   # break the pattern now and then (keyframe -> cur, extractor only)
        fu.feature_extractor(args, imgs[(i + 5) % 16], det)
    kp_prev, des_prev = kp, des
dt = time.perf_counter() - t0
print(f"{N} frames in {dt:.1f} s ({N / dt:.0f} frames/s incl. the checks), {total} matches, {checked} pairs re-checked through the host path: all equal")
