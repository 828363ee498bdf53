"""Timeline of a drop-in frame with the look-ahead match: when does the matcher's stream finish relative to the calls?"""
import importlib, os, sys, time
os.environ.setdefault("SSLAM_ALLOW_RANDOM_WEIGHTS", "1")
from pathlib import Path
from types import SimpleNamespace
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
fu = importlib.import_module("opencv-simpleslam_amd.slam.core.features_utils")
args = SimpleNamespace(use_lightglue=True, max_features=2048, min_conf=0.7, detector="aliked", matcher="lightglue")
det, mat = fu.init_feature_pipeline(args)
ring = fu._ring_of(det)
imgs = [frames.structured_frame(i) for i in range(14)]
kp_prev, des_prev = fu.feature_extractor(args, imgs[0], det)
rows = []
real_sync = mat.ctx.sync
for im in imgs[1:]:
    t0 = time.perf_counter()
    kp, des = fu.feature_extractor(args, im, det)
    t1 = time.perf_counter()
    had = ring.ahead is not None
    marks = {}
    def sync_probe():
        marks["before_sync"] = time.perf_counter(); real_sync(); marks["after_sync"] = time.perf_counter()
    mat.ctx.sync = sync_probe
    m = fu.feature_matcher(args, kp_prev, kp, des_prev, des, mat)
    t2 = time.perf_counter()
    if "before_sync" not in marks:
        print("no sync seen: ring of matcher is ring of detector:", getattr(mat, "_feature_ring", None) is ring, "lookup a:",
              ring.lookup(des_prev, kp_prev, 0) is not None, "b:", ring.lookup(des, kp, 1) is not None, type(des_prev), len(kp_prev), len(kp))
        marks = dict(before_sync=t1, after_sync=t2)
    mat.ctx.sync = real_sync
    rows.append((had, t1 - t0, marks["before_sync"] - t1, marks["after_sync"] - marks["before_sync"], t2 - marks["after_sync"], marks["after_sync"] - t0))
    kp_prev, des_prev = kp, des
for r in rows[2:]:
    print(f"look-ahead {r[0]}: extractor {r[1]*1e3:.2f} ms | matcher: host before the wait {r[2]*1e3:.2f}, wait {r[3]*1e3:.2f}, after {r[4]*1e3:.2f} | match done {r[5]*1e3:.2f} ms after the extractor call began")
