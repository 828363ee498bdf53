"""sha1 of ALIKED outputs (keypoints, descriptors, scores, score map) on a fixed set of frames: run before and after a
kernel change that claims bit-identical results and diff the two printouts."""
import hashlib, importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
rng = np.random.default_rng(5)
cases = [("noise 1241x376", frames.noise_frame(0)), ("structured 1241x376", frames.structured_frame(3)),
         ("gray", frames.structured_frame(2, c=1)), ("1920x1080", rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)),
         ("portrait 300x500", rng.integers(0, 256, (500, 300, 3), dtype=np.uint8)), ("tiny 40x33", rng.integers(0, 256, (33, 40, 3), dtype=np.uint8))]
for sd_kw in (dict(), dict(score_gain=-0.1)):
    al = AL(W.random_aliked_state_dict(0, **sd_kw), max_num_keypoints=2048, max_h=1100, max_w=1950)
    for name, img in cases:
        xy, desc, sc = al.extract(img, 2048, return_scores=True)
        d = al.debug_read(2, (8,), np.int32)
        score = al.debug_read(0, (int(d[0]), int(d[1])))
        h = hashlib.sha1()
        for a in (xy, desc, sc, score):
            h.update(np.ascontiguousarray(a).tobytes())
        print(f"{str(sd_kw):24s} {name:22s} n={len(xy):5d} {h.hexdigest()}")
    al.close()
