import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
sd = W.random_aliked_state_dict(0)
al = AL(sd, max_num_keypoints=2048, max_h=400, max_w=1300)
xy, desc, sc = al.extract(frames.noise_frame(0), 2048, return_scores=True)
d = al.debug_read(2, (8,), np.int32)
h, w = int(d[0]), int(d[1])
np.savez(ROOT / "gpurun_out" / "aliked_dump.npz", xy=xy, desc=desc, sc=sc, dims=d,
         score=al.debug_read(0, (h, w)), nms=al.debug_read(8, (h, w)), idx=al.debug_read(1, (len(xy),), np.int32))
print("dumped", d)
