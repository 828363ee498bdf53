"""Device-side timing of ALIKED extraction alone (dev entry, one stream)."""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
nat = pkg._native
ctx = nat.default_context()
K = 2048
al = AL(W.random_aliked_state_dict(0), max_num_keypoints=K, max_h=376, max_w=1241)
img = frames.noise_frame(0)
d_img = ctx.upload(img)
xy = ctx.malloc(K * 8); desc = ctx.malloc(K * 512); sc = ctx.malloc(K * 4); n = ctx.malloc(64)
for _ in range(3):
    al.extract_dev(d_img, 376, 1241, 3, xy, desc, sc, n)
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    al.extract_dev(d_img, 376, 1241, 3, xy, desc, sc, n)
ms = ctx.timer_stop() / iters
print(f"ALIKED 1241x376 -> 2048 kpts: {ms:.3f} ms/frame ({1000/ms:.0f} frames/s)")
