"""Device-side timing of ALIKED extraction alone (one stream): the single-frame entry, or with F > 1 the batched
entry (F frames per launch sequence).  usage: time_aliked.py [iters=20] [F=1] [graphs=0]"""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1
graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
nat = pkg._native
ctx = nat.default_context()
K = 2048
al = AL(W.random_aliked_state_dict(0), max_num_keypoints=K, max_h=376, max_w=1241, max_frames=F)
al.use_graphs(bool(graphs))
d_img = [ctx.upload(frames.noise_frame(i) if i % 2 == 0 else frames.structured_frame(i)) for i in range(F)]
xy = [ctx.malloc(K * 8) for _ in range(F)]; desc = [ctx.malloc(K * 512) for _ in range(F)]
sc = [ctx.malloc(K * 4) for _ in range(F)]; n = [ctx.malloc(64) for _ in range(F)]
def run():
    if F == 1:
        al.extract_dev(d_img[0], 376, 1241, 3, xy[0], desc[0], sc[0], n[0])
    else:
        al.extract_batch_dev(d_img, 376, 1241, 3, xy, desc, sc, n)
for _ in range(3):
    run()
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    run()
ms = ctx.timer_stop() / iters
print(f"ALIKED 1241x376 -> 2048 kpts, {F} frame(s) per launch sequence, graphs {graphs}: {ms:.3f} ms per call = {ms / F:.3f} ms/frame ({1000 * F / ms:.0f} frames/s)")
