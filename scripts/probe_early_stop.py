"""Which random-init weight offsets give a non-trivial early-stop / pruning histogram on the bench streams."""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
pkg = importlib.import_module("opencv-simpleslam_amd"); nat = pkg._native
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
B, P, K = 16, 8, 2048
dets = [AL(W.random_aliked_state_dict(0), max_num_keypoints=K, max_h=376, max_w=1241, ctx=nat.Context(0))]
frames = np.stack([bench.structured_frame(f) for f in range(B)])
import itertools
grid = list(itertools.product((1.0, 1.3, 1.6, 2.0, 2.5), (-9.0,), (4.0, 8.0, 16.0)))
for cb, mb, cg in grid:
    sd = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=mb, conf_bias=cb, conf_gain=cg)
    mats = [LG(sd, max_kpts=K, ctx=nat.Context(0), max_pairs=P)]
    pipe = fs.FrameStreamPipeline(dets, mats, fs.ShardPlan(1, 0, B), K, 0.7, batch_pairs=P)
    chunk = pipe.ctx.upload(frames)
    pipe.round(chunk, 376, 1241, 3); pipe.round(chunk, 376, 1241, 3)
    info = pipe.infos()
    lay, cnt = np.unique(info[info[:, 2] > 0, 1], return_counts=True)
    print(f"conf_bias {cb} match_bias {mb} conf_gain {cg}: layers {dict(zip(lay.tolist(), cnt.tolist()))}  kept kpts {info[:, 2].min()}..{info[:, 2].max()} / {info[:, 3].min()}..{info[:, 3].max()}  matches {info[:, 0].min()}..{info[:, 0].max()}", flush=True)
    pipe.ctx.free(chunk)
    for m in mats: m.close()
