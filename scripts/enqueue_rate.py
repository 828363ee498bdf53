"""Host cost of enqueueing one pipeline round (bench configuration) vs the round's wall time."""
import importlib, os, sys, time
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
B, P, K = bench.FRAMES_PER_RANK, bench.BATCH_PAIRS, bench.MAX_KPTS
dets = [AL(W.random_aliked_state_dict(0), max_num_keypoints=K, max_h=376, max_w=1241, ctx=nat.Context(0)) for _ in range(2)]
mats = [LG(W.random_lightglue_state_dict(0), max_kpts=K, ctx=nat.Context(0), max_pairs=P) for _ in range(2)]
pipe = fs.FrameStreamPipeline(dets, mats, fs.ShardPlan(1, 0, B), K, 0.7, batch_pairs=P)
pool = [pipe.ctx.upload(np.stack([bench.noise_frame(f) for f in range(r * B, (r + 1) * B)])) for r in range(2)]
for i in range(3):
    pipe.round(pool[i % 2], 376, 1241, 3)
pipe.sync()
R = 10
t0 = time.perf_counter(); host = 0.0
for i in range(R):
    h0 = time.perf_counter(); pipe.round(pool[i % 2], 376, 1241, 3); host += time.perf_counter() - h0
pipe.sync()
wall = time.perf_counter() - t0
print(f"round wall {wall / R * 1e3:.2f} ms, host enqueue {host / R * 1e3:.2f} ms per round = {host / wall * 100:.1f} % of wall; "
      f"{R * B / wall:.1f} frames/s")
# enqueue-only cost with an idle GPU queue: sync before each round
host = 0.0
for i in range(R):
    pipe.sync(); h0 = time.perf_counter(); pipe.round(pool[i % 2], 376, 1241, 3); host += time.perf_counter() - h0
pipe.sync()
print(f"host enqueue with an empty queue: {host / R * 1e3:.2f} ms per round")
