"""Host time of FrameStreamPipeline.round() (enqueue only) with and without the collation path (RCCL, one rank)."""
import importlib, os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames
dist_mode = len(sys.argv) > 1 and sys.argv[1] == "dist"
if dist_mode:
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pkg = importlib.import_module("opencv-simpleslam_amd")
nat = pkg._native
W = importlib.import_module("opencv-simpleslam_amd.weights")
fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
K, B = 2048, 24
dets = [AL(W.random_aliked_state_dict(0), max_num_keypoints=K, max_h=376, max_w=1241, ctx=nat.Context(0), max_frames=8)]
mats = [LG(W.random_lightglue_state_dict(0), max_kpts=K, ctx=nat.Context(0), max_pairs=8) for _ in range(3)]
pipe = fs.FrameStreamPipeline(dets, mats, fs.ShardPlan(1, 0, B), K, 0.7, batch_pairs=8, collate_always=dist_mode)
pool = [dets[0].ctx.upload(np.stack([frames.noise_frame(24 * r + f) for f in range(B)])) for r in range(2)]
for i in range(6):
    pipe.round(pool[i % 2], 376, 1241, 3)
pipe.sync()
ts = []
t_all = time.perf_counter()
for i in range(40):
    t0 = time.perf_counter(); pipe.round(pool[i % 2], 376, 1241, 3); ts.append(time.perf_counter() - t0)
pipe.sync()
wall = time.perf_counter() - t_all
print(f"{'dist' if dist_mode else 'plain'}: host time per round() median {np.median(ts)*1e3:.2f} ms, p90 {np.percentile(ts,90)*1e3:.2f}, max {max(ts)*1e3:.2f}; "
      f"40 rounds in {wall*1e3:.0f} ms = {40*B/wall:.0f} frames/s")
if dist_mode:
    dist.destroy_process_group()
