"""Launched by tests/test_pipeline_gpu.py under torch.distributed.run with 2 ranks (gloo, both on
GPU 0): every rank runs the frame-sharded pipeline for two rounds and checks its own frames against
the sequential host API - keypoints, and the match indices of EVERY pair including the ones that
straddle a rank / round boundary (which need the gathered features of the neighbour)."""
import importlib
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames                                                      # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
pkg = importlib.import_module("opencv-simpleslam_amd")
nat = pkg._native
W = importlib.import_module("opencv-simpleslam_amd.weights")
fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
K, B, H, Wd, ROUNDS = 384, 3, 160, 256, 2
sd_a = W.random_aliked_state_dict(0)
sd_l = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
n_frames = ROUNDS * world * B
imgs = [frames.structured_frame(i, h=H, w=Wd) for i in range(n_frames)]
ctx0 = nat.default_context(0)
det0 = AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=ctx0)
mat0 = LG(sd_l, max_kpts=K, ctx=ctx0, filter_threshold=0.0)     # every mutual arg-max: non-vacuous
feats = [det0.extract(im, K) for im in imgs]
ref = [None] + [mat0.match(feats[i - 1][0], feats[i - 1][1], feats[i][0], feats[i][1], min_conf=0.0)
                for i in range(1, n_frames)]
plan = fs.ShardPlan(world, rank, B)
checked = 0
dets = [AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=nat.Context(0)) for _ in range(2)]
mats = [LG(sd_l, max_kpts=K, ctx=nat.Context(0), max_pairs=2, filter_threshold=0.0) for _ in range(2)]
pipe = fs.FrameStreamPipeline(dets, mats, plan, K, 0.0, batch_pairs=2)
for rnd in range(ROUNDS):
    mine = list(plan.frames(rnd))
    chunk = pipe.ctx.upload(np.stack([imgs[f] for f in mine]))
    pipe.round(chunk, H, Wd, 3)
    res = pipe.results()
    got = pipe.features()
    for s, f in enumerate(mine):
        assert len(got[s][0]) == len(feats[f][0]), (rank, f)
        np.testing.assert_array_equal(got[s][0], feats[f][0])
        if f == 0:
            continue
        np.testing.assert_array_equal(res[s][0], ref[f][0], err_msg=f"rank {rank} frame {f}")
        checked += 1
    # the collated map holds every rank's features of this round, in global frame order
    torch.cuda.synchronize()
    sm_all = pipe.shared_map.cpu().numpy()
    for j in range(world * B):
        f = rnd * world * B + j
        n, xy, desc = fs.unpack_record(sm_all[j], K)
        assert n == len(feats[f][0])
        np.testing.assert_array_equal(xy, feats[f][0])
        np.testing.assert_array_equal(desc, feats[f][1])
assert sum(len(r[0]) for r in ref[1:]) > 5, "vacuous: the sequential reference found no matches"
dist.barrier()
print(f"rank {rank}: {checked} pairs identical to the sequential API", flush=True)
dist.destroy_process_group()
