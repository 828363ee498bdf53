"""Is the frame pipeline host-bound?  Time the enqueue (pipe.round returning) against the GPU wall time."""
import importlib, os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
AlikedHIP = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
LightGlueHIP = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
NE, NM, B = int(os.environ.get("NE", 2)), int(os.environ.get("NM", 6)), int(os.environ.get("B", 24))
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    se = [torch.cuda.Stream() for _ in range(NE)]; sm = [torch.cuda.Stream() for _ in range(NM)]
    ce = [pkg._native.Context(0, stream=s.cuda_stream) for s in se]
    cm = [pkg._native.Context(0, stream=s.cuda_stream) for s in sm]
    sd_a, sd_l = W.random_aliked_state_dict(0), W.random_lightglue_state_dict(0)
    dets = [AlikedHIP(sd_a, max_num_keypoints=2048, max_h=376, max_w=1241, ctx=c) for c in ce]
    mats = [LightGlueHIP(sd_l, max_kpts=2048, ctx=c) for c in cm]
    plan = fs.ShardPlan(1, 0, B)
    pipe = fs.FrameStreamPipeline(dets, mats, plan, 2048, 0.7, streams_e=se, streams_m=sm)
    pool = [torch.from_numpy(np.stack([bench.noise_frame(f) for f in plan.frames(r)])).cuda() for r in range(2)]
    for i in range(3):
        pipe.round(pool[i % 2], 376, 1241, 3)
    torch.cuda.synchronize()
    enq, tot = [], []
    for i in range(6):
        t0 = time.perf_counter()
        pipe.round(pool[i % 2], 376, 1241, 3)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"B={B} NE={NE} NM={NM}: enqueue {np.median(enq)*1e3:.2f} ms, total {np.median(tot)*1e3:.2f} ms per round "
          f"({B/np.median(tot):.1f} frames/s); enqueue per frame {np.median(enq)/B*1e6:.0f} us")
    # back-to-back rounds without sync (as the bench does)
    t0 = time.perf_counter()
    for i in range(8):
        pipe.round(pool[i % 2], 376, 1241, 3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"8 rounds back-to-back: enqueue done at {1e3*(t1-t0):.1f} ms, GPU done at {1e3*(t2-t0):.1f} ms -> {8*B/(t2-t0):.1f} frames/s")
