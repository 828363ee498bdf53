"""Two LightGlue instances on two HIP streams: does the chip overlap two latency-bound chains?"""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
nat = pkg._native
sd = W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0)
k0, d0, k1, d1 = lg_inputs.make_pair(N, seed=11)
ctxs = [nat.Context(0) for _ in range(NS)]
lgs = [LG(sd, max_kpts=N, ctx=c) for c in ctxs]
bufs = []
for c in ctxs:
    bufs.append(tuple(c.upload(a) for a in (k0, d0, k1, d1)) + (c.malloc(N * 8), c.malloc(N * 4), c.malloc(32)))
def run(n):
    for _ in range(n):
        for lg, b in zip(lgs, bufs):
            lg.match_dev(b[0], b[1], N, b[2], b[3], N, b[4], b[5], b[6])
    for c in ctxs: c.sync()
run(3)
t0 = time.perf_counter(); run(iters); dt = time.perf_counter() - t0
print(f"N={N} streams={NS}: {dt/iters/NS*1e3:.3f} ms per pair  ({iters*NS/dt:.1f} pairs/s)")
