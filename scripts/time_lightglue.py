"""Quick device-side timing of one LightGlue pair (dev entry, HIP events)."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
nat = pkg._native
ctx = nat.default_context()
lg = LG(W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0), max_kpts=N)
import os
if os.environ.get("SSLAM_BIG_GEMM"):
    lg.debug_big_gemm(int(os.environ["SSLAM_BIG_GEMM"]))
if os.environ.get("SSLAM_KEY_SPLIT"):
    lg.debug_key_split(int(os.environ["SSLAM_KEY_SPLIT"]))
if os.environ.get("SSLAM_GRAPHS"):
    lg.use_graphs(True)
k0, d0, k1, d1 = lg_inputs.make_pair(N, seed=11)
dk0, dd0, dk1, dd1 = (ctx.upload(a) for a in (k0, d0, k1, d1))
ij = ctx.malloc(N * 8); sc = ctx.malloc(N * 4); info = ctx.malloc(32)
for _ in range(3):
    lg.match_dev(dk0, dd0, N, dk1, dd1, N, ij, sc, info)
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    lg.match_dev(dk0, dd0, N, dk1, dd1, N, ij, sc, info)
ms = ctx.timer_stop() / iters
inf = np.empty(4, np.int32); ctx.d2h(inf, info)
gf = (4*N*128*256 + 9*(2*(6*N*256*256+4*N*N*256+2*N*256*256+8*N*256*256+4*N*256*256) + 2*(4*N*256*256+4*N*N*256+2*N*256*256+8*N*256*256+4*N*256*256)) + 4*N*256*256+2*N*N*256)/1e9
lg.profile(True)
for _ in range(4):
    lg.match_dev(dk0, dd0, N, dk1, dd1, N, ij, sc, info)
ctx.sync(); lg.profile(False)
ams, an = lg.profile_read()
print(f"attention: {an} launches, {ams/max(an,1)*1e3:.1f} us avg")
print(f"N={N} pair {ms:.3f} ms  -> {1000/ms:.1f} pairs/s  {gf/ms:.1f} TFLOP/s algorithmic ({gf:.1f} GF)  info={inf}")
