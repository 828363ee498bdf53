"""Device-side timing of the batched LightGlue entry (B pairs per enqueue, HIP events).
usage: time_lightglue_batch.py [N=2048] [B=4] [iters=10] [precision=2]   (2 = "f16x3p1", the default of an instance since r05; 1 = "f16x3"; 0 = "f32")"""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
ctx = pkg._native.default_context()
lg = LG(W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0), max_kpts=N, max_pairs=B)
lg.set_precision(prec)
import os
if os.environ.get("SSLAM_BIG_GEMM"):
    lg.debug_big_gemm(int(os.environ["SSLAM_BIG_GEMM"]))
if os.environ.get("SSLAM_KEY_SPLIT"):          # 1 = the 4-wave attention kernel without key split (A/B against the ping-pong form)
    lg.debug_key_split(int(os.environ["SSLAM_KEY_SPLIT"]))
if os.environ.get("SSLAM_GRAPHS"):             # replay the launch sequence as a cached hipGraph (what the pipeline and the drop-in path do)
    lg.use_graphs(bool(int(os.environ["SSLAM_GRAPHS"])))
pairs = []
for b in range(B):
    k0, d0, k1, d1 = lg_inputs.make_pair(N, seed=11 + b)
    a = [ctx.upload(v) for v in (k0, d0, k1, d1)]
    pairs.append((a[0], a[1], N, a[2], a[3], N))
ij = ctx.malloc(B * N * 8); sc = ctx.malloc(B * N * 4); info = ctx.malloc(B * 16)
for _ in range(2):
    lg.match_batch_dev(pairs, ij, sc, info, N)
ctx.sync()
ctx.timer_start()
for _ in range(iters):
    lg.match_batch_dev(pairs, ij, sc, info, N)
ms = ctx.timer_stop() / iters
inf = np.empty((B, 4), np.int32); ctx.d2h(inf, info)
gf = (4*N*128*256 + 9*(2*(6*N*256*256+4*N*N*256+2*N*256*256+8*N*256*256+4*N*256*256) + 2*(4*N*256*256+4*N*N*256+2*N*256*256+8*N*256*256+4*N*256*256)) + 4*N*256*256+2*N*N*256)/1e9
lg.profile(True)
for _ in range(2):
    lg.match_batch_dev(pairs, ij, sc, info, N)
ctx.sync(); lg.profile(False)
ams, an = lg.profile_read()
print(f"attention: {an} launches, {ams/max(an,1)*1e3:.1f} us avg per launch of {B} pairs = {ams/max(an,1)*1e3/B:.1f} us per pair")
print(f"N={N} B={B} prec={prec}: batch {ms:.3f} ms = {ms/B:.3f} ms/pair -> {1000*B/ms:.1f} pairs/s  {gf*B/ms:.1f} TFLOP/s algorithmic  info={inf[0]}")
