"""Repeatability stress of the batched LightGlue forward: the same 8-pair batch N times (plain and graph replay), every
result compared bit for bit with the first - a race in the counted-wait streaming of the fused FFN or in the assembly
attention kernel would show as a differing run.  usage: stress_lg_repeat.py [runs=60] [kpts=2048] [pairs=8] [beside=none]
(pairs = 1: the single-pair path - key ranges in the attention, their merge inside the fused FFN's prologue; 2: the keyframe frames' launch;
beside: other work on streams of its own while the forwards run, scripts/aggressor_util.py - e.g. "aliked:2", two extractor streams)"""
import importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
import lg_inputs
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
ctx = pkg._native.default_context()
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lg = LG(W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0), max_kpts=N, max_pairs=B)
pairs = []
for b in range(B):
    k0, d0, k1, d1 = lg_inputs.make_pair(N - 37 * b, N - 11 * b, seed=11 + b)
    a = [ctx.upload(v) for v in (k0, d0, k1, d1)]
    pairs.append((a[0], a[1], len(k0), a[2], a[3], len(k1)))
ij = ctx.malloc(B * N * 8); sc = ctx.malloc(B * N * 4); info = ctx.malloc(B * 16)
BESIDE = sys.argv[4] if len(sys.argv) > 4 else "none"
from aggressor_util import make_aggressor
aggr_ctx, aggressor = make_aggressor(BESIDE, pkg._native, W, ROOT)
def run():
    aggressor()
    lg.match_batch_dev(pairs, ij, sc, info, N)
    ctx.sync()
    getattr(aggr_ctx, "sync_all", aggr_ctx.sync)()
    a = np.empty((B, N, 2), np.int32); s = np.empty((B, N), np.float32); i = np.empty((B, 4), np.int32)
    ctx.d2h(a, ij); ctx.d2h(s, sc); ctx.d2h(i, info)
    return a, s, i
ref = run()
bad = 0
for r in range(runs):
    if r == runs // 2:
        lg.use_graphs(True)
    got = run()
    for p in range(B):
        k = ref[2][p, 0]
        if not (np.array_equal(got[2][p], ref[2][p]) and np.array_equal(got[0][p, :k], ref[0][p, :k]) and np.array_equal(got[1][p, :k], ref[1][p, :k])):
            bad += 1
print(f"{runs} repeats of an {B}-pair batch at {N} keypoints (beside: {BESIDE}): {bad} differing (pair, run) results; matches per pair {ref[2][:, 0].tolist()}")
assert bad == 0
