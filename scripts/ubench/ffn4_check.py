"""Bit-identity of the 4-wave / 32-token fused FFN (debug_big_gemm 6) against the 8-wave / 64-token kernel (2) and the 8-wave
/ 32-token form (3) on a batch with ragged sizes, early stop and pruning; then the batched forward's time with each."""
import importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
ctx = pkg._native.default_context()
def run(sd, sizes, cap, modes, min_conf=0.0):
    lg = LG(sd, max_kpts=cap, max_pairs=len(sizes))
    pairs, keep = [], []
    for i, (m, n) in enumerate(sizes):
        k0, d0, k1, d1 = lg_inputs.make_pair(m, n, seed=40 + i)
        a = [ctx.upload(v) for v in (k0, d0, k1, d1)]; keep += a
        pairs.append((a[0], a[1], m, a[2], a[3], n))
    B = len(sizes)
    ij = ctx.malloc(B * cap * 8); sc = ctx.malloc(B * cap * 4); info = ctx.malloc(B * 16)
    out = {}
    for mode in modes:
        lg.debug_big_gemm(mode)
        lg.match_batch_dev(pairs, ij, sc, info, cap, min_conf=min_conf); ctx.sync()
        a = np.empty((B, cap, 2), np.int32); b = np.empty((B, cap), np.float32); c = np.empty((B, 4), np.int32)
        ctx.d2h(a, ij); ctx.d2h(b, sc); ctx.d2h(c, info)
        x = lg.debug_read(0, (2, lg.capacity, 256))
        out[mode] = (a, b, c, x)
    lg.close()
    return out
sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
o = run(sd, [(512, 512), (300, 417), (64, 33), (640, 1), (129, 128), (640, 640)], 640, (2, 3, 6))
for m in (3, 6):
    same = all(np.array_equal(o[2][k][: ], o[m][k]) for k in (2,)) and all(
        np.array_equal(o[2][0][p, :o[2][2][p, 0]], o[m][0][p, :o[m][2][p, 0]]) and np.array_equal(o[2][1][p, :o[2][2][p, 0]], o[m][1][p, :o[m][2][p, 0]])
        for p in range(6)) and np.array_equal(o[2][3], o[m][3])
    print(f"mode {m} vs 2: matches, scores, info, token state of pair 0 bit-identical: {same}   matches {o[m][2][:, 0].tolist()}")
sd2 = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=-4.6, conf_bias=2.3)          # early stop + pruning
o = run(sd2, [(400, 350), (256, 256), (128, 200)], 512, (2, 6))
print("pruning / early-stop weights: identical:", all(np.array_equal(o[2][k], o[6][k]) for k in (2, 3)), o[6][2].tolist())
