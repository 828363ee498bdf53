"""sha1 of LightGlue outputs (match indices, scores, per-pair info; batched and single-pair entries, with and without early stop /
pruning) on fixed inputs: run before and after a kernel change that claims bit-identical results and diff the two printouts."""
import hashlib, importlib, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
ctx = pkg._native.default_context()
sd = W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0)
for N, B in ((2048, 8), (512, 2)):
    lg = LG(sd, max_kpts=N, max_pairs=B)
    pairs, host = [], []
    for b in range(B):
        n0, n1 = (N, N) if b % 2 == 0 else (N - 37 * b, N - 11 * b)
        k0, d0, k1, d1 = lg_inputs.make_pair(n0, n1, seed=11 + b)
        host.append((k0, d0, k1, d1))
        a = [ctx.upload(v) for v in (k0, d0, k1, d1)]
        pairs.append((a[0], a[1], len(k0), a[2], a[3], len(k1)))
    ij = ctx.malloc(B * N * 8); sc = ctx.malloc(B * N * 4); info = ctx.malloc(B * 16)
    lg.match_batch_dev(pairs, ij, sc, info, N); ctx.sync()
    inf = np.empty((B, 4), np.int32); ctx.d2h(inf, info)
    IJ = np.empty((B, N, 2), np.int32); ctx.d2h(IJ, ij)
    SC = np.empty((B, N), np.float32); ctx.d2h(SC, sc)
    h = hashlib.sha1()
    for b in range(B):
        m = max(int(inf[b, 0]), 0)
        h.update(IJ[b, :m].tobytes()); h.update(SC[b, :m].tobytes()); h.update(inf[b].tobytes())
    print(f"batch N={N} B={B} matches={inf[:, 0].tolist()} {h.hexdigest()}")
    k0, d0, k1, d1 = host[0]
    ij1, sc1, stop = lg.match(k0, d0, k1, d1, min_conf=0.1)
    print(f"single N={N} matches={len(ij1)} stop={stop} {hashlib.sha1(ij1.tobytes() + sc1.tobytes()).hexdigest()}")
    lg.close()
