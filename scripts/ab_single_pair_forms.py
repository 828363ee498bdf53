"""A/B of the single-pair forms of the LightGlue forward (r05): the key-range merge as a launch of its own
(debug_key_split(-5), the r04 form) or folded into the fused FFN's prologue (0, the default).
For both: the match indices and scores of the same pairs (compared bit for bit) and the device time of one forward.
(Measured with this script and not kept: a 4-stage LDS-DMA ring in the 128 x 128 projections - no faster; the 64-row ring
projections with producer waves under the fused FFN - 1 356 against 1 338 us per forward.)

    python scripts/ab_single_pair_forms.py [N=2048] [iters=20]
"""
import importlib
import os
import sys
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
ctx = pkg._native.default_context()
sd = W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0)
sizes = [(N, N), (N - 37, N - 411), (N // 2 + 5, N)]
inputs = [lg_inputs.make_pair(m, n, seed=31 + i) for i, (m, n) in enumerate(sizes)]
base = None
for fold in (0, 1):
    lg = LG(sd, max_kpts=N, max_pairs=1)
    lg.debug_key_split(0 if fold else -5)
    outs = [lg.match(k0, d0, k1, d1, min_conf=0.0) for (k0, d0, k1, d1) in inputs]
    k0, d0, k1, d1 = inputs[0]
    a = [ctx.upload(v) for v in (k0, d0, k1, d1)]
    pair = [(a[0], a[1], N, a[2], a[3], N)]
    ij = ctx.malloc(N * 8); sc = ctx.malloc(N * 4); info = ctx.malloc(16)
    for _ in range(3):
        lg.match_batch_dev(pair, ij, sc, info, N)
    ctx.sync(); ctx.timer_start()
    for _ in range(iters):
        lg.match_batch_dev(pair, ij, sc, info, N)
    ms = ctx.timer_stop() / iters
    same = "reference form"
    if base is None:
        base = outs
    else:
        ok_ij = all(np.array_equal(o[0], b[0]) for o, b in zip(outs, base))
        ok_sc = all(np.array_equal(o[1], b[1]) for o, b in zip(outs, base))
        dmax = max((float(np.max(np.abs(o[1] - b[1]))) if o[1].shape == b[1].shape and len(o[1]) else 0.0) for o, b in zip(outs, base))
        same = f"indices {'identical' if ok_ij else 'DIFFER'}, scores {'bit-identical' if ok_sc else f'differ (max {dmax:.2e})'}"
    print(f"fold_merge={fold}: {ms * 1e3:8.1f} us per forward, matches {[len(o[0]) for o in outs]}, stop {[o[2] for o in outs]}; {same}", flush=True)
    for p_ in a + [ij, sc, info]:
        ctx.free(p_)
    lg.close()
