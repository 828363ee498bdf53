"""Determinism of ALIKED extraction under concurrency: N extractor instances on streams of their own extract the same batches of
frames over and over without host synchronisation in between; every repeat must reproduce the first one bit for bit.
usage: stress_aliked_repeat.py [repeats=40] [instances=3] [F=2] [beside=none]
`beside`: other work on a stream of its own beside the extractors (scripts/aggressor_util.py), e.g. "lightglue:ring,noasm" - a
LightGlue matcher on its ring GEMMs and HIP attention kernel, the strongest trigger r06 found for the one fault this guards
against (profiles/r06_aggregate_rnorm_diagnosis.md)."""
import importlib, sys, hashlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
import frames
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 3
F = int(sys.argv[3]) if len(sys.argv) > 3 else 2
BESIDE = sys.argv[4] if len(sys.argv) > 4 else "none"
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
nat = pkg._native
K, H, Wd = 384, 376, 1241
sd = W.random_aliked_state_dict(0)
ctxs = [nat.Context(0) for _ in range(NI)]
dets = [AL(sd, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=c, max_frames=F) for c in ctxs]
imgs = [frames.structured_frame(i, h=H, w=Wd) for i in range(F * NI)]
d_img = [[ctxs[j].upload(imgs[j * F + f]) for f in range(F)] for j in range(NI)]
out = [[dict(xy=ctxs[j].malloc(K * 8), desc=ctxs[j].malloc(K * 512), sc=ctxs[j].malloc(K * 4), n=ctxs[j].malloc(64)) for f in range(F)] for j in range(NI)]
def enqueue(j):
    o = out[j]
    dets[j].extract_batch_dev(d_img[j], H, Wd, 3, [x["xy"] for x in o], [x["desc"] for x in o], [x["sc"] for x in o], [x["n"] for x in o])
def snapshot():
    res = []
    for j in range(NI):
        ctxs[j].sync()
        for f in range(F):
            n = np.empty(16, np.int32); ctxs[j].d2h(n, out[j][f]["n"])
            xy = np.empty((K, 2), np.float32); ctxs[j].d2h(xy, out[j][f]["xy"])
            desc = np.empty((K, 128), np.float32); ctxs[j].d2h(desc, out[j][f]["desc"])
            res.append((int(n[0]), xy[:n[0]].copy(), desc[:n[0]].copy()))
    return res
def stage_hashes(j):                         # frame 0 of instance j: sha1 of the stage buffers the debug hook exposes
    d = dets[j].debug_read(2, (8,), np.int32)
    Hp, Wp = int(d[2]), int(d[3])
    shapes = {0: (int(d[0]), int(d[1])), 3: (16, Hp, Wp), 4: (32, Hp // 2, Wp // 2), 5: (64, Hp // 8, Wp // 8), 6: (128, Hp // 32, Wp // 32),
              10: (Hp, Wp, 32), 11: (Hp, Wp), 12: (32, Hp // 2, Wp // 2), 13: (32, Hp // 8, Wp // 8), 14: (32, Hp // 32, Wp // 32),
              15: (13, Hp // 2, Wp // 2), 16: (13, Hp // 8, Wp // 8), 17: (13, Hp // 32, Wp // 32), 18: (8, Hp, Wp)}
    return {k: hashlib.sha1(dets[j].debug_read(k, sh).tobytes()).hexdigest()[:10] for k, sh in shapes.items()}
NAMES = {0: "score", 3: "x1", 4: "x2", 5: "x3", 6: "x4", 10: "g1cl", 11: "rnorm", 12: "g2", 13: "g3", 14: "g4", 15: "pre2", 16: "pre3", 17: "pre4", 18: "s8"}
from aggressor_util import make_aggressor
aggr_ctx, aggressor = make_aggressor(BESIDE, nat, W, ROOT)
for j in range(NI):
    enqueue(j)
ref = snapshot()
ref_st = [stage_hashes(j) for j in range(NI)]
ref_rn = []
for j in range(NI):
    d = dets[j].debug_read(2, (8,), np.int32)
    ref_rn.append(dets[j].debug_read(11, (int(d[2]), int(d[3]))))
bad = 0
for r in range(reps):
    for k in range(4):                      # four un-synchronised calls per instance, interleaved across the streams
        if k % 2 == 0:
            aggressor()
        for j in range(NI):
            enqueue(j)
    got = snapshot()
    aggr_ctx.sync()
    for i, (a, b) in enumerate(zip(ref, got)):
        if a[0] != b[0] or not np.array_equal(a[1], b[1]) or not np.array_equal(a[2], b[2]):
            nd = int((a[2] != b[2]).any(axis=1).sum()) if a[0] == b[0] else -1
            print(f"repeat {r}: frame {i} differs (n {a[0]} / {b[0]}, descriptors differing: {nd}, max |d| {np.abs(a[2] - b[2]).max() if a[0] == b[0] else -1:.3g})", flush=True)
            bad += 1
    for j in range(NI):
        st = stage_hashes(j)
        diff = [NAMES[k] for k in st if st[k] != ref_st[j][k]]
        if diff:
            print(f"repeat {r}: instance {j} frame 0 stage buffers differ: {diff}", flush=True)
            if "rnorm" in diff:
                d = dets[j].debug_read(2, (8,), np.int32); Hp, Wp = int(d[2]), int(d[3])
                cur = dets[j].debug_read(11, (Hp, Wp)); ys, xs = np.nonzero(cur != ref_rn[j])
                print(f"    rnorm: {len(ys)} pixels differ; rows {sorted(set(ys.tolist()))[:8]} cols {xs.min()}..{xs.max()}; "
                      f"ref {ref_rn[j][ys[0], xs[0]]!r} now {cur[ys[0], xs[0]]!r}; max rel {np.abs(cur / ref_rn[j] - 1).max():.3g}", flush=True)
print(f"{reps} repeats x {NI} instances x {F} frames (beside: {BESIDE}): {bad} mismatching frame results", flush=True)
sys.exit(1 if bad else 0)
