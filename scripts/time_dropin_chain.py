"""Unprofiled timeline of one drop-in frame of the `value` loop (extract + look-ahead match + filter): HIP timing events on the
extractor's and the matcher's streams (white box: the ring's own streams) and host stamps around the three calls.
    python scripts/time_dropin_chain.py [frames=40]"""
import gc, importlib, os, statistics, sys, time
os.environ.setdefault("SSLAM_ALLOW_RANDOM_WEIGHTS", "1")
os.environ.setdefault("SSLAM_RANDOM_LIGHTGLUE_ARGS", "seed=1,match_gain=4.0,match_bias=3.0")      # (bench.py's matcher: matches survive)
from pathlib import Path
from types import SimpleNamespace
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames, lg_inputs
fu = importlib.import_module("opencv-simpleslam_amd.slam.core.features_utils")
nat = importlib.import_module("opencv-simpleslam_amd")._native
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
import bench
args = SimpleNamespace(use_lightglue=True, max_features=bench.MAX_KPTS, min_conf=bench.MIN_CONF, ransac_thresh=2.5)
det, mat = fu.init_feature_pipeline(args)
ring = fu._ring_of(det)
if os.environ.get("CHAIN_GRAPHS"):                # (the ring replays the extraction as a hipGraph; CHAIN_GRAPHS=0: plain launches)
    det.use_graphs(os.environ["CHAIN_GRAPHS"] != "0")
planter = None if os.environ.get("CHAIN_NO_PLANT") else lg_inputs.PlantedExtractor(det, lg_inputs.make_chain(16, 2048, seed=7, noise=0.035, drop=0.1))
imgs = [frames.noise_frame(i) for i in range(8)]
ectx, mctx = ring.ctx, ring.mctx
# events: e0 before the image upload, e1 after it, e2 after extraction (+ planted copy), m0 / m1 around the matcher stream's work
EV = lambda: nat.Context.timing_event(ectx) if hasattr(nat.Context, "timing_event") else None
rows = {}
def add(k, v): rows.setdefault(k, []).append(v)
real_h2d = ectx.h2d_async
real_extract_dev = det.extract_dev
state = {}
def h2d_async(dptr, arr):
    if dptr == ring.img_dev:
        state["e0"] = ectx.timing_event(); ectx.record(state["e0"])
        t = time.perf_counter(); real_h2d(dptr, arr); state["h2d_host"] = time.perf_counter() - t
        state["e1"] = ectx.timing_event(); ectx.record(state["e1"])
    else:
        real_h2d(dptr, arr)
ectx.h2d_async = h2d_async
real_enqueue = ring._enqueue
def _enqueue(pairs, thr):
    state["m0"] = mctx.timing_event(); mctx.record(state["m0"])
    t = time.perf_counter(); real_enqueue(pairs, thr); state["enq_host"] = time.perf_counter() - t
    state["m1"] = mctx.timing_event(); mctx.record(state["m1"])
ring._enqueue = _enqueue
real_record = ectx.record
def record(ev):
    if ev == ring.ev_extracted:
        state["e2"] = ectx.timing_event(); real_record(state["e2"])
    real_record(ev)
ectx.record = record
prev = None
gc.collect(); gc.freeze()
t_prev_end = None
for f in range(N + 8):
    state.clear()
    im = imgs[f % len(imgs)]
    t0 = time.perf_counter()
    kp, des = fu.feature_extractor(args, im, det)
    t1 = time.perf_counter()
    if prev is not None:
        m = fu.feature_matcher(args, prev[0], kp, prev[1], des, mat)
        t2 = time.perf_counter()
        good = fu.filter_matches_ransac(prev[0], kp, m, args.ransac_thresh)
        t3 = time.perf_counter()
        if f >= 8 and "m1" in state:
            el = nat.Context.elapsed_ms
            add("host: feature_extractor call", (t1 - t0) * 1e3)
            add("host: feature_matcher call", (t2 - t1) * 1e3)
            add("host: filter_matches_ransac call", (t3 - t2) * 1e3)
            if t_prev_end is not None:
                add("host: between frames (loop bookkeeping)", (t0 - t_prev_end) * 1e3)
            add("host: image upload call (blocking, pageable source)", state["h2d_host"] * 1e3)
            add("host: look-ahead enqueue (match + filter + read-back)", state["enq_host"] * 1e3)
            add("gpu: image upload (e0 -> e1)", el(state["e0"], state["e1"]))
            add("gpu: extraction + planted copy (e1 -> e2)", el(state["e1"], state["e2"]))
            add("gpu: matcher stream, wait + match + filter + read-back (m0 -> m1)", el(state["m0"], state["m1"]))
            add("gpu: e0 -> m1 (the frame's whole device chain)", el(state["e0"], state["m1"]))
            add("matches", float(len(m))); add("kept by the filter", float(len(good)))
        t_prev_end = time.perf_counter()
    prev = (kp, des)
gc.unfreeze()
for k, v in rows.items():
    print(f"{k:70s} median {statistics.median(v):8.3f}   p10 {sorted(v)[len(v) // 10]:8.3f}   p90 {sorted(v)[9 * len(v) // 10]:8.3f}")
tot = [a + b + c + d for a, b, c, d in zip(rows["host: feature_extractor call"], rows["host: feature_matcher call"],
                                            rows["host: filter_matches_ransac call"], rows["host: between frames (loop bookkeeping)"] + [0.0])]
print(f"frame total (three calls + bookkeeping): median {statistics.median(tot):.3f} ms -> {1000 / statistics.median(tot):.1f} frames/s")
