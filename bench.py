#!/usr/bin/env python3
"""bench.py - frames/s of the ALIKED + LightGlue hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): frames/sec ALIKED+LightGlue @1241x376.  Workload =
config C2/C4: synthetic 1241x376x3 uint8 frames (white
noise, default_rng(1234 + frame)), ALIKED-n16 -> 2048 keypoints per frame,
LightGlue(features='aliked') on every (t-1, t) pair, min_conf 0.7, random-init
weights of the upstream architecture (no network for checkpoints).

A "step" = one round of the frame-sharded pipeline: every rank extracts its
B frames and matches each against its predecessor (B extracts + B matches per
rank; the matches run as batched launches of P pairs).  Inputs are resident in
HBM before the timed region.  One process per GPU.  `python bench.py --gpus N`
with N > 1 starts the N ranks itself (`python -m torch.distributed.run`, as a
CHILD process before anything touches the GPU) and relays rank 0's JSON line,
so one command shape works for N = 1, 2, 4, 8; under torch.distributed.run it
is one of the ranks.  At N = 1 the whole run is on the C-ABI (no torch).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     - the dominant kernel (LightGlue attention, split-f16 MFMA):
                 HIP events around every launch; `frac` from the launches of a
                 batch replayed on an otherwise idle GPU right after the timed
                 region, `in_pipeline_*` from a few rounds of the running
                 pipeline with the brackets on (the timed region itself replays
                 hipGraphs, which host-side event records cannot bracket)
  exact_f32    - the same pipeline with every contraction on the exact-fp32
                 matrix-core instruction (precision 0)
  step_ms      - p10 / p50 / p90 of the per-round durations inside the timed
                 region (timing events on a collector stream, no host sync)
  dropin       - the six names of slam/core/features_utils.py driven as
                 slam/monocular/main_revamped.py drives them (one frame at a
                 time, host objects in and out): frames/s of that literal path
                 (+ `dropin.cv2_classes`: the matched loops in a child interpreter where `cv2`
                 is importable - a stand-in with C value classes - so the lists hold cv2's own
                 KeyPoint / DMatch objects, as they do wherever the reference runs; and the same
                 loops in a partner child without cv2: `vs_duck_types_same_conditions`)
  planted_matches - the pipeline of `value` on frames that MATCH (records overwritten
                 behind every batched extraction with a synthetic matched chain, LightGlue
                 weights with a sharp assignment head): hundreds of matches per pair
  early_stop   - the pipeline with random weights whose token-confidence biases
                 are calibrated on one pair of the stream so that points are
                 pruned and pairs stop early (depth AND width control under load)
  f16x3        - the same pipeline in the other split form "f16x3" (three MFMAs per product in P.V too; the default until r04),
                 with the attention launch's roofline figure in that mode (never `value`)
  pcie         - the same pipeline with the host in the loop: every round's frames uploaded from page-locked host
                 memory and its {count, pairs} read back, both on copy streams of their own (never `value`)
  c5, kpts4000 - the other stated sizes: 1920x1080 frames and the reference CLI's default of 4000
                 keypoints per frame (main_revamped.py:206), each with the attention kernel's own roofline fraction
  ba, reproject- the C3 local-BA solve and the 2D-3D association
  cpu_baseline - the torch-CPU oracle (kind "port") on a bounded sample
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

H_IMG, W_IMG, C_IMG = 376, 1241, 3
MAX_KPTS = 2048
MIN_CONF = 0.7
FRAMES_PER_RANK = int(os.environ.get("SSLAM_BENCH_FRAMES", 24))     # frames per GPU per step
BATCH_PAIRS = int(os.environ.get("SSLAM_BENCH_PAIRS", 8))           # pairs per batched LightGlue enqueue
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak (spec, no sparsity)
F32_MFMA_PEAK_TFLOPS = 157.3           # same table: v_mfma_f32_32x32x2_f32, the exact-fp32 matrix-core rate
HBM_PEAK_GBS = 8000.0
ALIKED_GFLOP_PER_FRAME = 8.9           # 6.56 dense conv + 2.34 SDDH at 2048 keypoints
# early_stop leg: the fraction of a pair's points each token-confidence head is calibrated to call "confident" (layers 0..7) and
# the matchability bias (every point unmatchable -> confident points are pruned); calibrated_confidence_heads() below
EARLY_STOP_CONFIDENT = [float(v) for v in os.environ.get("SSLAM_BENCH_CONFIDENT", "0.4,0.4,0.5,0.6,0.8,0.97,0.97,0.97").split(",")]
EARLY_STOP_MATCH_BIAS = float(os.environ.get("SSLAM_BENCH_MATCH_BIAS", -9.0))


def _pmc_traffic():
    """Per-launch HBM-side bytes of the attention kernel at the bench size, from profiles/ (None when absent)."""
    for name in ("r06_attention_traffic.json", "r05_attention_traffic.json", "r04_attention_traffic.json", "r03_attention_traffic.json", "r02final_attention_traffic.json", "r02_attention_traffic.json", "r01_attention_traffic.json"):
        try:
            d = json.loads((ROOT / "profiles" / name).read_text())
            return int(d["fetch_bytes_per_launch"]) + int(d["write_bytes_per_launch"]), name
        except Exception:
            continue
    return None, None


def _source_digest():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sslam_build", ROOT / "opencv-simpleslam_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_digest()


def _git_head():
    try:
        r = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=str(ROOT), capture_output=True, text=True, timeout=5)
        return r.stdout.strip() or None if r.returncode == 0 else None
    except Exception:
        return None


def lightglue_gflop(n, layers):
    """SURVEY 8(d) F(N, L): algorithmic FLOPs of one pair with M = N = n keypoints, L layers executed."""
    D, d_in = 256, 128
    blk_self = 6 * n * D * D + 4 * n * n * D + 2 * n * D * D + 8 * n * D * D + 4 * n * D * D
    blk_cross = 4 * n * D * D + 4 * n * n * D + 2 * n * D * D + 8 * n * D * D + 4 * n * D * D
    return (4 * n * d_in * D + layers * (2 * blk_self + 2 * blk_cross) + 4 * n * D * D + 2 * n * n * D) / 1e9


def noise_frame(idx):
    return np.random.default_rng(1234 + idx).integers(0, 256, (H_IMG, W_IMG, C_IMG), dtype=np.uint8)


_STRUCT_BASE = None


def structured_frame(idx):
    """SURVEY 8(d) second input: the noise low-pass filtered with a 9x9 box (stretched back to
    0..255) and translated 3 px per frame, so consecutive frames really overlap."""
    global _STRUCT_BASE
    if _STRUCT_BASE is None:
        from scipy.ndimage import uniform_filter
        base = np.random.default_rng(1234).integers(0, 256, (H_IMG, W_IMG + 512, C_IMG)).astype(np.float32)
        low = uniform_filter(base, size=(9, 9, 1), mode="reflect")
        lo, hi = low.min(), low.max()
        _STRUCT_BASE = np.clip((low - lo) * (255.0 / (hi - lo)), 0, 255).astype(np.uint8)
    off = (3 * idx) % 512
    return np.ascontiguousarray(_STRUCT_BASE[:, off:off + W_IMG])


def attention_flops(n0, n1):
    """Algorithmic FLOPs of the attention of ONE pair in one half layer (both images, 4 heads x 64):
    SURVEY 8(d) counts 4 N^2 D per image per block (QK^T + AV)."""
    return 2.0 * 256 * (n0 * n1 * 2) * 2


# --------------------------------------------------------------------------- CPU baseline
def _cpu_leg(threads, warmup, timed, budget_s, probe_limit_s=6.0):
    """`warmup` untimed + `timed` timed frames, but never more than ~budget_s seconds in all: a host
    where one frame takes tens of seconds (hundreds of threads on small operators) gets fewer
    warm-up and timed frames (at least one of each), and the leg says how many it took."""
    import torch
    from oracle import aliked_ref, lightglue_ref
    W = importlib.import_module("opencv-simpleslam_amd.weights")
    torch.set_num_threads(int(threads))
    sd_a, sd_l = W.random_aliked_state_dict(0), W.random_lightglue_state_dict(0)
    t_begin = time.perf_counter()
    prev = aliked_ref.aliked_extract(sd_a, noise_frame(0), MAX_KPTS)
    probe = time.perf_counter() - t_begin
    if probe > probe_limit_s:
        # e.g. 256 intra-op threads on these small operators: ~200 s per frame (measured); say so
        # and stop instead of spending the run on it
        return {"threads": int(torch.get_num_threads()), "skipped": True, "first_extract_s": round(probe, 2),
                "why": f"the first ALIKED extraction alone took {probe:.1f} s (limit {probe_limit_s} s)",
                "frames_per_s": 0.0}
    per_frame, n_warm, i = [], 0, 0
    while len(per_frame) < timed:
        i += 1
        t0 = time.perf_counter()
        cur = aliked_ref.aliked_extract(sd_a, noise_frame(i), MAX_KPTS)
        lightglue_ref.reference_feature_matcher(sd_l, prev["keypoints"], cur["keypoints"],
                                                prev["descriptors"], cur["descriptors"], MIN_CONF)
        prev = cur
        dt = time.perf_counter() - t0
        elapsed = time.perf_counter() - t_begin
        if n_warm < warmup and (n_warm == 0 or elapsed + (warmup - n_warm + 2) * dt < budget_s):
            n_warm += 1                                  # still warming up (and the budget allows it)
            continue
        per_frame.append(dt)
        if elapsed + dt > budget_s:
            break
    a = np.array(per_frame)
    return {"threads": int(torch.get_num_threads()), "frames_timed": len(a), "warmup_frames": n_warm,
            "median_s_per_frame": round(float(np.median(a)), 4), "p10_s": round(float(np.percentile(a, 10)), 4),
            "p90_s": round(float(np.percentile(a, 90)), 4), "frames_per_s": round(1.0 / float(np.median(a)), 4)}


def dropin_leg(n_frames=96, only_matched_loops=False):
    """The literal drop-in path: init_feature_pipeline / feature_extractor / feature_matcher / filter_matches_ransac exactly as
    slam/monocular/main_revamped.py calls them - one frame at a time, host arrays and KeyPoint / DMatch objects in and out,
    nothing overlapped by the caller.  Three loops over the same 1241x376 frames:

      frame_loop   real extracted features through all three calls.  With random-init networks almost nothing matches (an
                   untrained ALIKED head gives descriptors with pairwise cosine 0.9995), so the filter / DMatch / read-back work is
                   empty: extract + match only.
      value        one match per frame (main_revamped.py:325-330 prev -> cur + RANSAC) on PLANTED features: right behind every
                   extraction, on the extractor's own stream, the frame's device record is overwritten with the next frame of a
                   synthetic chain (tests/lg_inputs.py::PlantedExtractor - 2048 keypoints, any two frames share true
                   correspondences), so the matcher returns hundreds of matches, the F-matrix RANSAC runs on them, the DMatch
                   objects are built and filtered.  The overwrite (three asynchronous uploads, ~1 MB) is INSIDE the timed calls.
      slam_loop    the reference's real call pattern on the same planted chain: per frame extract +
                   prev -> cur + RANSAC; beyond the keyframe cooldown (kf_cooldown 5, main_revamped.py:221) additionally
                   keyframe -> cur + RANSAC (keyframe_utils.py:153-154) and - the frame being promoted - the same pair again
                   (triangulation_utils.py:131-132); the new frame becomes the keyframe.  frames/s over ALL frames, the keyframe
                   frames' extra calls included."""
    os.environ.setdefault("SSLAM_ALLOW_RANDOM_WEIGHTS", "1")         # no checkpoints in the image
    # random-init weights whose assignment head is sharp enough to match the synthetic features (as the parity tests use them)
    os.environ.setdefault("SSLAM_RANDOM_LIGHTGLUE_ARGS", "seed=1,match_gain=4.0,match_bias=3.0")
    from types import SimpleNamespace
    fu = importlib.import_module("opencv-simpleslam_amd.slam.core.features_utils")
    args = SimpleNamespace(use_lightglue=True, max_features=MAX_KPTS, min_conf=MIN_CONF)
    import logging
    logging.getLogger("opencv_simpleslam_amd").setLevel(logging.ERROR)
    det, mat = fu.init_feature_pipeline(args)
    ring = fu._ring_of(det)
    imgs = [structured_frame(i) for i in range(n_frames + 3)]
    med = lambda a: round(float(np.median(a)) * 1e3, 3)
    KF_COOLDOWN, RANSAC_THR = 5, 2.5                                   # main_revamped.py:221, :212

    def loop(keyframes, mat=mat):
        ring.forget_patterns()
        stats0 = dict(ring.stats)
        kp_prev, des_prev = fu.feature_extractor(args, imgs[0], det)
        kf, last_kf = (kp_prev, des_prev), 0
        te, tm, tr, tk, tot, nm, nf, nk, kf_flags = [], [], [], [], [], [], [], [], []
        for i, im in enumerate(imgs[1:]):
            f = i + 1
            t0 = time.perf_counter(); kp, des = fu.feature_extractor(args, im, det); t1 = time.perf_counter()
            m = fu.feature_matcher(args, kp_prev, kp, des_prev, des, mat); t2 = time.perf_counter()
            flt = fu.filter_matches_ransac(kp_prev, kp, m, RANSAC_THR); t3 = time.perf_counter()
            extra = 0
            if keyframes and f - last_kf > KF_COOLDOWN:
                r1 = fu.feature_matcher(args, kf[0], kp, kf[1], des, mat)                       # select_keyframe
                k1 = fu.filter_matches_ransac(kf[0], kp, r1, RANSAC_THR)
                r2 = fu.feature_matcher(args, kf[0], kp, kf[1], des, mat)                       # triangulate_between_kfs_2view
                k2 = fu.filter_matches_ransac(kf[0], kp, r2, RANSAC_THR)
                kf, last_kf, extra = (kp, des), f, 1
                if f >= 2 * (KF_COOLDOWN + 1):
                    nk.append((len(r1), len(k1), len(r2), len(k2)))
            t4 = time.perf_counter()
            kp_prev, des_prev = kp, des
            if f >= (2 * (KF_COOLDOWN + 1) if keyframes else 3):            # warm-up: graph capture, first touches, the learned patterns
                te.append(t1 - t0); tm.append(t2 - t1); tr.append(t3 - t2); tot.append(t4 - t0); kf_flags.append(bool(extra))
                if extra:
                    tk.append(t4 - t3)
                nm.append(len(m)); nf.append(len(flt))
        out = {"value": round(len(tot) / float(np.sum(tot)), 1) if keyframes else round(1.0 / float(np.median(tot)), 1),
               "unit": "frames/s", "frames_timed": len(tot),
               "feature_extractor_ms": med(te), "feature_matcher_ms": med(tm), "filter_matches_ransac_ms": med(tr),
               "keypoints": len(kp), "matches_median": int(np.median(nm)), "ransac_inliers_median": int(np.median(nf)),
               "answered_from": {k: ring.stats[k] - stats0[k] for k in ring.stats}}
        if keyframes:
            out["keyframe_frames"] = len(tk)
            out["keyframe_extra_calls_ms"] = med(tk)
            out["keyframe_matches_median"] = [int(v) for v in np.median(np.array(nk), axis=0)] if nk else None
            out["mean_ms_per_frame"] = round(float(np.mean(tot)) * 1e3, 3)
            plain = [t for t, k in zip(tot, kf_flags) if not k]
            out["frame_ms"] = {"plain_median": med(plain), "keyframe_median": med([t for t, k in zip(tot, kf_flags) if k]),
                               "p90": round(float(np.percentile(tot, 90)) * 1e3, 3), "max": round(float(np.max(tot)) * 1e3, 3)}
        return out

    # the loops below create ~4 000 small objects per frame (KeyPoint / DMatch), so the cyclic collector runs all the time; what
    # must NOT be in these per-call latencies is its full pass over everything the OTHER legs of this process left alive
    # (millions of objects: tens of ms every ~17 frames).  gc.freeze() parks the objects that exist now in the permanent
    # generation; collection of what the loops themselves allocate stays on, as in a caller's process.
    import gc
    gc.collect(); gc.freeze()
    frame_loop = None if only_matched_loops else loop(False)

    sys.path.insert(0, str(ROOT / "tests"))
    import lg_inputs
    chain = lg_inputs.make_chain(16, MAX_KPTS, seed=7, noise=0.035, drop=0.1)      # (two frames: the noise and 81 % co-visibility of the parity pairs)
    planter = lg_inputs.PlantedExtractor(det, chain)
    adaptive = None
    try:
        if only_matched_loops:
            # a child process of `dropin.cv2_classes` starts on a GPU that has idled for the seconds its imports and weight
            # set-up took, and this loop is a LIGHT load (short kernels, one stream at a time): ~1 500 untimed frames (3 s)
            # before the timed ones - without them whichever child ran second measured 2 - 4 % faster
            for _ in range(16):
                loop(False); planter.i = 0
        planted = loop(False)
        planter.i = 0
        slam = loop(True)
        # the same slam loop with LightGlue's own DEPTH control able to fire.  Random-init token-confidence heads are never
        # confident, so every pair above runs all 9 layers - the worst case; trained weights on overlapping frames stop after a
        # few.  Here the BIAS of each confidence head is set from one pair of the chain (calibrated_confidence_heads: the
        # fractions EARLY_STOP_CONFIDENT of the points above the layer's threshold), matchability untouched: the decisions
        # are data dependent, pairs stop early, matches survive.  Never `value`.
        try:
            if only_matched_loops:
                raise StopIteration
            Wm = importlib.import_module("opencv-simpleslam_amd.weights")
            LightGlueHIP = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
            sd_e = calibrated_confidence_heads(Wm.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0, conf_gain=16.0),
                                               chain[:2], mat.ctx)
            mat_e = LightGlueHIP(sd_e, max_kpts=MAX_KPTS, ctx=mat.ctx, max_pairs=ring.PAIRS)
            mat_e._feature_ring = ring
            ring.attach_matcher(mat_e)
            planter.i = 0
            adaptive = loop(True, mat=mat_e)
            adaptive["lightglue_layers_last_pair"] = int(ring.pin_info[0, 1])
            ring.attach_matcher(mat)
            mat_e.close()
        except StopIteration:
            pass
        except Exception as e:                       # (an auxiliary figure must not cost the leg)
            adaptive = {"error": repr(e)}
    finally:
        planter.restore()
        gc.unfreeze()
    det.close(); mat.close()
    out = dict(planted)
    if only_matched_loops:
        out["slam_loop"] = slam
        out["matches_last_pair"] = planted["matches_median"]
        return out
    out["frame_loop"] = dict(frame_loop, what="real structured frames through all three calls; random-init networks match (almost) "
                             "nothing, so this is extract + match only (the prev -> cur match rides behind the extraction)")
    out["slam_loop"] = dict(slam, what="main_revamped.py's call pattern with kf_cooldown 5 on the planted chain: every frame extract + "
                            "prev -> cur + RANSAC, every 6th frame also keyframe -> cur + RANSAC twice (select_keyframe, then the "
                            "triangulation's repeat of the same pair), the frame becoming the keyframe; value = timed frames / their "
                            "summed time (median x 1 would hide the keyframe frames)")
    if adaptive is not None:
        out["slam_loop_depth_control"] = dict(adaptive, what="slam_loop with token-confidence heads whose BIASES are calibrated on one pair of "
                                              "the chain so that LightGlue's early stop fires (matchability untouched: no pruning, matches kept): "
                                              "what the same calls cost when the network's own depth control is active, as it is with trained "
                                              "weights on overlapping frames; an illustration, never `value`")
    out["matches_last_pair"] = planted["matches_median"]
    try:
        out["value_kpts4000"] = dropin_value_at(4000, 20)
    except Exception as e:                       # (an auxiliary figure must not cost the leg)
        out["value_kpts4000"] = {"error": repr(e)}
    out["what"] = ("sequential host API as main_revamped.py drives it, one frame at a time, host objects in and out: feature_extractor on "
                   "1241x376 frames whose device record is overwritten, on the extractor's stream inside the timed call, with the next "
                   "frame of a synthetic MATCHED chain (2048 keypoints) so that RANSAC, DMatch construction and the read-back do real "
                   "work; then feature_matcher + filter_matches_ransac (threshold 2.5 px, the reference's default)")
    return out


def dropin_cv2_classes_leg():
    """`dropin.value` and `dropin.slam_loop` once more in a CHILD interpreter where `cv2` is importable, so that the overlay
    hands out cv2's own KeyPoint / DMatch objects - the only environment slam/monocular/main_revamped.py runs in
    (features_utils.py:2, :61-63, :80-83).  The wheel is absent from the image: `cv2` is tests/cv2_stub.py with the value
    classes of tests/cv2like/cv2like.c (C structs behind python objects, eager construction, a fresh tuple per `pt` read,
    KeyPoint_convert in one C pass - the cost model of the wheel's classes).  Two child processes (scripts/dropin_bench_cv2.py),
    with the stand-in and - the like-for-like partner - without; main() runs them BEFORE this process creates its own GPU
    contexts (run behind the other legs, with this process's dozen idle streams mapped on the hardware queues, the child's
    frame chain measured 3 - 9 % slower than the same child alone: 0.91 - 0.98 of `dropin.value` from call to call)."""
    def child(*flags):
        res = subprocess.run([sys.executable, str(ROOT / "scripts" / "dropin_bench_cv2.py"), *flags], capture_output=True, text=True,
                             timeout=900, env=dict(os.environ, SSLAM_ALLOW_RANDOM_WEIGHTS="1"))
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not lines:
            raise RuntimeError(res.stdout[-500:] + res.stderr[-1500:])
        return json.loads(lines[-1])
    try:
        r = child()
        duck = child("--duck-types")            # the like-for-like partner: same loops, same kind of process, no cv2
    except Exception as e:
        return {"error": repr(e)}
    keep = ("value", "unit", "frames_timed", "feature_extractor_ms", "feature_matcher_ms", "filter_matches_ransac_ms", "keypoints",
            "matches_median", "ransac_inliers_median", "answered_from")
    out = {k: r[k] for k in keep if k in r}
    out["slam_loop"] = {k: r["slam_loop"][k] for k in ("value", "unit", "frames_timed", "mean_ms_per_frame", "frame_ms", "keyframe_frames",
                                                       "keyframe_matches_median", "answered_from") if k in r["slam_loop"]}
    out["classes"] = r.get("classes")
    out["duck_types_same_conditions"] = {"value": duck["value"], "slam_loop": duck["slam_loop"]["value"],
                                         "feature_matcher_ms": duck["feature_matcher_ms"], "filter_matches_ransac_ms": duck["filter_matches_ransac_ms"]}
    out["vs_duck_types_same_conditions"] = round(r["value"] / duck["value"], 3)
    out["what"] = ("`dropin.value` / `dropin.slam_loop` in a child interpreter with `cv2` importable (a stand-in whose KeyPoint / DMatch are C "
                   "structs built eagerly, like the wheel's): the lists handed out hold cv2's own classes - cv2.KeyPoint_convert builds a "
                   "frame's keypoints in one C pass, DMatch objects are made behind the running match and their indices stored when it "
                   "returns, every list is read back in full before a device-resident result is trusted")
    return out


def dropin_value_at(max_kpts, n_frames):
    """The `value` loop of the drop-in leg (one match per frame + the filter, planted chain) at another keypoint budget - 4000 is
    the reference's own default (`main_revamped.py:206` --max_features); BASELINE.json quotes the metric at 2048."""
    import gc
    from types import SimpleNamespace
    fu = importlib.import_module("opencv-simpleslam_amd.slam.core.features_utils")
    sys.path.insert(0, str(ROOT / "tests"))
    import lg_inputs
    args = SimpleNamespace(use_lightglue=True, max_features=max_kpts, min_conf=MIN_CONF)
    det, mat = fu.init_feature_pipeline(args)
    planter = lg_inputs.PlantedExtractor(det, lg_inputs.make_chain(8, max_kpts, seed=7, noise=0.035, drop=0.1))
    imgs = [structured_frame(i) for i in range(n_frames + 4)]
    tot, nm, nf = [], [], []
    gc.collect(); gc.freeze()
    try:
        kp_prev, des_prev = fu.feature_extractor(args, imgs[0], det)
        for f, im in enumerate(imgs[1:], 1):
            t0 = time.perf_counter()
            kp, des = fu.feature_extractor(args, im, det)
            m = fu.feature_matcher(args, kp_prev, kp, des_prev, des, mat)
            flt = fu.filter_matches_ransac(kp_prev, kp, m, 2.5)
            t1 = time.perf_counter()
            kp_prev, des_prev = kp, des
            if f >= 4:
                tot.append(t1 - t0); nm.append(len(m)); nf.append(len(flt))
    finally:
        planter.restore()
        gc.unfreeze()
        det.close(); mat.close()
    return {"value": round(1.0 / float(np.median(tot)), 1), "unit": "frames/s", "frames_timed": len(tot), "keypoints": len(kp),
            "frame_ms": round(float(np.median(tot)) * 1e3, 3), "matches_median": int(np.median(nm)), "ransac_inliers_median": int(np.median(nf)),
            "what": f"the one-match-per-frame loop of `dropin.value` at max_features = {max_kpts} (the reference's default is 4000)"}


def cpu_baseline():
    """Reference path restated on torch-CPU (oracle/), timed on this host's cores:
    same process, same inputs, fp32, 3 warm-up + 10 timed frames, median and p10 / p90, with
    torch.set_num_threads(os.cpu_count()) and the count printed.  torch intra-op threading stops
    scaling well before a 256-thread host is full on these small operators, so a 16-thread leg is
    timed as well and the faster of the two is `value` (both are reported).  Bounded: each leg
    stops after its time budget with at least 3 timed frames."""
    ncpu = os.cpu_count() or 1
    # the all-cores leg is probed in a CHILD process with a hard limit first: on a 256-thread host the
    # first extraction alone was measured at 86 - 200 s, and a thread cannot be abandoned in-process
    probe = None
    if ncpu > 16:
        import torch                                       # page the wheel in here, not inside the child's time limit
        code = ("import sys, time, importlib; sys.path.insert(0, %r); import torch; import bench; "
                "from oracle import aliked_ref; W = importlib.import_module('opencv-simpleslam_amd.weights'); "
                "torch.set_num_threads(%d); t = time.perf_counter(); "
                "aliked_ref.aliked_extract(W.random_aliked_state_dict(0), bench.noise_frame(0), %d); "
                "print(time.perf_counter() - t)") % (str(ROOT), ncpu, MAX_KPTS)
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=45)
            probe = float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 1e9
        except Exception:
            probe = 1e9                                    # timed out: far beyond the 6 s limit
    if probe is not None and probe > 6.0:
        full = {"threads": ncpu, "skipped": True, "first_extract_s": None if probe >= 1e9 else round(probe, 2),
                "why": "the first ALIKED extraction with %d torch threads did not finish within the 45 s probe (imports included) "
                       "(limit for running the leg: 6 s)" % ncpu if probe >= 1e9 else
                       f"the first ALIKED extraction alone took {probe:.1f} s (limit 6.0 s)",
                "frames_per_s": 0.0}
    else:
        full = _cpu_leg(ncpu, 3, 10, 30.0)
    legs = {"all_cores": full}
    if ncpu > 16:
        # torch intra-op threading on these small operators peaks far below the host's thread count: look for the
        # best of 16 / 32 / 64 (each leg bounded: ~12 s, at least one warm-up and three timed frames)
        for nt in (16, 32, 64):
            if nt < ncpu:
                legs[f"threads_{nt}"] = _cpu_leg(nt, 2, 8, 12.0)
    best = max(legs.values(), key=lambda d: d["frames_per_s"])
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "host_logical_cores": ncpu,
            "all_cores_note": "torch.set_num_threads(os.cpu_count()) on a 256-thread host runs this path at 86 - 200 s per "
                              "frame (r02 measurements, profiles/r02_bench_n1.json): the leg is probed in a child process "
                              "with a 45 s limit and skipped when its first extraction exceeds 6 s" if legs["all_cores"].get("skipped") else None,
            "sample": f"{best['frames_timed']} timed frames 1241x376 after {best['warmup_frames']} warm-up (extract + "
                      f"match t-1->t, 2048 kpts, 9 layers), torch-CPU oracle, median {best['median_s_per_frame']} s/frame "
                      f"(p10 {best['p10_s']}, p90 {best['p90_s']})",
            "legs": legs}


def ba_cpu_baseline(prob, max_iters, target_cost=None, scipy_max_nfev=30):
    """BA on the host cores, two figures.

    `value` side, LIKE FOR LIKE: the oracle's Levenberg-Marquardt (oracle/ba_ref.py::solve_dense_lm, sparse form: ONE Jacobian
    in CSR, the full normal equations factorised by SuperLU) - the same objective as the reference configures and the device
    solves (Huber(2.0) on every 2-vector reprojection block, ba_utils.py:236; quaternion manifold; Ceres' default trust-region
    policy), the same iteration cap, costs in the same definition 0.5 sum rho(||r||^2).  kind "port".

    `scipy_trf`, NOT like for like (SURVEY 8(d)'s suggestion, kept for continuity with r01 - r04): SciPy
    least_squares(trf, loss='huber', f_scale=2, jac_sparsity) applies Huber per scalar COMPONENT - a different objective -
    and is capped at `scipy_max_nfev` evaluations, so its robust cost in the device's definition stays far above the
    device's: an unconverged solve of another problem, reported as such."""
    from scipy.optimize import least_squares
    from scipy.sparse import lil_matrix
    from oracle import ba_ref
    t0 = time.perf_counter()
    _, _, _, info = ba_ref.solve_dense_lm(prob.q, prob.t, prob.pose_const, prob.X, prob.intr, prob.obs_pose, prob.obs_point,
                                          prob.obs_uv, max_iters, 2.0, sparse=True)
    dt_lm = time.perf_counter() - t0
    out = {"kind": "port", "seconds": round(dt_lm, 3), "iterations": int(info["iterations"]), "cost0": round(float(info["initial_cost"]), 1),
           "final_robust_cost": round(float(info["final_cost"]), 1),
           "device_final_cost": None if target_cost is None else round(float(target_cost), 1),
           "within_1pct_of_device": None if target_cost is None else bool(abs(info["final_cost"] - target_cost) <= 0.01 * target_cost),
           "cores": os.cpu_count(),
           "what": "oracle/ba_ref.py::solve_dense_lm(sparse=True): Levenberg-Marquardt with Ceres' default policy, Huber(2.0) per "
                   "reprojection block, quaternion manifold, sparse normal equations by SuperLU (numpy / scipy on the host cores), "
                   f"{int(max_iters)} iterations as the device solve; costs are 0.5 sum rho(||r||^2)"}
    opt = np.flatnonzero(~prob.pose_const)
    Po, Q, n = len(opt), len(prob.X), len(prob.obs_pose)
    slot = -np.ones(len(prob.q), int); slot[opt] = np.arange(Po)

    def unpack(x):
        q, t = prob.q.copy(), prob.t.copy()
        d = x[:6 * Po].reshape(Po, 6)
        q[opt] = np.stack([ba_ref.quat_plus(prob.q[r], d[i, :3]) for i, r in enumerate(opt)])
        t[opt] = prob.t[opt] + d[:, 3:]
        return q, t, prob.X + x[6 * Po:].reshape(Q, 3)

    def fun(x):
        q, t, X = unpack(x)
        return ba_ref.reproj_residual_jacobian(prob.obs_pose, prob.obs_point, prob.obs_uv, q, t, X, prob.intr)[0].ravel()

    S = lil_matrix((2 * n, 6 * Po + 3 * Q), dtype=np.int8)
    for i in range(n):
        s = slot[prob.obs_pose[i]]
        if s >= 0:
            S[2 * i:2 * i + 2, 6 * s:6 * s + 6] = 1
        c = 6 * Po + 3 * prob.obs_point[i]
        S[2 * i:2 * i + 2, c:c + 3] = 1
    x0 = np.zeros(6 * Po + 3 * Q)
    t0 = time.perf_counter()
    res = least_squares(fun, x0, jac_sparsity=S.tocsr(), method="trf", loss="huber", f_scale=2.0, max_nfev=scipy_max_nfev)
    dt = time.perf_counter() - t0
    sq = (fun(res.x).reshape(-1, 2) ** 2).sum(1)
    out["scipy_trf"] = {"like_for_like": False, "seconds": round(dt, 3), "nfev": int(res.nfev),
                        "scipy_cost_per_component_huber": round(float(res.cost), 1),
                        "robust_cost_in_the_device_definition": round(0.5 * float(np.sum(ba_ref.huber_rho(sq, 2.0)[0])), 1),
                        "what": f"scipy least_squares(trf, loss='huber' per scalar component, f_scale=2, jac_sparsity), max_nfev={scipy_max_nfev}: "
                                "another objective, not converged - NOT comparable with the device solve"}
    return out


def ba_and_reproject_records(ctx, with_cpu):
    """C3 local-BA solve on the device + the residual kernel's HBM fraction + the C2-size 2D-3D
    association.  Host-inclusive wall times of the `_host` entry points."""
    sys.path.insert(0, str(ROOT / "tests"))
    import ba_scenes
    import reproject_scenes as RS
    pkg = importlib.import_module("opencv-simpleslam_amd")
    S = importlib.import_module("opencv-simpleslam_amd.ba_solver")
    bau = importlib.import_module("opencv-simpleslam_amd.slam.core.ba_utils")
    pnp = importlib.import_module("opencv-simpleslam_amd.slam.core.pnp_utils")
    nat = pkg._native
    import copy
    wmap, kfs, K = ba_scenes.scaled_scene()
    prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(5, 15)), list(range(0, 5)), 5000)
    S.solve_device(copy.deepcopy(prob), 12, 2.0, ctx=ctx)
    ts = []
    for _ in range(5):
        p = copy.deepcopy(prob)
        t0 = time.perf_counter(); summ = S.solve_device(p, 12, 2.0, ctx=ctx); ts.append(time.perf_counter() - t0)
    ba = {"scene": "C3: 10 opt + 5 fixed KFs, 5000 points (SURVEY 8(d))", "observations": int(len(prob.obs_pose)),
          "device_lm_ms": round(float(np.median(ts)) * 1e3, 3), "iterations": summ.iterations,
          "cost": [round(summ.initial_cost, 1), round(summ.final_cost, 1)]}
    # residual + Jacobian kernel against the HBM roofline (8 M observations, past the Infinity Cache)
    L, P = nat.lib(), nat.ptr
    rng = np.random.default_rng(0)
    n, Pn = 8_000_000, 15
    Qn = n // 6
    q = rng.standard_normal((Pn, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d = {k: ctx.upload(v) for k, v in dict(
        pi=rng.integers(0, Pn, n).astype(np.int32), xi=rng.integers(0, Qn, n).astype(np.int32),
        uv=rng.uniform(0, 1000, (n, 2)), q=q, t=rng.standard_normal((Pn, 3)),
        X=rng.standard_normal((Qn, 3)) + [0, 0, 12.0], intr=np.array([718.856, 718.856, 607.19, 185.2])).items()}
    o = {k: ctx.malloc(n * w * 8) for k, w in dict(r=2, Jq=8, Jt=6, JX=6).items()}

    def run():
        nat.check(L.sslam_ba_residual_jacobian_dev(ctx.handle, n, P(d["pi"]), P(d["xi"]), P(d["uv"]), Pn, P(d["q"]),
                                                   P(d["t"]), Qn, P(d["X"]), P(d["intr"]), P(o["r"]), P(o["Jq"]),
                                                   P(o["Jt"]), P(o["JX"])))
    for _ in range(3):
        run()
    ctx.sync(); ctx.timer_start()
    for _ in range(10):
        run()
    us = ctx.timer_stop() / 10 * 1e3
    for p_ in list(d.values()) + list(o.values()):
        ctx.free(p_)
    gbs = n * 200 / us / 1e3
    ba["residual_kernel"] = {"observations": n, "us": round(us, 1), "algorithmic_bytes_per_obs": 200,
                             "achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4)}
    if with_cpu:
        ba["cpu_baseline"] = ba_cpu_baseline(prob, 12, target_cost=summ.final_cost)
    sc = RS.make_case(11, 5000, 2048, 12.0, 0.8, False)
    args = (sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"])
    pnp.reproject_and_match_2d3d(*args)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m = pnp.reproject_and_match_2d3d(*args); ts.append(time.perf_counter() - t0)
    rp = {"scene": "5000 map points x 2048 keypoints (C2 size)", "wall_ms": round(float(np.median(ts)) * 1e3, 3),
          "matches": int(len(m.kp_indices)),
          "what": "reproject_and_match_2d3d, host arrays in and out; wall_ms with the reference's dict-of-objects Map "
                  "(a Python walk rebuilds the arrays per call), soa_map_wall_ms with the overlay's array-backed Map "
                  "(slam/core/landmark_utils.py, device mirror, sslam_reproject_match_dev)"}
    lm = importlib.import_module("opencv-simpleslam_amd.slam.core.landmark_utils")
    soa = lm.Map.from_reference(sc["wmap"])
    sargs = (soa,) + args[1:]
    m2 = pnp.reproject_and_match_2d3d(*sargs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m2 = pnp.reproject_and_match_2d3d(*sargs); ts.append(time.perf_counter() - t0)
    rp["soa_map_wall_ms"] = round(float(np.median(ts)) * 1e3, 3)
    rp["soa_map_matches"] = int(len(m2.kp_indices))
    return ba, rp


# --------------------------------------------------------------------------- self launch
def _self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process
    (nothing in this process has touched the GPU) and relay rank 0's JSON line."""
    port = int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 1000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_extras:
        cmd.append("--no-extras")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    res = subprocess.run(cmd, env=env, capture_output=True, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stderr.write(res.stdout[-4000:] + "\n" + res.stderr[-8000:])
        raise SystemExit(res.returncode or 1)
    print(lines[-1], flush=True)
    raise SystemExit(0)


def calibrated_confidence_heads(sd, feats, ctx):
    """Random-init token-confidence heads cannot be confident about anything, so the depth / width control never fires on them.
    For the early_stop leg the BIAS of head i is set from data: one pair of the stream runs i + 1 layers (debug hook), the head's
    logits over both images' token states are read back, and the bias is shifted so that the fraction EARLY_STOP_CONFIDENT[i] of
    them lies above the layer's confidence threshold (0.8 + 0.1 exp(-4 i / 9), upstream lightglue.py confidence_threshold).  The
    weights stay the seeded random ones; every kernel of the control path (lg_token_heads -> lg_decide -> lg_gather) then runs on
    genuinely data-dependent decisions."""
    (xy0, d0), (xy1, d1) = feats
    n = min(len(xy0), len(xy1))
    if n < 64:
        return sd
    probe = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP(
        sd, max_kpts=MAX_KPTS, ctx=ctx, depth_confidence=-1.0, width_confidence=-1.0)
    sd = dict(sd)
    for i in range(8):
        probe.debug_layers(i + 1, False)
        probe.match(xy0, d0, xy1, d1, min_conf=0.0)
        x = probe.debug_read(0, (2, probe.capacity, 256))
        w = np.asarray(sd[f"token_confidence.{i}.token.0.weight"], np.float64).reshape(-1)
        z = np.concatenate([x[0, :len(xy0)], x[1, :len(xy1)]]).astype(np.float64) @ w
        thr = min(max(0.8 + 0.1 * np.exp(-4.0 * i / 9), 0.0), 1.0)
        frac = EARLY_STOP_CONFIDENT[min(i, len(EARLY_STOP_CONFIDENT) - 1)]
        sd[f"token_confidence.{i}.token.0.bias"] = np.asarray([np.log(thr / (1 - thr)) - np.quantile(z, 1.0 - frac)], np.float32)
    probe.debug_layers(9, False)
    probe.close()
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=140)         # x 24 frames: a timed region of ~3 s
    ap.add_argument("--warmup", type=int, default=8)          # a multiple of the 4-round input pool x 2 record sets: every graph key is captured
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-f32 / BA / reproject legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world == 1 and args.gpus > 1:
        _self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cv2_classes_leg = None
    if rank == 0 and world == 1 and not args.no_extras and os.environ.get("SSLAM_BENCH_FORCE_DIST") != "1":
        try:                                     # (child processes: before this one touches the GPU)
            cv2_classes_leg = dropin_cv2_classes_leg()
        except Exception as e:                   # never lose the headline line to an auxiliary leg
            cv2_classes_leg = {"error": repr(e)}

    dist = torch = None
    # SSLAM_BENCH_FORCE_DIST=1 (tests): take the N > 1 branches - process group, collective barrier, max-reduce of the
    # time, the pipeline's collation path - with ONE rank, so the RCCL code has run on a single-GPU box
    distributed = world > 1 or os.environ.get("SSLAM_BENCH_FORCE_DIST") == "1"
    # ONE collation path (frame_shard.FrameStreamPipeline + a `comm`): RCCL driven directly (opencv-simpleslam_amd/rccl.py) -
    # the library, and with it the SYSTEM HIP runtime, loads first; torch serves the CPU-side rendezvous over gloo (the
    # 128-byte communicator id, the barrier, the max-reduce of the times) and never initialises its GPU side.
    # SSLAM_DIST_BACKEND=gloo is a test-only mode for ranks that share a GPU: the same choreography with the rows
    # exchanged through the host (frame_shard.GlooRowsComm).
    backend = os.environ.get("SSLAM_DIST_BACKEND", "rccl")
    comm = None
    rccl_fallback = None
    if distributed:
        if backend not in ("rccl", "gloo"):
            raise SystemExit(f"SSLAM_DIST_BACKEND={backend!r}: 'rccl' (default) or 'gloo' (ranks sharing a GPU, tests)")
        pkg_ = importlib.import_module("opencv-simpleslam_amd")
        if pkg_._native.device_count() < 1:
            raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
        device_index = local_rank % pkg_._native.device_count()
        pkg_._native.default_context(device_index)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29621")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if backend == "rccl":
            rccl = importlib.import_module("opencv-simpleslam_amd.rccl")

            def _exchange(payload):
                box = [payload]
                dist.broadcast_object_list(box, src=0)
                return box[0]

            def _all_agree(err):                 # every rank takes the same branch: one rank's failure is everybody's
                flags = [None] * world
                dist.all_gather_object(flags, err)
                return next((f"rank {r}: {e}" for r, e in enumerate(flags) if e), None)
            try:
                rccl.lib()
                err = None
            except Exception as e:               # (librccl.so missing / not loadable on this node)
                err = repr(e)
            rccl_fallback = _all_agree(err)
            if rccl_fallback is None:
                try:
                    comm = rccl.RcclComm.create(rank, world, _exchange)
                    err = None
                except Exception as e:
                    err = repr(e)
                rccl_fallback = _all_agree(err)
            if rccl_fallback is not None:
                # never lose the scaling line to the collation transport: the exchange is two 100-KB gathers per round, the
                # frames shard with no data-path collective either way; the JSON line says which transport ran and why
                if rank == 0:
                    print(f"[bench] RCCL communicator not available ({rccl_fallback}); collation through the host over gloo", file=sys.stderr, flush=True)
                backend = "gloo"
                comm = None
        if backend == "gloo":
            comm = importlib.import_module("opencv-simpleslam_amd.frame_shard").GlooRowsComm(rank, world)

    pkg = importlib.import_module("opencv-simpleslam_amd")
    nat = pkg._native
    W = importlib.import_module("opencv-simpleslam_amd.weights")
    AlikedHIP = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
    LightGlueHIP = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
    fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
    if nat.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if not distributed:
        device_index = 0

    # extractor / matcher instances, one HIP stream (context) each.  The matcher runs BATCHES of
    # pairs, every launch over the whole batch: two matcher streams overlap the under-filled tail
    # of one batch with the head of the next; the extractors' short kernels fill in beside them.
    # r03: with the batched ALIKED entry ONE extractor stream is best (2 / 3 streams: 948 / 870 frames/s against 970)
    N_EXT = int(os.environ.get("SSLAM_BENCH_NE", 1))
    N_MAT = int(os.environ.get("SSLAM_BENCH_NM", 3))     # r03, beside ONE extractor stream: 2 / 3 / 4 matcher streams 967 / 979 / 954 frames/s
    ctx_e = [nat.Context(device_index) for _ in range(N_EXT)]
    ctx_m = [nat.Context(device_index) for _ in range(N_MAT)]
    sd_a, sd_l = W.random_aliked_state_dict(0), W.random_lightglue_state_dict(0)
    # extractors take EXT_FRAMES frames per call (one launch sequence per chunk, sslam_aliked_extract_batch_dev)
    EXT_FRAMES = int(os.environ.get("SSLAM_BENCH_EF", 8))
    dets = [AlikedHIP(sd_a, max_num_keypoints=MAX_KPTS, max_h=H_IMG, max_w=W_IMG, ctx=c, max_frames=EXT_FRAMES) for c in ctx_e]
    mats = [LightGlueHIP(sd_l, max_kpts=MAX_KPTS, ctx=c, max_pairs=BATCH_PAIRS) for c in ctx_m]
    if os.environ.get("SSLAM_BIG_GEMM"):                 # A/B hook (scripts/): linear-kernel form of the batched forward
        for mat in mats:
            mat.debug_big_gemm(int(os.environ["SSLAM_BIG_GEMM"]))
    plan = fs.ShardPlan(world, rank, FRAMES_PER_RANK)
    USE_GRAPHS = os.environ.get("SSLAM_BENCH_GRAPHS", "1") != "0"      # A/B hook (scripts/): plain launches instead of cached hipGraphs
    pipe = fs.FrameStreamPipeline(dets, mats, plan, MAX_KPTS, MIN_CONF, batch_pairs=BATCH_PAIRS,
                                  collate_always=distributed, comm=comm, use_graphs=USE_GRAPHS)
    c0 = ctx_e[0]

    # synthetic stream, resident in HBM: a pool of rounds that the timed loop cycles through
    n_pool = 4
    pool = [c0.upload(np.stack([noise_frame(f) for f in plan.frames(r)])) for r in range(n_pool)]

    def barrier(p_=None):
        pipe.sync()
        if p_ is not None and p_ is not pipe:
            p_.sync()                                            # (a leg's own matcher streams)
        if distributed:
            dist.barrier()                                       # (gloo, host side; the device is idle after pipe.sync())

    col = nat.Context(device_index)                          # collector stream: stamps the end of every round
    stamps = []

    def timed_rounds(frames_pool, steps, warmup, stamp=False, p_=None):
        p_ = p_ or pipe
        for i in range(warmup):
            p_.round(frames_pool[i % len(frames_pool)], H_IMG, W_IMG, C_IMG)
        barrier(p_)
        if stamp:
            while len(stamps) < steps + 1:
                stamps.append(col.timing_event())
            col.record(stamps[0])
        t0 = time.perf_counter()
        for i in range(steps):
            p_.round(frames_pool[(warmup + i) % len(frames_pool)], H_IMG, W_IMG, C_IMG)
            if stamp:
                # the round is complete when its matches are: the collector waits for the round's batch events and
                # records a timing event - no host synchronisation inside the timed region
                ps = p_.last_set
                for ev in p_.ev_batch[ps][:p_.n_batches[ps]]:
                    col.wait(ev)
                col.record(stamps[i + 1])
        barrier(p_)
        return time.perf_counter() - t0

    timed_rounds(pool, 0, args.warmup)                      # warm-up (untimed; also captures the hipGraphs)
    dt = timed_rounds(pool, args.steps, 0, stamp=True)
    col.sync()
    step_ms = np.array([nat.Context.elapsed_ms(stamps[i], stamps[i + 1]) for i in range(args.steps)])
    if pipe.range_overflow():
        raise SystemExit("bench.py: a LightGlue activation left the fp16 range of the split-precision path inside the "
                         "timed region - the matches are not fp32-grade, the number is void")
    # the dominant kernel inside the running pipeline: a few more rounds with every attention launch
    # bracketed by HIP events on its matcher stream (event records are host-side, so these rounds run
    # un-graphed; the bracket includes time the launch waits for CUs held by other streams' kernels)
    for mat in mats:
        mat.profile(True)
    timed_rounds(pool, min(8, max(2, args.steps // 4)), 0)
    attn_ms, attn_n = 0.0, 0
    for mat in mats:
        mat.profile(False)
        ms_, n_ = mat.profile_read()
        attn_ms += ms_; attn_n += n_
    info = pipe.infos()
    # Kernel-level figure for the roofline: one full batch replayed on ONE stream with the other
    # streams idle.  In the timed region several streams share the chip, so the HIP-event bracket
    # of a launch there also contains the time its blocks wait for CUs held by other streams'
    # kernels; the single-stream bracket is the kernel's own duration (it is what rocprofv3
    # --kernel-trace reports for the kernel in either mode).  Both are printed.
    P = min(BATCH_PAIRS, FRAMES_PER_RANK - 1)
    K = MAX_KPTS
    sb = pipe.last_set * FRAMES_PER_RANK                 # record set of the last round
    pairs = [(pipe.xy_ptr(sb + s - 1), pipe.desc_ptr(sb + s - 1), K, pipe.xy_ptr(sb + s), pipe.desc_ptr(sb + s), K,
              pipe.count_ptr(sb + s - 1), pipe.count_ptr(sb + s)) for s in range(1, P + 1)]
    m0 = mats[0]
    m0.profile(True)
    for rep in range(3):
        m0.match_batch_dev(pairs, pipe.ij + K * 8, pipe.msc + K * 4, pipe.info + 16, K, min_conf=MIN_CONF)
    m0.ctx.sync()
    m0.profile(False)
    iso_ms, iso_n = m0.profile_read()
    t0 = time.perf_counter()
    for rep in range(3):
        m0.match_batch_dev(pairs, pipe.ij + K * 8, pipe.msc + K * 4, pipe.info + 16, K, min_conf=MIN_CONF)
    m0.ctx.sync()
    lg_batch_ms = (time.perf_counter() - t0) / 3 * 1e3

    # second input of SURVEY 8(d): the structured (low-pass, translating) stream, same pipeline
    spool = [c0.upload(np.stack([structured_frame(f) for f in plan.frames(r)])) for r in range(2)]
    s_steps = max(2, args.steps // 4)
    s_dt = timed_rounds(spool, s_steps, 2)
    s_info = pipe.infos()

    # the auxiliary legs run at N = 1 only (they build pipelines of their own; at N > 1 every GPU-second of the lease
    # belongs to the scaling number)
    extras = not args.no_extras and not distributed
    plane_frames = plan.frames_per_round()

    # early stop + point pruning under load: same pipeline, matchers whose token-confidence / matchability heads are
    # biased (random-init weights cannot learn to be confident) so that the device-side depth / width control
    # (lg_token_heads -> lg_decide -> lg_gather) actually stops pairs early and compacts token sets
    e_dt = e_info = None
    if extras:
        ctx_x = [nat.Context(device_index) for _ in range(N_MAT)]
        sd_e = calibrated_confidence_heads(W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=EARLY_STOP_MATCH_BIAS, conf_gain=16.0),
                                           pipe.features()[:2], ctx_x[0])
        mats_e = [LightGlueHIP(sd_e, max_kpts=MAX_KPTS, ctx=c, max_pairs=BATCH_PAIRS) for c in ctx_x]
        pipe_e = fs.FrameStreamPipeline(dets, mats_e, plan, MAX_KPTS, MIN_CONF, batch_pairs=BATCH_PAIRS)
        e_steps = max(4, args.steps // 4)
        e_dt = timed_rounds(spool, e_steps, 4, p_=pipe_e)
        e_info = pipe_e.infos()
        for m_ in mats_e:
            m_.close()

    # matched frames through the SAME batched pipeline (`value`'s random-init networks emit no match, so
    # emit / compaction / the per-pair outputs never carry anything in it): LightGlue weights whose assignment head is sharp
    # enough to match (the parity tests' `match_gain` set), and behind every batched extraction - on the extractor's stream,
    # INSIDE the timed region - each frame's record is overwritten with the next frame of a synthetic matched chain
    # (tests/lg_inputs.py::PlantedBatchExtractor; tests/test_bench_config_gpu.py holds exactly this form to the oracle)
    planted = None
    if extras:
        try:
            sys.path.insert(0, str(ROOT / "tests"))
            import lg_inputs
            chain = lg_inputs.make_chain(FRAMES_PER_RANK, MAX_KPTS, seed=7, noise=0.035, drop=0.1)
            sd_p = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
            mats_p = [LightGlueHIP(sd_p, max_kpts=MAX_KPTS, ctx=nat.Context(device_index), max_pairs=BATCH_PAIRS) for _ in range(N_MAT)]
            pipe_p = fs.FrameStreamPipeline(dets, mats_p, plan, MAX_KPTS, MIN_CONF, batch_pairs=BATCH_PAIRS)
            planters = [lg_inputs.PlantedBatchExtractor(d_, chain, MAX_KPTS) for d_ in dets]
            try:
                m_steps = max(4, args.steps // 4)
                m_dt = timed_rounds(pool, m_steps, 4, p_=pipe_p)
                m_info = pipe_p.infos()
            finally:
                for pl_ in planters:
                    pl_.restore()
            planted = {"value": round(m_steps * plane_frames / m_dt, 2), "unit": "frames/s", "steps": m_steps,
                       "matches_per_pair": round(float(m_info[:, 0].mean()), 1),
                       "matches_per_pair_min_max": [int(m_info[:, 0].min()), int(m_info[:, 0].max())],
                       "lightglue_layers_executed": int(m_info[-1, 1]), "kpts_matched": [int(m_info[-1, 2]), int(m_info[-1, 3])],
                       "what": "the pipeline of `value` (same extractor instance, same streams, graphs, batch sizes) on frames that MATCH: "
                               "LightGlue weights with a sharp assignment head and every frame's record overwritten, behind its batched "
                               "extraction on the extractor's stream inside the timed region, with the next frame of a synthetic matched "
                               "chain (2048 keypoints, 81 % co-visible) - all nine layers run, every pair emits hundreds of matches; "
                               "tests/test_bench_config_gpu.py holds this form to the oracle at this size"}
            for m_ in mats_p:
                m_.close()
        except Exception as e:                           # never lose the headline line to an auxiliary leg
            planted = {"error": repr(e)}

    # exact-fp32 leg (precision 0): same pipeline, every contraction on v_mfma_f32_32x32x2_f32
    x_dt = None
    if extras:
        for mat in mats:
            mat.set_precision("f32")
        x_steps = max(2, args.steps // 8)
        x_dt = timed_rounds(pool, x_steps, 1)
        for mat in mats:
            mat.set_precision("f16x3p1")                     # (the default)

    # the other split form, "f16x3" (three MFMAs per product in P.V too - the default until r04; profiles/r05_flip_soak.md): same
    # pipeline, and the attention launches of one batch bracketed on an otherwise idle GPU like the headline's roofline figure
    p1 = None
    if extras:
        for mat in mats:
            mat.set_precision("f16x3")
        q_steps = max(4, args.steps // 4)
        q_dt = timed_rounds(pool, q_steps, 2)
        q_info = pipe.infos()
        m0.profile(True)
        for rep in range(3):
            m0.match_batch_dev(pairs, pipe.ij + K * 8, pipe.msc + K * 4, pipe.info + 16, K, min_conf=MIN_CONF)
        m0.ctx.sync(); m0.profile(False)
        q_ms, q_n = m0.profile_read()
        q_tf = attention_flops(int(q_info[-1, 2]), int(q_info[-1, 3])) * P / (q_ms / max(q_n, 1) * 1e-3) / 1e12 if q_n else None
        p1 = {"value": round(q_steps * plane_frames / q_dt, 2), "unit": "frames/s", "steps": q_steps,
              "attention": {"avg_launch_us": round(q_ms / max(q_n, 1) * 1e3, 2), "achieved": round(q_tf, 2) if q_tf else None,
                            "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(q_tf / F16_MFMA_PEAK_TFLOPS, 4) if q_tf else None,
                            "executed_mfma_frac": round(3 * q_tf / F16_MFMA_PEAK_TFLOPS, 4) if q_tf else None},
              "what": "same pipeline with sslam_lightglue_set_precision(lg, 1): three MFMAs per product in P.V as well (24 instead of 20 "
                      "MFMAs per 32-key sub-step) - the default until r04; the same match indices as the shipped form over 131 199 oracle "
                      "matches, score error 4.4e-5 against 1.06e-4, token states 4e-6 from exact against 2.4e-5 "
                      "(profiles/r05_flip_soak.md) - selectable, never `value`"}
        for mat in mats:
            mat.set_precision("f16x3p1")

    # PCIe-inclusive leg: the same pipeline with the host in the loop - every round's frames come up from page-locked
    # host memory and its {info, pairs} go back, on copy streams of their own, double-buffered like the record sets
    pcie = None
    if extras:
        B, K = FRAMES_PER_RANK, MAX_KPTS
        fbytes = H_IMG * W_IMG * C_IMG
        up, down = nat.Context(device_index), nat.Context(device_index)
        host_in = c0.host_alloc(n_pool * B * fbytes).reshape(n_pool, B * fbytes)
        for r in range(n_pool):
            host_in[r] = np.stack([noise_frame(f) for f in plan.frames(r)]).reshape(-1)
        chunks = [c0.malloc(B * fbytes) for _ in range(2)]
        host_out = [c0.host_alloc(B * 16 + B * K * 8) for _ in range(2)]
        ev_up, ev_down = [up.event(), up.event()], [down.event(), down.event()]

        def pcie_round(i):
            q = pipe.rounds & 1                              # record / output set of the round about to be enqueued
            if i >= 2:
                for s_ in range(B):
                    up.wait(pipe.ev_ext[q][s_])              # the extracts that last read this chunk buffer
            up.h2d_async(chunks[q], host_in[i % n_pool])
            up.record(ev_up[q])
            for d_ in pipe.dets:
                d_.ctx.wait(ev_up[q])
            if i >= 2:
                for m_ in pipe.mats:
                    m_.ctx.wait(ev_down[q])                  # the read-back of the outputs this round overwrites
            pipe.round(chunks[q], H_IMG, W_IMG, C_IMG)
            for ev in pipe.ev_batch[q][:pipe.n_batches[q]]:
                down.wait(ev)
            down.d2h_async(host_out[q][:B * 16], pipe.info, B * 16)
            down.d2h_async(host_out[q][B * 16:], pipe.ij, B * K * 8)
            down.record(ev_down[q])

        p_steps = max(4, args.steps // 4)
        for i in range(4):
            pcie_round(i)
        barrier(); up.sync(); down.sync()
        t0 = time.perf_counter()
        for i in range(4, 4 + p_steps):
            pcie_round(i)
        barrier(); up.sync(); down.sync()
        p_dt = time.perf_counter() - t0
        last = host_out[pipe.last_set][:B * 16].view(np.int32).reshape(B, 4)
        pcie = {"value": round(p_steps * B / p_dt, 2), "unit": "frames/s", "steps": p_steps,
                "h2d_bytes_per_round": B * fbytes, "d2h_bytes_per_round": B * 16 + B * K * 8,
                "matches_read_back_last_round": int(last[:, 0].sum()),
                "what": "same pipeline, per round: frames uploaded from page-locked host memory (copy stream) and "
                        "{info, index pairs} read back (second copy stream); never `value`"}
        for q in chunks:
            c0.free(q)

    # the other stated sizes, each with the attention kernel's (kpts4000) / the extraction's (c5) own roofline figure
    sized = {}
    if extras:
        def sized_leg(name, Hh, Ww, K, B, P, steps):
            plan_s = fs.ShardPlan(1, 0, B)
            det_s = [AlikedHIP(sd_a, max_num_keypoints=K, max_h=Hh, max_w=Ww, ctx=nat.Context(device_index), max_frames=min(8, B))]
            own = K != MAX_KPTS
            mats_s = ([LightGlueHIP(sd_l, max_kpts=K, ctx=nat.Context(device_index), max_pairs=P) for _ in range(2)] if own else mats)
            pipe_s = fs.FrameStreamPipeline(det_s, mats_s, plan_s, K, MIN_CONF, batch_pairs=P)
            rng = np.random.default_rng(99)
            pool_s = [c0.upload(rng.integers(0, 256, (B, Hh, Ww, 3), dtype=np.uint8)) for _ in range(2)]
            def rounds(n):
                for i in range(n):
                    pipe_s.round(pool_s[i % 2], Hh, Ww, 3)
            rounds(4); barrier(pipe_s)
            t0 = time.perf_counter(); rounds(steps); barrier(pipe_s)
            dt_s = time.perf_counter() - t0
            info_s = pipe_s.infos()
            # extraction alone (one stream, idle GPU): kernel time per frame against the HBM roofline
            d0 = det_s[0]
            fr = list(range(min(8, B)))
            fb = Hh * Ww * 3
            args_e = ([pool_s[0] + s_ * fb for s_ in fr], Hh, Ww, 3, [pipe_s.xy_ptr(s_) for s_ in fr], [pipe_s.desc_ptr(s_) for s_ in fr],
                      [pipe_s.score + s_ * K * 4 for s_ in fr], [pipe_s.count_ptr(s_) for s_ in fr])
            d0.extract_batch_dev(*args_e, max_kpts=K); d0.ctx.sync(); d0.ctx.timer_start()
            for _ in range(5):
                d0.extract_batch_dev(*args_e, max_kpts=K)
            ext_ms = d0.ctx.timer_stop() / 5 / len(fr)
            # ALIKED runs at the resized size (long side 1024, short side padded to 32): its 0.2 GB / frame at 1024 x 320 scales with
            # that area (x1.8 at C5), the 3-byte input image with its own
            net = lambda hh, ww: 1024 * ((int(round(min(hh, ww) * 1024.0 / max(hh, ww))) + 31) // 32 * 32)      # noqa: E731
            scale = net(Hh, Ww) / float(net(H_IMG, W_IMG))
            rec = {"value": round(steps * B / dt_s, 2), "unit": "frames/s", "steps": steps, "frames_per_step": B,
                   "image": [Ww, Hh], "max_kpts": K, "kpts_matched": [int(info_s[-1, 2]), int(info_s[-1, 3])],
                   "lightglue_layers_executed": int(info_s[-1, 1]),
                   "aliked_ms_per_frame_isolated": round(ext_ms, 4),
                   "aliked_hbm": {"algorithmic_GB_per_frame": round(0.2 * scale, 3), "achieved_GBs": round(0.2 * scale / ext_ms * 1e3, 1),
                                  "peak_GBs": HBM_PEAK_GBS, "frac": round(0.2 * scale / ext_ms * 1e3 / HBM_PEAK_GBS, 4)}}
            if own:
                n0_, n1_ = int(info_s[-1, 2]), int(info_s[-1, 3])
                prs = [(pipe_s.xy_ptr(s_ - 1), pipe_s.desc_ptr(s_ - 1), K, pipe_s.xy_ptr(s_), pipe_s.desc_ptr(s_), K,
                        pipe_s.count_ptr(s_ - 1), pipe_s.count_ptr(s_)) for s_ in range(1, min(P, B - 1) + 1)]
                m_ = mats_s[0]
                m_.profile(True)
                for _ in range(2):
                    m_.match_batch_dev(prs, pipe_s.ij + K * 8, pipe_s.msc + K * 4, pipe_s.info + 16, K, min_conf=MIN_CONF)
                m_.ctx.sync(); m_.profile(False)
                ms_, n_ = m_.profile_read()
                if n_:
                    tf = attention_flops(n0_, n1_) * len(prs) / (ms_ / n_ * 1e-3) / 1e12
                    rec["attention"] = {"pairs_per_launch": len(prs), "avg_launch_us": round(ms_ / n_ * 1e3, 2), "achieved": round(tf, 2),
                                        "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F16_MFMA_PEAK_TFLOPS, 4)}
                for m_ in mats_s:
                    m_.close()
            for q in pool_s:
                c0.free(q)
            det_s[0].close()
            return rec
        for name, cfg in (("c5", (1080, 1920, MAX_KPTS, 8, min(8, BATCH_PAIRS), max(4, args.steps // 12))),
                          ("kpts4000", (H_IMG, W_IMG, 4000, 8, 4, max(4, args.steps // 12)))):
            try:
                sized[name] = sized_leg(name, *cfg)
            except Exception as e:                       # never lose the headline line to an auxiliary leg
                sized[name] = {"error": repr(e)}

    times = np.array([dt, s_dt, x_dt or 0.0, e_dt or 0.0])
    if distributed:
        t = torch.tensor(times, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = t.cpu().numpy()
    dt_max, s_dt_max, x_dt_max, e_dt_max = (float(v) for v in times)

    if rank == 0:
        frames_total = args.steps * plan.frames_per_round()
        n0, n1, stop = int(info[-1, 2]), int(info[-1, 3]), int(info[-1, 1])
        per_launch = attention_flops(n0, n1) * P
        ach = per_launch / (iso_ms / max(iso_n, 1) * 1e-3) / 1e12 if iso_n else None
        ach_region = attention_flops(n0, n1) * BATCH_PAIRS / (attn_ms / max(attn_n, 1) * 1e-3) / 1e12 if attn_n else None
        traffic, traffic_src = _pmc_traffic()
        out = {
            "metric": "frames/sec ALIKED+LightGlue @1241x376",
            "value": round(frames_total / dt_max, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "timed_region_s": round(dt_max, 3),
            # GPU-side span of the timed region on rank 0: first to last round stamp (device timing events), so the line
            # corroborates its own busy time when an external utilisation sampler reads nothing
            "gpu_busy_s": round(float(step_ms.sum()) / 1e3, 3),
            "step_ms": {"p10": round(float(np.percentile(step_ms, 10)), 3), "p50": round(float(np.percentile(step_ms, 50)), 3),
                        "p90": round(float(np.percentile(step_ms, 90)), 3), "max": round(float(step_ms.max()), 3),
                        "what": "per-round durations on rank 0 (timing events on a collector stream, no host sync in the region)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # which sources were measured: digest of csrc/ + the C-ABI header (build.py --digest; the GPU box has no .git)
            # and, when the caller passed it along (SSLAM_GIT_HEAD, scripts/final_evidence.sh), the commit
            "build": {"csrc_digest": _source_digest(), "git_head": os.environ.get("SSLAM_GIT_HEAD") or _git_head()},
            "dtype": "f32 (contractions: f16 hi/lo split operands, f32 accumulate; 3 MFMA per product, 2 in attention's P.V = "
                     "precision 'f16x3p1', the default since r05: profiles/r05_flip_soak.md, r06_flip_soak.md)",
            "data": "synthetic",
            "config": {"workload": "C2/C4: synthetic 1241x376x3 uint8 frame stream, ALIKED-n16 extract + "
                                   "LightGlue match (t-1,t), 2048 kpts/frame, min_conf 0.7, random-init weights; "
                                   "inputs resident in HBM before the timed region, features and matches left on the "
                                   "device (the `pcie` leg times the same pipeline with per-round H2D of the frames and "
                                   "D2H of {count, pairs})",
                       "frames_per_step_per_gpu": FRAMES_PER_RANK, "max_kpts": MAX_KPTS,
                       "lightglue_layers_executed": stop, "kpts_matched": [n0, n1],
                       # depth is data dependent (early stop) - layers executed over the pairs of the last round
                       "lightglue_layers_histogram": {str(int(k)): int(v) for k, v in
                                                      zip(*np.unique(info[info[:, 2] > 0, 1], return_counts=True))},
                       "pairs_per_lightglue_launch": BATCH_PAIRS, "frames_per_aliked_launch": pipe.EF,
                       "parallelism": f"frame-shard x{world}; per GPU {N_EXT} extractor + {N_MAT} matcher streams, "
                                      f"ALIKED in batches of {pipe.EF} frames, LightGlue in batches of {BATCH_PAIRS} pairs"
                                      + (f"; collation: {'RCCL directly' if backend == 'rccl' else 'rows through the host over gloo (test mode)'}"
                                         if distributed else ""),
                       # N > 1: the size of the communicator the collation ran on, as the communicator reports it
                       "rccl_ranks": (comm.count() if distributed and backend == "rccl" else None),
                       "collation_backend": (("rccl (direct)" if backend == "rccl" else
                                              "gloo (host round trip" + (f"; RCCL not available: {rccl_fallback})" if rccl_fallback else ", test mode)")) if distributed else None)},
            # achieved = ALGORITHMIC flops (8 n0 n1 256 per pair, x pairs per launch) / HIP-event launch
            # duration on an otherwise idle GPU; the kernel issues 20 v_mfma_f32_32x32x16_f16 per 32-key sub-step for 8
            # algorithmic products' worth (3 per product in K.Q^T, 2 in P.V): executed = 2.5x
            "roofline": {"bound": "mfma", "kernel": "lg_attention_asm_p1_kernel (hand-scheduled gfx950 assembly; v_mfma_f32_32x32x16_f16: 3 per product in K.Q^T, 2 in P.V)",
                         "achieved": round(ach, 2) if ach else None, "peak": F16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / F16_MFMA_PEAK_TFLOPS, 4) if ach else None,
                         # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
                         # scripts/pmc_traffic.sh); not re-measured here: PMC collection needs rocprofv3
                         "traffic": traffic, "traffic_source": traffic_src,
                         "executed_mfma_frac": round(2.5 * ach / F16_MFMA_PEAK_TFLOPS, 4) if ach else None,
                         "pairs_per_launch": P, "launches_timed": iso_n,
                         "avg_launch_us": round(iso_ms / max(iso_n, 1) * 1e3, 2),
                         # the same launches inside the running pipeline (other streams' kernels share the chip)
                         "in_pipeline_launches": attn_n,
                         "in_pipeline_avg_bracket_us": round(attn_ms / max(attn_n, 1) * 1e3, 2),
                         "in_pipeline_frac": round(ach_region / F16_MFMA_PEAK_TFLOPS, 4) if ach_region else None,
                         "lightglue_batch_ms_isolated": round(lg_batch_ms, 3),
                         "pipeline_algorithmic_tflops": round(
                             frames_total / dt_max * (lightglue_gflop(min(n0, n1), stop) + ALIKED_GFLOP_PER_FRAME)
                             / 1e3 / world, 2)},
        }
        out["structured_input"] = {
            "value": round(s_steps * plan.frames_per_round() / s_dt_max, 2), "unit": "frames/s", "steps": s_steps,
            "what": "same pipeline on the 9x9-box low-pass noise translating 3 px/frame (SURVEY 8(d))",
            "matches_last_pair": int(s_info[-1, 0]), "lightglue_layers_executed": int(s_info[-1, 1]),
            "lightglue_layers_histogram": {str(int(k)): int(v) for k, v in
                                           zip(*np.unique(s_info[s_info[:, 2] > 0, 1], return_counts=True))},
            "kpts_matched": [int(s_info[-1, 2]), int(s_info[-1, 3])]}
        if e_info is not None:
            e_steps = max(4, args.steps // 4)
            ok = e_info[:, 2] > 0
            pruned = bool(ok.any() and int(e_info[ok, 2:4].min()) < MAX_KPTS)
            out["early_stop"] = {
                "value": round(e_steps * plan.frames_per_round() / e_dt_max, 2), "unit": "frames/s", "steps": e_steps,
                "what": f"same pipeline, structured stream, random-init weights whose token-confidence biases are calibrated on one pair "
                        f"of the stream to call {EARLY_STOP_CONFIDENT} of the points confident after layers 0..7 (matchability bias "
                        f"{EARLY_STOP_MATCH_BIAS}): pairs stop early"
                        + (" and points are pruned on the device" if pruned else " (no point was pruned before the stop on these inputs)"),
                "points_pruned": pruned,
                "lightglue_layers_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(e_info[ok, 1], return_counts=True))},
                "kpts_after_pruning_min_max": [int(e_info[ok, 2:4].min()), int(e_info[ok, 2:4].max())] if ok.any() else None}
        if x_dt is not None:
            x_steps = max(2, args.steps // 8)
            out["exact_f32"] = {"value": round(x_steps * plan.frames_per_round() / x_dt_max, 2), "unit": "frames/s",
                                "steps": x_steps, "peak": F32_MFMA_PEAK_TFLOPS,
                                "what": "same pipeline, every contraction on v_mfma_f32_32x32x2_f32 (precision 0)"}
        if planted is not None:
            out["planted_matches"] = planted
        if pcie is not None:
            out["pcie"] = pcie
        if p1 is not None:
            out["f16x3"] = p1
        out.update(sized)
        if extras:
            try:
                out["ba"], out["reproject"] = ba_and_reproject_records(c0, with_cpu=not args.no_cpu_baseline)
            except Exception as e:                       # never lose the headline line to an auxiliary leg
                out["ba"] = {"error": repr(e)}
        if extras:
            try:
                out["dropin"] = dropin_leg()
            except Exception as e:                       # never lose the headline line to an auxiliary leg
                out["dropin"] = {"error": repr(e)}
            if "error" not in out["dropin"] and cv2_classes_leg is not None:
                out["dropin"]["cv2_classes"] = cv2_classes_leg
                v, c = out["dropin"]["value"], cv2_classes_leg.get("value")
                if c:
                    cv2_classes_leg["vs_duck_types"] = round(c / v, 3)      # (against the in-process leg above; `vs_duck_types_same_conditions`: against its partner child)
        if not args.no_cpu_baseline and world == 1:        # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
