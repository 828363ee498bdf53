"""Feature extraction + matching entry points backed by the HIP ALIKED /
LightGlue kernels.

Drop-in for the LightGlue path of the reference's slam/core/features_utils.py:
same names, arguments, return types and error behaviour

    init_feature_pipeline(args)                      features_utils.py:18
    feature_extractor(args, img, detector)           features_utils.py:85
    feature_matcher(args, kp0, kp1, des0, des1, m)   features_utils.py:109
    filter_matches_ransac(kp1, kp2, matches, thresh) features_utils.py:185
    detect_and_match(img1, img2, det, m, args)       features_utils.py:250   (legacy pair entry)

`args` is the CLI namespace of slam/monocular/main_revamped.py (fields read:
use_lightglue, max_features, min_conf).  Keypoints are `cv2.KeyPoint` when
OpenCV is importable, otherwise the duck type in .types (only `.pt` is read
downstream); matches likewise `cv2.DMatch` / duck type.

The OpenCV ORB/SIFT/AKAZE + BF/FLANN branch of the reference (cv2 C++, config 1
"plumbing") is delegated to cv2 when it is installed and raises otherwise: this
package accelerates the LightGlue path only.  There is no CPU fallback for the
LightGlue path: without the HIP library / an MI355X it raises.
"""
from __future__ import annotations

import logging
import os
from typing import List

import numpy as np

import ctypes as _C
import weakref

from .types import (DMatch, KeyPoint, KeyPointList, HAVE_CV2, keypoint_shells, keypoints_from_xy, match_shells,
                    matches_from_ij, xy_from_keypoints)
from ... import _native, weights as _weights
from ...aliked import AlikedHIP
from ...lightglue import LightGlueHIP

_log = logging.getLogger("opencv_simpleslam_amd")

# Optional upstream checkpoints (no network here): set these to .pth paths
ENV_ALIKED = "SSLAM_ALIKED_WEIGHTS"          # aliked-n16.pth
ENV_LIGHTGLUE = "SSLAM_LIGHTGLUE_WEIGHTS"    # aliked_lightglue.pth
MAX_IMAGE_H = int(os.environ.get("SSLAM_MAX_IMAGE_H", 2160))
MAX_IMAGE_W = int(os.environ.get("SSLAM_MAX_IMAGE_W", 4096))


ENV_ALLOW_RANDOM = "SSLAM_ALLOW_RANDOM_WEIGHTS"
ENV_RANDOM_LG_ARGS = "SSLAM_RANDOM_LIGHTGLUE_ARGS"   # e.g. "seed=1,match_gain=4.0,match_bias=3.0": arguments of the random fallback


def _state_dict(env_name, random_fn, what):
    """Checkpoint named by `env_name`, else - ONLY when SSLAM_ALLOW_RANDOM_WEIGHTS=1 - seeded random
    weights of the same architecture, with a WARNING.  The reference never runs untrained: it loads
    `aliked-n16` / `aliked_lightglue` through torch.hub (features_utils.py:25-26) or fails; dropped into
    main_revamped.py a silent random fallback would produce garbage tracks without a word."""
    path = os.environ.get(env_name)
    if path:
        return _weights.load_state_dict(path)
    if os.environ.get(ENV_ALLOW_RANDOM, "") != "1":
        raise RuntimeError(
            f"init_feature_pipeline: no {what} checkpoint - set {env_name} to the upstream .pth file (there is no "
            f"network here to download it as the reference does, features_utils.py:25-26), or set "
            f"{ENV_ALLOW_RANDOM}=1 to run seeded RANDOM weights (synthetic benchmarking / parity tests only)")
    _log.warning("%s: %s unset - running seeded RANDOM-INIT weights (%s=1): matches are meaningless on real imagery",
                 what, env_name, ENV_ALLOW_RANDOM)
    kw = {}
    if env_name == ENV_LIGHTGLUE and os.environ.get(ENV_RANDOM_LG_ARGS):
        # synthetic benchmarking only: a random init whose assignment head is sharp enough to produce matches
        for item in os.environ[ENV_RANDOM_LG_ARGS].split(","):
            k, v = item.split("=")
            kw[k.strip()] = int(v) if k.strip() == "seed" else float(v)
    return random_fn(kw.pop("seed", 0), **kw)


def init_feature_pipeline(args):
    """Instantiate detector & matcher according to CLI arguments -> (detector, matcher)."""
    if args.use_lightglue:
        max_kpts = int(getattr(args, "max_features", 4000))
        ctx = _native.default_context(int(os.environ.get("LOCAL_RANK", 0)) % max(1, _native.device_count()))
        detector = AlikedHIP(_state_dict(ENV_ALIKED, _weights.random_aliked_state_dict, "ALIKED (aliked-n16)"),
                             max_num_keypoints=max_kpts, max_h=MAX_IMAGE_H, max_w=MAX_IMAGE_W, ctx=ctx)
        # the matcher runs on a stream of its own (same device): it reads the extractor's device records directly, ordered
        # behind the extraction by an event, so a match can run while a frame's results are still on their way to the host
        mctx = _native.Context(ctx.device)
        matcher = LightGlueHIP(_state_dict(ENV_LIGHTGLUE, _weights.random_lightglue_state_dict, "LightGlue (aliked_lightglue)"),
                               max_kpts=max_kpts, ctx=mctx)
        matcher._feature_ring = _ring_of(detector)
        matcher._feature_ring.attach_matcher(matcher)
        return detector, matcher
    if not HAVE_CV2:
        raise ImportError("the OpenCV detector/matcher branch needs cv2; this backend accelerates "
                          "the --use_lightglue path only")
    import cv2
    det = getattr(args, "detector", "orb")
    nfeat = int(getattr(args, "max_features", 6000))
    if det == "orb":
        detector = cv2.ORB_create(nfeat)
    elif det == "sift":
        detector = cv2.SIFT_create(nfeatures=nfeat)
    elif det == "akaze":
        detector = cv2.AKAZE_create()
    else:
        raise ValueError(f"Unsupported detector: {det}")
    if getattr(args, "matcher", "bf") == "flann":
        matcher = cv2.FlannBasedMatcher(dict(algorithm=1, trees=5), dict(checks=50))
    else:
        norm = cv2.NORM_HAMMING if det in ("orb", "akaze") else cv2.NORM_L2
        matcher = cv2.BFMatcher(norm, crossCheck=True)
    return detector, matcher


def _convert_lg_kps_to_opencv(xy: np.ndarray) -> List[KeyPoint]:
    return keypoints_from_xy(np.asarray(xy))


def _convert_opencv_to_lg_kps(kps) -> np.ndarray:
    return xy_from_keypoints(kps)


def _convert_lg_matches_to_opencv(ij: np.ndarray) -> List[DMatch]:
    return matches_from_ij(np.asarray(ij))


class _DeviceFeatureRing:
    """Device-resident features of the last few frames `feature_extractor` served (a ring of SLOTS records of
    {xy [K,2], desc [K,128], count}), so that `feature_matcher` on (t-1, t) reads both operands where the
    extractor left them: no keypoint rebuild, no 2 x 1 MB descriptor upload, no staging copy.

    Ownership: the arrays handed to the caller are the caller's - the descriptor array is
    returned READ-ONLY (the reference never writes into it; an in-place edit would silently desynchronise the
    device copy, so numpy refuses it), the keypoint list is a `KeyPointList` that knows when it was edited.  A
    hit needs the SAME descriptor array object (identity, checked through a weak reference so a recycled `id`
    cannot alias) and keypoints equal to the remembered ones; anything else takes the host path.  A slot is
    recycled after SLOTS further extractions (a keyframe's features held longer simply miss).

    Look-ahead (r03): the reference's frame loop calls `feature_matcher(prev, cur)` right after every
    `feature_extractor(cur)` (main_revamped.py:325-330).  Once the ring has seen that pattern it enqueues exactly that
    match - previous record, this record, the last threshold - on the matcher's stream from INSIDE `extract`, ordered
    behind the extraction by an event: the GPU goes from the extraction straight into the match while the host still
    copies the features out and builds its objects, and the `feature_matcher` call that follows only collects the
    result.  A call with other arguments waits for the look-ahead to finish and runs normally; an extraction that
    finds the previous look-ahead unused switches it off until the pattern is seen again (one wasted match)."""
    SLOTS = 4

    def __init__(self, detector):
        self.det = detector
        self.ctx = detector.ctx
        K = self.K = int(detector.max_num_keypoints)
        m = self.ctx.malloc
        # one device block per slot, [count 16 B | xy K x 2 | desc K x 128 | score K]: {count, xy, desc} come back
        # in ONE copy into a page-locked mirror (three pageable copies + the count's own round trip were 140 us)
        self.o_xy, self.o_desc, self.o_score = 16, 16 + K * 8, 16 + K * 8 + K * 512
        self.rec_bytes = self.o_score + K * 4
        self.slots = []
        for _ in range(self.SLOTS):
            base = m(self.rec_bytes)
            self.slots.append(dict(base=base, cnt=base, xy=base + self.o_xy, desc=base + self.o_desc,
                                   score=base + self.o_score, key=None))
        self.pin_rec = self.ctx.host_alloc(self.o_score)
        self.pin_cnt = self.pin_rec[:16].view(np.int32)
        self.pin_xy = self.pin_rec[self.o_xy:self.o_desc].view(np.float32).reshape(K, 2)
        self.pin_desc = self.pin_rec[self.o_desc:self.o_score].view(np.float32).reshape(K, 128)
        self.by_id = {}                  # id(descriptor array) -> entry
        self.turn = 0
        self.img_dev, self.img_cap = 0, 0
        self.tmp_xy = [m(K * 8), m(K * 8)]               # keypoints of an edited list (uploaded per call)
        # match results [info 16 B | pairs K x 2 | RANSAC info 16 B | RANSAC mask K] + scores K: the first four come back
        # in ONE copy (the filter of filter_matches_ransac runs on the device right behind the match, see below)
        KM = (K + 15) // 16 * 16
        out = m(16 + K * 8 + 16 + KM + K * 4)
        self.out_info, self.out_ij = out, out + 16
        self.rs_info, self.rs_mask = out + 16 + K * 8, out + 16 + K * 8 + 16
        self.out_sc = out + 16 + K * 8 + 16 + KM
        self.match_bytes, self.filtered_bytes = 16 + K * 8, 16 + K * 8 + 16 + KM
        self.pin_match = self.ctx.host_alloc(self.filtered_bytes)
        self.pin_info = self.pin_match[:16].view(np.int32)
        self.pin_ij = self.pin_match[16:16 + K * 8].view(np.int32).reshape(K, 2)
        self.pin_rs_info = self.pin_match[16 + K * 8:16 + K * 8 + 16].view(np.int32)
        self.pin_rs_mask = self.pin_match[16 + K * 8 + 16:16 + K * 8 + 16 + K]
        self.ransac_thr = None           # threshold of the last filter_matches_ransac call on a resident match (None: not seen)
        self.filtered = None             # the resident match whose filter already ran on the device: dict(matches=list, kp0, kp1, thr)
        detector.use_graphs(True)        # the slots are a fixed set of buffers: the launch sequence replays as a graph
        self.matcher, self.mctx = None, None
        self.ev_extracted = self.ctx.event()
        self.last_entry = None           # the most recently extracted frame
        self.ahead_on = False            # the prev -> cur pattern has been seen
        self.ahead = None                # outstanding look-ahead: dict(a=entry, b=entry, thr=float)
        self.last_thr = None

    def attach_matcher(self, matcher):
        self.matcher, self.mctx = matcher, matcher.ctx

    def extract(self, img):
        det, ctx = self.det, self.ctx
        if not isinstance(img, np.ndarray):
            img = np.asarray(img)
        if img.dtype != np.uint8:
            raise TypeError("feature extraction expects a uint8 image (cv2.imread output)")
        if img.ndim == 2:
            H, Wd, Cn = img.shape[0], img.shape[1], 1
        elif img.ndim == 3:
            H, Wd, Cn = img.shape
        else:
            raise ValueError(f"unsupported image shape {img.shape}")
        if img.nbytes > self.img_cap:
            if self.img_dev:
                ctx.sync(); ctx.free(self.img_dev)
            self.img_cap = max(img.nbytes, 1241 * 376 * 3)
            self.img_dev = ctx.malloc(self.img_cap)
        sl = self.slots[self.turn % self.SLOTS]
        self.turn += 1
        if sl["key"] is not None:
            # (only if the entry under that id is still THIS slot's: the id of a descriptor array the caller has dropped is
            #  handed out again by the allocator, possibly to a newer frame's array)
            e = self.by_id.get(sl["key"])
            if e is not None and e["slot"] is sl:
                del self.by_id[sl["key"]]
            sl["key"] = None
        K = self.K
        # (the image goes up straight from the caller's pageable array: the runtime's own staged copy, 69 us for
        #  1.4 MB, beats a host copy into a page-locked stage + DMA, 57 + 41 us)
        staged = np.ascontiguousarray(img)   # (a non-contiguous image: this copy must outlive the DMA - it is held until the ctx.sync() below)
        ctx.h2d_async(self.img_dev, staged)  # (pageable source: the runtime stages it before the call returns; page-locked: the DMA reads it in place)
        prev = self.last_entry
        if self.ahead is not None:       # the last look-ahead was never collected: the caller is not in the prev -> cur loop
            self.mctx.sync()             # (it may still read a record this ring is about to recycle)
            self.ahead, self.ahead_on = None, False
        det.extract_dev(self.img_dev, H, Wd, Cn, sl["xy"], sl["desc"], sl["score"], sl["cnt"], max_kpts=K)
        ctx.record(self.ev_extracted)
        ctx.d2h_async(self.pin_rec, sl["base"])
        look = (self.ahead_on and self.matcher is not None and prev is not None and prev["slot"] is not sl
                and prev["n"] > 0 and self.last_thr is not None)
        if look:
            # (this frame's count is only known on the device yet: K bounds it, the matcher clamps to the record's count)
            self.mctx.wait(self.ev_extracted)
            ps = prev["slot"]
            self.matcher.match_dev(ps["xy"], ps["desc"], prev["n"], sl["xy"], sl["desc"], K, self.out_ij, self.out_sc,
                                   self.out_info, min_conf=self.last_thr, m_dev=ps["cnt"], n_dev=sl["cnt"])
            look_filter = self.ransac_thr if self.enqueue_filter_and_readback(ps["xy"], sl["xy"]) else None
        # the GPU needs ~0.5 ms from here: build the frame's KeyPoint objects meanwhile (their coordinates resolve
        # against the array below on first use)
        shells, src = keypoint_shells(K) if keypoint_shells is not None else (None, None)
        ctx.sync()
        del staged                           # the upload is done: the caller's image may change from here on
        n = int(self.pin_cnt[0])
        xy = self.pin_xy[:n].copy(); desc = self.pin_desc[:n].copy()
        desc.setflags(write=False)
        if shells is not None:
            src.xy = xy
            if n < K:
                del shells[n:]
            kps = KeyPointList(shells, xy)
        else:
            kps = KeyPointList(keypoints_from_xy(xy), xy)
        entry = dict(slot=sl, n=n, desc_ref=weakref.ref(desc), xy=xy, prev=prev)
        sl["key"] = id(desc)
        self.by_id[id(desc)] = entry
        self.last_entry = entry
        if look:
            self.ahead = dict(a=prev, b=entry, thr=self.last_thr, filter_thr=look_filter)
        if prev is not None:
            prev["prev"] = None          # (no chains of dead frames)
        return kps, desc

    def enqueue_filter_and_readback(self, xy_a, xy_b):
        """Behind a match on the matcher's stream: the reference's frame loop filters every match with F-matrix RANSAC
        right away (main_revamped.py:118-126) - once that has been seen, the filter runs on the device on the matcher's
        own output (sslam_fmat_ransac_dev: no host round trip, no pixel gather on the host) and its mask rides back with
        {count, pairs} in the same copy."""
        if self.ransac_thr is not None:
            from ... import epipolar
            epipolar.filter_matches_dev(self.mctx, self.K, self.out_info, xy_a, xy_b, self.out_ij, None, self.rs_info,
                                        thresh=self.ransac_thr, confidence=0.99, mask_out_dev=self.rs_mask)
            self.mctx.d2h_async(self.pin_match, self.out_info)
            return True
        self.mctx.d2h_async(self.pin_match[:self.match_bytes], self.out_info)
        return False

    def lookup(self, des, kps, which):
        """(device xy, device desc, n, device count) for a frame this ring still holds, else None."""
        e = self.by_id.get(id(des))
        if e is None or e["desc_ref"]() is not des or len(kps) != e["n"]:
            return None
        xy = kps.pristine_xy() if isinstance(kps, KeyPointList) else None
        if xy is not None and xy is e["xy"]:
            return e["slot"]["xy"], e["slot"]["desc"], e["n"], e["slot"]["cnt"], e
        # another list / an edited one: rebuild the keypoints like the reference does (features_utils.py:65-77);
        # the descriptors on the device are still the ones of `des`
        xy = xy_from_keypoints(kps)
        if np.array_equal(xy, e["xy"]):
            return e["slot"]["xy"], e["slot"]["desc"], e["n"], e["slot"]["cnt"], e
        self.ctx.h2d(self.tmp_xy[which], xy)
        return self.tmp_xy[which], e["slot"]["desc"], e["n"], e["slot"]["cnt"], None      # (edited keypoints: no look-ahead)


def _ring_of(detector):
    ring = getattr(detector, "_feature_ring", None)
    if ring is None:
        ring = detector._feature_ring = _DeviceFeatureRing(detector)
    return ring


def feature_extractor(args, img: np.ndarray, detector):
    """One image -> (list[KeyPoint], descriptors [N,128] float32 unit rows)."""
    if args.use_lightglue:
        # includes the reference's second L2 normalisation (:100); the features also stay on the GPU for the matcher
        return _ring_of(detector).extract(img)
    kp0, des0 = detector.detectAndCompute(img, None)
    if des0 is None:
        return [], []
    return kp0, des0


def _as_numpy_f32(x):
    if hasattr(x, "detach"):                    # torch tensor (the reference accepts both)
        x = x.detach().to("cpu").numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


_last_ring = None        # weak reference to the ring of the last resident match (filter_matches_ransac has no matcher argument)


def _match_resident(ring, matcher, a, b, thr):
    """Both frames are still on the GPU: enqueue the match on their records (or collect the look-ahead that already runs
    on exactly them), read back {count, pairs} -> list[DMatch]."""
    mctx = matcher.ctx
    ahead, ring.ahead = ring.ahead, None
    hit = (ahead is not None and ahead["a"] is a[4] and ahead["b"] is b[4] and ahead["thr"] == thr)
    ring.filtered = None
    if hit:
        with_filter = ahead.get("filter_thr")
    else:
        if ahead is not None:
            mctx.sync()                          # (its outputs share the buffers below)
        matcher.match_dev(a[0], a[1], a[2], b[0], b[1], b[2], ring.out_ij, ring.out_sc, ring.out_info, min_conf=thr,
                          m_dev=a[3], n_dev=b[3])
        # {count, pairs} (and the RANSAC mask when the caller is known to filter) in one copy into page-locked memory
        with_filter = ring.ransac_thr if ring.enqueue_filter_and_readback(a[0], b[0]) else None
    # the prev -> cur pattern of the reference's frame loop: from now on `extract` enqueues this match itself
    ring.last_thr = thr
    ring.ahead_on = b[4] is not None and b[4] is ring.last_entry and a[4] is not None and b[4].get("prev") is a[4]
    # the GPU needs > 1 ms from here: build the DMatch objects meanwhile (indices resolve against the array below)
    shells, src = match_shells(min(a[2], b[2])) if match_shells is not None else (None, None)
    mctx.sync()
    k = int(ring.pin_info[0])
    if k < 0:
        matcher.range_overflow()                # reported here: clear the instance's sticky word
        raise _native.NativeError("feature_matcher: an activation left the fp16 range of the split-precision path "
                                  "(|value| >= 65520); rescale the descriptors or use matcher.set_precision('f32')")
    ij = ring.pin_ij[:k].copy()
    if shells is None:
        out = _convert_lg_matches_to_opencv(ij)
    else:
        src.ij = ij
        del shells[k:]
        out = shells
    global _last_ring
    _last_ring = weakref.ref(ring)
    if a[4] is None or b[4] is None:
        ring.filtered = None                     # (edited keypoint lists: filter_matches_ransac takes the host path)
    elif with_filter is not None:
        # filter_matches_ransac(kp0, kp1, <this list>, <that threshold>) only has to apply the mask that is already here
        ring.filtered = dict(matches=out, n=k, a=a[4], b=b[4], thr=with_filter, kept=int(ring.pin_rs_info[0]),
                             mask=ring.pin_rs_mask[:k].copy())
    else:
        ring.filtered = dict(matches=out, n=k, a=a[4], b=b[4], thr=None)
    return out


def feature_matcher(args, kp0, kp1, des0, des1, matcher):
    """Two frames -> list[DMatch] (LightGlue: ascending queryIdx, score > args.min_conf)."""
    if (des0 is None or des1 is None or kp0 is None or kp1 is None
            or len(kp0) == 0 or len(kp1) == 0 or len(des0) == 0 or len(des1) == 0):
        return []
    if args.use_lightglue:
        thr = float(getattr(args, "min_conf", 0.7))
        ring = getattr(matcher, "_feature_ring", None)
        if ring is not None and isinstance(des0, np.ndarray) and isinstance(des1, np.ndarray):
            a = ring.lookup(des0, kp0, 0)
            b = ring.lookup(des1, kp1, 1) if a is not None else None
            if a is not None and b is not None:
                return _match_resident(ring, matcher, a, b, thr)
        ij, _scores, _stop = matcher.match(_convert_opencv_to_lg_kps(kp0), _as_numpy_f32(des0),
                                           _convert_opencv_to_lg_kps(kp1), _as_numpy_f32(des1), min_conf=thr)
        return _convert_lg_matches_to_opencv(ij)
    matches = matcher.match(des0, des1)
    return sorted(matches, key=lambda m: m.distance)


def _lightglue_detect_and_match(img1, img2, extractor, matcher):
    """Legacy pair entry (reference features_utils.py:233-247; no longer on its main path, still what its own
    tests/test_lightglue_vs_manual.py drives): extract both images, match the feature dicts as they come out of
    `extractor.extract` - they carry 'image_size', so LightGlue normalises the keypoints by the IMAGE size, not by
    their bounding box - and return every match LightGlue keeps (filter_threshold only, no min_conf cut)."""
    ring = _ring_of(extractor)
    kp0, des0 = ring.extract(img1)
    kp1, des1 = ring.extract(img2)
    if len(kp0) == 0 or len(kp1) == 0:
        return kp0, kp1, des0, des1, []
    h0, w0 = np.asarray(img1).shape[:2]
    h1, w1 = np.asarray(img2).shape[:2]
    ij, _scores, _stop = matcher.match(kp0._xy, des0, kp1._xy, des1, min_conf=0.0, size0=(w0, h0), size1=(w1, h1))
    return kp0, kp1, des0, des1, _convert_lg_matches_to_opencv(ij)


def _opencv_detect_and_match(img1, img2, detector, matcher):
    kp1, des1 = detector.detectAndCompute(img1, None)
    kp2, des2 = detector.detectAndCompute(img2, None)
    if des1 is None or des2 is None:
        return [], [], [], [], []        # gracefully handle empty images
    return kp1, kp2, des1, des2, sorted(matcher.match(des1, des2), key=lambda m: m.distance)


def detect_and_match(img1, img2, detector, matcher, args):
    """Front-end entry of the reference (features_utils.py:250-256): OpenCV or LightGlue depending on the CLI flag."""
    if args.use_lightglue:
        return _lightglue_detect_and_match(img1, img2, detector, matcher)
    return _opencv_detect_and_match(img1, img2, detector, matcher)


def filter_matches_ransac(kp1, kp2, matches, thresh=1.0):
    """Drop outliers with fundamental-matrix RANSAC, as the reference does at
    features_utils.py:185-200 - scored on the GPU (`sslam_fmat_ransac_host`: OpenCV's classic
    7-point RANSAC / LMedS restated, no cv2 needed)."""
    if len(matches) < 8:
        return matches
    ring = _last_ring() if _last_ring is not None else None
    f = ring.filtered if ring is not None else None
    if (f is not None and f["matches"] is matches and len(matches) == f["n"]
            and isinstance(kp1, KeyPointList) and isinstance(kp2, KeyPointList)
            and kp1.pristine_xy() is f["a"]["xy"] and kp2.pristine_xy() is f["b"]["xy"]):
        # the list feature_matcher just returned for exactly these frames
        if f["thr"] is not None and f["thr"] == float(thresh):
            # the filter already ran on the device behind the match (same matches, same pixels, same threshold):
            # its mask came back with the matches
            if f["kept"] == -1:                  # no model: cv2 returns mask None
                return []
            return [m for m, ok in zip(matches, f["mask"].tolist()) if ok]
        ring.ransac_thr = float(thresh)          # from now on the filter rides behind the match
    from ... import epipolar
    pts1 = np.float32([kp1[m.queryIdx].pt for m in matches])
    pts2 = np.float32([kp2[m.trainIdx].pt for m in matches])
    _, mask, _ = epipolar.find_fundamental_ransac(pts1, pts2, thresh, 0.99)
    if mask is None:
        return []
    return [m for m, ok in zip(matches, mask) if ok]
