"""Device residency of the frames `feature_extractor` served, and the matches scheduled on them.

This is the state behind the drop-in names of slam/core/features_utils.py: the reference's frame loop hands every
frame's features back to `feature_matcher` one or more times (slam/monocular/main_revamped.py:325-330 prev -> cur on
every frame, :343 ref -> cur before the bootstrap; slam/core/keyframe_utils.py:153-154 keyframe -> cur once the
cooldown has passed; slam/core/triangulation_utils.py:131-132 THE SAME keyframe pair again when the frame was promoted),
always as the very objects `feature_extractor` returned.  The ring keeps those frames where the extractor left them on
the GPU so that a match reads both operands in place, and it knows three things about the calls that follow:

  residency by use   a frame stays resident while it is being matched: victims are chosen dead-first (the caller
                     dropped the descriptor array), then least-recently-USED; the frame in the keyframe role (the
                     query side of a non-consecutive match) and a frame just promoted to it are never victims.  A frame
                     that did fall out (a cooldown longer than the ring) is uploaded again on its next use - its
                     descriptor array is read-only, so the host copy is still the truth - and stays.
  memo               the result of a match is a pure function of (frame a, frame b, threshold, matcher settings); the
                     last few are kept, so the duplicate keyframe pair of the triangulation costs a list build.
  look-ahead         `feature_extractor(cur)` enqueues the matches the loop is about to ask for behind the extraction,
                     on the matcher's stream: prev -> cur once that pattern has been seen, and keyframe -> cur IN THE
                     SAME batched launch when the loop is due to ask for it (asked on the previous frame and not yet
                     promoted, or the learned cooldown gap has passed) - two pairs fill the chip one pair leaves half
                     empty.  The F-matrix filter of `filter_matches_ransac` rides behind each pair once its threshold
                     is known.  A wrong guess costs time, never a result: `feature_matcher` answers from the memo only
                     for exactly (a, b, threshold).

Ownership: the arrays handed to the caller are the caller's - the descriptor array is returned READ-ONLY
(the reference never writes into it; an in-place edit would silently desynchronise the device copy, so numpy refuses
it), the keypoint list is a `KeyPointList` and the match list a `MatchList`, which know when they were edited.
"""
from __future__ import annotations

import weakref
from collections import deque
from itertools import compress

import numpy as np

from . import _native, epipolar
from .slam.core.types import (KeyPointList, MatchList, bind_matches, dmatch_edit_epoch, keypoint_shells, keypoints_from_xy,
                              match_shells, matches_from_ij, xy_from_keypoints)


class FrameRecord:
    """One extracted frame: where its features live on the device (`slot`, None while evicted), what the caller holds
    (`desc_ref`: weak reference to the returned descriptor array, `xy`: the array its KeyPointList was built from)."""
    __slots__ = ("slot", "n", "desc_ref", "xy", "seq", "used", "__weakref__")

    def __init__(self, slot, n, desc, xy, seq, on_drop):
        self.slot, self.n, self.xy, self.seq, self.used = slot, n, xy, seq, seq
        self.desc_ref = weakref.ref(desc, on_drop)


class DeviceFeatureRing:
    SLOTS = 8
    PAIRS = 2                # pairs of one look-ahead launch: prev -> cur, keyframe -> cur
    MEMO = 6

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
                                   score=base + self.o_score, rec=None))
        self.pin_rec = self.ctx.host_alloc(self.o_score)
        self.pin_cnt = self.pin_rec[:16].view(np.int32)
        self.pin_xy = self.pin_rec[self.o_xy:self.o_desc].view(np.float32).reshape(K, 2)
        self.pin_desc = self.pin_rec[self.o_desc:self.o_score].view(np.float32).reshape(K, 128)
        self.records = {}                # id(descriptor array) -> FrameRecord (resident or evicted), while the array lives
        self.seq = 0
        self.img_dev, self.img_cap = 0, 0
        self.tmp_xy = [m(K * 8), m(K * 8)]               # keypoints of an edited list (uploaded per call)
        # match results of up to PAIRS pairs, [info P x 16 B | pairs P x K x 2 | RANSAC info P x 16 B | RANSAC mask P x KM]
        # + scores P x K: everything but the scores comes back in ONE copy (the filter of filter_matches_ransac runs on the
        # device right behind each match, see enqueue_filters_and_readback)
        P = self.PAIRS
        KM = self.KM = (K + 15) // 16 * 16
        self.o_ij, self.o_rsi, self.o_rsm = 16 * P, 16 * P + 8 * K * P, 16 * P + 8 * K * P + 16 * P
        self.match_bytes = self.o_rsm + KM * P
        out = m(self.match_bytes + 4 * K * P)
        self.out_info, self.out_ij = out, out + self.o_ij
        self.rs_info, self.rs_mask = out + self.o_rsi, out + self.o_rsm
        self.out_sc = out + self.match_bytes
        self.pin_match = self.ctx.host_alloc(self.match_bytes)
        self.pin_info = self.pin_match[:16 * P].view(np.int32).reshape(P, 4)
        self.pin_ij = self.pin_match[self.o_ij:self.o_rsi].view(np.int32).reshape(P, K, 2)
        self.pin_rs_info = self.pin_match[self.o_rsi:self.o_rsm].view(np.int32).reshape(P, 4)
        self.pin_rs_mask = self.pin_match[self.o_rsm:].reshape(P, KM)
        self.ransac_thr = None           # threshold of the last filter_matches_ransac call on a resident match (None: not seen)
        self.results = deque(maxlen=self.MEMO)     # match lists handed out: dict(matches, k, ij, a, b, thr, entry = the memo entry with none / mask)
        self.memo = deque(maxlen=self.MEMO)        # dict(a, b, thr, epoch, k, ij, filter_thr, none, mask, asked)
        import os
        # the slots are a fixed set of buffers, so the launch sequence COULD replay as a cached hipGraph (SSLAM_RING_GRAPHS=1) - the
        # frame loop is a dependent chain on the device, though, and there a graph's replay stops behind its 15th node or pays a
        # hand-over per piece (0.405 against 0.381 ms per single-frame call by HIP events, profiles/r06_graph_segments.md) while
        # the host enqueues the 32 plain launches as the first ones already run.  (At the level of the whole loop the two forms
        # are inside the run-to-run noise of +-3 %.)  The batched pipeline, which has host time to save and long kernels, keeps
        # its graphs.
        detector.use_graphs(os.environ.get("SSLAM_RING_GRAPHS", "0") == "1")
        self.matcher, self.mctx = None, None
        # the F-matrix filter behind a match runs on a stream of its own (r06): `feature_matcher` returns when {count, pairs}
        # are back, the filter's ~70 us of small kernels run while the host builds the match list and the caller gets to its
        # `filter_matches_ransac` call, which then only waits for the mask
        self.fctx, self.ev_matched = None, None
        self.filter_inflight = False     # filter launches (and the mask's read-back) are queued on fctx behind ev_matched
        self.pending_masks = []          # [(memo entry, pair index)]: entries whose mask is still on its way
        self.ev_extracted = self.ctx.event()
        self.last = None                 # the most recently extracted frame
        self.ahead_on = False            # the prev -> cur pattern has been seen
        self.ahead = None                # outstanding look-ahead: dict(pairs=[(a, b), ...], thr, filter_thr)
        self.last_thr = None
        self.shell_hint = 1 << 30        # DMatch shells to prepare behind a running match: from the last result (first time: all)
        # the keyframe role (the query side of non-consecutive matches against the newest frame)
        self.kf = None                   # the frame in that role
        self.kf_next = None              # a frame the loop has just promoted (its keyframe pair was asked twice)
        self.kf_asked_seq = -1           # seq of the newest frame a keyframe match was asked for
        self.kf_gap = None               # frames between a keyframe and the first match against it (the loop's cooldown + 1)
        self.stats = dict(resident=0, memo=0, ahead=0, ahead_kf=0, reupload=0, wasted=0)

    def attach_matcher(self, matcher):
        self._resolve_masks()
        if self.fctx is None or matcher.ctx is not self.mctx:
            self.fctx = type(matcher.ctx)(matcher.ctx.device)
            self.ev_matched = matcher.ctx.event()
        self.matcher, self.mctx = matcher, matcher.ctx

    def forget_patterns(self):
        """Drop everything learned about the caller (look-ahead, keyframe role, filter threshold, memo); the resident
        frames stay."""
        self._finish_ahead()
        self._resolve_masks()
        self.memo.clear(); self.results.clear()
        self.ahead_on, self.last_thr, self.ransac_thr = False, None, None
        self.kf = self.kf_next = self.kf_gap = None
        self.kf_asked_seq = -1

    # ------------------------------------------------------------------ residency
    def _dropped(self, key, ref):
        """The caller dropped a descriptor array: its frame can never be asked for again."""
        rec = self.records.get(key)
        if rec is not None and rec.desc_ref is ref:
            del self.records[key]
            if rec is self.kf:
                self.kf = None
            if rec is self.kf_next:
                self.kf_next = None
            if rec.slot is not None and rec.slot["rec"] is rec and not self._busy(rec):
                rec.slot["rec"] = None
                rec.slot = None

    def _busy(self, rec):
        return self.ahead is not None and any(rec is a or rec is b for a, b in self.ahead["pairs"])

    def _take_slot(self, keep=None):
        """A free slot, else the one holding a dead frame, else the least recently used one; never the newest frame, the
        keyframe, a frame just promoted, an operand of the outstanding look-ahead, or `keep` (the other operand of the call
        being answered)."""
        victim = None
        for sl in self.slots:
            rec = sl["rec"]
            if rec is None:
                return sl
            if rec is self.last or rec is self.kf or rec is self.kf_next or rec is keep or self._busy(rec):
                continue
            key = (rec.desc_ref() is not None, rec.used)
            if victim is None or key < victim[0]:
                victim = (key, sl)
        sl = victim[1]                               # (SLOTS >= 7: at most newest + keyframe + promoted + `keep` + 2 look-ahead operands are held)
        rec = sl["rec"]                              # (None by now if a collection inside this loop ran the victim's drop callback)
        if rec is not None:
            rec.slot = None                          # evicted: the record stays known and is uploaded again on its next use
            sl["rec"] = None
        return sl

    def _make_resident(self, rec, des, keep=None):
        sl = self._take_slot(keep)
        self.ctx.h2d(sl["xy"], rec.xy)
        self.ctx.h2d(sl["desc"], des)
        self.ctx.h2d(sl["cnt"], np.array([rec.n, 0, 0, 0], np.int32))
        sl["rec"], rec.slot = rec, sl
        self.stats["reupload"] += 1

    def lookup(self, des, kps, which, keep=None):
        """(device xy, device desc, n, device count, record or None, record) for a frame `feature_extractor` returned, else
        None.  [4] is None when the keypoints are not the ones remembered (an edited list: no memo, no look-ahead); [5] is the
        record whose slot holds the descriptors either way (`keep` of the second operand's lookup)."""
        rec = self.records.get(id(des))
        if rec is None or rec.desc_ref() is not des or len(kps) != rec.n:
            return None
        if rec.slot is None:
            self._make_resident(rec, des, keep)      # a frame held longer than the ring: up again, the array is read-only
        rec.used = self.seq
        sl = rec.slot
        xy = kps.pristine_xy() if isinstance(kps, KeyPointList) else None
        if xy is not None and xy is rec.xy:
            return sl["xy"], sl["desc"], rec.n, sl["cnt"], rec, rec
        # another list / an edited one: rebuild the keypoints like the reference does (features_utils.py:65-77);
        # the descriptors on the device are still the ones of `des`
        xy = xy_from_keypoints(kps)
        if np.array_equal(xy, rec.xy):
            return sl["xy"], sl["desc"], rec.n, sl["cnt"], rec, rec
        self.ctx.h2d(self.tmp_xy[which], xy)         # (its only readers are matches `match` has already waited for)
        return self.tmp_xy[which], sl["desc"], rec.n, sl["cnt"], None, rec

    # ------------------------------------------------------------------ extraction
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
        if self.ahead is not None:       # the last look-ahead was never collected: the caller is not in the prev -> cur loop
            self._finish_ahead()
            self.ahead_on = False
        self._resolve_masks()            # (a filter still reading two frames' keypoints: their slots may be recycled below)
        self._retire_unasked()
        sl = self._take_slot()
        self.seq += 1
        K = self.K
        # (the image goes up straight from the caller's pageable array: the runtime's own staged copy, 69 us for
        #  1.4 MB, beats a host copy into a page-locked stage + DMA, 57 + 41 us)
        staged = np.ascontiguousarray(img)   # (a non-contiguous image: this copy must outlive the DMA - it is held until the ctx.sync() below)
        ctx.h2d_async(self.img_dev, staged)  # (pageable source: the runtime stages it before the call returns; page-locked: the DMA reads it in place)
        prev = self.last
        det.extract_dev(self.img_dev, H, Wd, Cn, sl["xy"], sl["desc"], sl["score"], sl["cnt"], max_kpts=K)
        ctx.record(self.ev_extracted)
        ctx.d2h_async(self.pin_rec, sl["base"])
        look = (self.ahead_on and self.matcher is not None and prev is not None and prev.slot is not None
                and prev.n > 0 and self.last_thr is not None)
        ahead = None
        if look:
            # (this frame's count is only known on the device yet: K bounds it, the matcher clamps to the record's count)
            self.mctx.wait(self.ev_extracted)
            operands = [prev]
            kf = self._keyframe_due(prev)
            if kf is not None:
                operands.append(kf)
            self._enqueue([(a.slot, a.n, sl, K) for a in operands], self.last_thr)
            # published at once (operands only, the new frame has no record yet): from here on `_busy` keeps their slots out
            # of the victim search, also for a drop callback that a collection runs before this call returns
            ahead = self.ahead = dict(pairs=[(a, a) for a in operands], thr=self.last_thr, filter_thr=self.ransac_thr)
        # the GPU needs ~0.5 ms from here: build the frame's KeyPoint objects meanwhile (their coordinates resolve
        # against the array below on first use)
        shells, src = keypoint_shells(K) if keypoint_shells is not None else (None, None)
        ctx.sync()
        del staged                           # the upload is done: the caller's image may change from here on
        n = int(self.pin_cnt[0])
        if n < 0:                            # (al_finalize_kernel: the frame's range flag; a look-ahead on it matched an empty frame)
            det.range_overflow()             # reported here: clear the instance's sticky word
            if ahead is not None:            # the look-ahead is still reading slot `sl` (and writing the match mirror): let it finish,
                self.mctx.sync()             # discard it - the slot has no record and is free for the next call
                self.ahead = None
            self.seq -= 1                    # no record was made for this number
            raise _native.NativeError("feature_extractor: an activation left the fp16 range of the split-precision stages "
                                      "(|value| >= 65520): the frame's features are void")
        xy = self.pin_xy[:n].copy(); desc = self.pin_desc[:n].copy()
        desc.setflags(write=False)
        if shells is not None:
            src.xy = xy
            if n < K:
                del shells[n:]
            kps = KeyPointList(shells, xy)
        else:
            kps = KeyPointList(keypoints_from_xy(xy), xy)
        key = id(desc)
        rec = FrameRecord(sl, n, desc, xy, self.seq, lambda ref, key=key, ring=weakref.ref(self): (
            ring() is not None and ring()._dropped(key, ref)))
        sl["rec"] = rec
        self.records[key] = rec
        self.last = rec
        if ahead is not None:
            ahead["pairs"] = [(a, rec) for a, _ in ahead["pairs"]]
        return kps, desc

    def _keyframe_due(self, prev):
        """The keyframe record if the loop is due to ask keyframe -> cur for the frame being extracted: it asked on the
        previous frame and has not promoted a frame since (beyond the cooldown it asks on every frame until one is
        promoted, keyframe_utils.py:146-154; before the bootstrap it asks ref -> cur on every frame, main_revamped.py:343),
        or the gap learned from the last keyframe has passed."""
        if self.matcher is None or self.matcher.max_pairs < 2:
            return None
        cand = self.kf_next if self.kf_next is not None else self.kf
        if cand is None or cand is prev or cand.slot is None or cand.n == 0 or cand.desc_ref() is None:
            return None
        seq = self.seq                                   # (already the new frame's)
        if self.kf_next is None and self.kf_asked_seq == prev.seq:
            return cand
        if self.kf_gap is not None and seq - cand.seq == self.kf_gap:
            return cand
        return None

    # ------------------------------------------------------------------ matches
    def _enqueue(self, pairs, thr):
        """pairs: [(slot a, bound a, slot b, bound b)] -> one launch sequence on the matcher's stream, the filter behind every
        pair, one read-back."""
        mt = self.matcher
        self._resolve_masks()            # the mirror and the device outputs are about to be overwritten
        if len(pairs) == 1:
            sa, na, sb, nb = pairs[0]
            mt.match_dev(sa["xy"], sa["desc"], na, sb["xy"], sb["desc"], nb, self.out_ij, self.out_sc, self.out_info,
                         min_conf=thr, m_dev=sa["cnt"], n_dev=sb["cnt"])
        else:
            mt.match_batch_dev([(sa["xy"], sa["desc"], na, sb["xy"], sb["desc"], nb, sa["cnt"], sb["cnt"])
                                for sa, na, sb, nb in pairs], self.out_ij, self.out_sc, self.out_info, self.K, min_conf=thr)
        self.enqueue_filters_and_readback([(sa["xy"], sb["xy"]) for sa, _, sb, _ in pairs])

    def enqueue_filters_and_readback(self, xys):
        """Behind a match on the matcher's stream: the reference's frame loop filters every match with F-matrix RANSAC
        right away (main_revamped.py:118-126) - once that has been seen, the filter runs on the device on the matcher's
        own output (sslam_fmat_ransac_dev: no host round trip, no pixel gather on the host) and its mask rides back with
        {count, pairs} in the same copy."""
        K = self.K
        self.mctx.d2h_async(self.pin_match[:self.o_rsi], self.out_info)           # {count, pairs}: what feature_matcher waits for
        if self.ransac_thr is not None:
            self.mctx.record(self.ev_matched)
            self.fctx.wait(self.ev_matched)
            for p, (xy_a, xy_b) in enumerate(xys):
                epipolar.filter_matches_dev(self.fctx, K, self.out_info + 16 * p, xy_a, xy_b, self.out_ij + 8 * K * p, None,
                                            self.rs_info + 16 * p, thresh=self.ransac_thr, confidence=0.99,
                                            mask_out_dev=self.rs_mask + self.KM * p)
            self.fctx.d2h_async(self.pin_match[self.o_rsi:], self.rs_info)        # {RANSAC verdict, mask}: what filter_matches_ransac waits for
            # (nothing is queued on the matcher's stream behind the filter - its synchronisation must not wait for it; whoever
            #  enqueues there next, or recycles a slot, resolves the masks first: _enqueue, extract)
            self.filter_inflight = True

    def _harvest(self, pairs, thr, filter_thr, asked):
        """After the matcher's stream has been synchronised: the results of `pairs` out of the page-locked mirror as memo entries
        (`asked` False: a look-ahead's results, nobody has asked for them yet) -> the entries; the caller decides which go into the memo."""
        epoch = self.matcher.epoch
        out = []
        for p, (a, b) in enumerate(pairs):
            k = int(self.pin_info[p, 0])
            e = dict(a=a, b=b, thr=thr, epoch=epoch, k=k, ij=self.pin_ij[p, :max(k, 0)].copy(), filter_thr=None,
                     none=False, mask=None, asked=asked)
            if filter_thr is not None and k >= 0:
                e["filter_thr"] = filter_thr
                self.pending_masks.append((e, p))                  # (the filter may still be running: _resolve_masks)
            out.append(e)
        return out

    def _resolve_masks(self):
        """The filter's verdicts of the last launch out of the page-locked mirror into their memo entries (waits for the filter's
        stream if they are still on their way)."""
        if self.filter_inflight:
            self.fctx.sync()
            self.filter_inflight = False
        for e, p in self.pending_masks:
            e["none"] = int(self.pin_rs_info[p, 3]) == -1          # no model (cv2 returns mask None): nothing is kept
            e["mask"] = self.pin_rs_mask[p, :e["k"]].copy()
        self.pending_masks = []

    def _finish_ahead(self):
        ahead, self.ahead = self.ahead, None
        if ahead is not None:
            self.mctx.sync()
            self.memo.extend(self._harvest(ahead["pairs"], ahead["thr"], ahead["filter_thr"], asked=False))

    def _retire_unasked(self):
        """Look-ahead results nobody asked for by the time the next frame arrives were wrong guesses: a keyframe pair
        un-learns the gap (the loop's cadence changed), and they leave the memo."""
        for e in [e for e in self.memo if not e["asked"]]:
            self.memo.remove(e)
            self.stats["wasted"] += 1
            if e["a"].seq < e["b"].seq - 1:
                self.kf_gap = None
                self.kf_asked_seq = -1

    def _memo_find(self, ra, rb, thr):
        epoch = self.matcher.epoch
        for e in self.memo:
            if e["a"] is ra and e["b"] is rb and e["thr"] == thr and e["epoch"] == epoch:
                return e
        return None

    def match(self, a, b, thr):
        """Both frames are on the GPU: answer (a, b, thr) from the look-ahead / the memo, or enqueue the match on their records
        now; -> MatchList."""
        mt, mctx = self.matcher, self.mctx
        ra, rb = a[4], b[4]
        known = ra is not None and rb is not None
        shells = src = None
        e = None
        if known:
            ra.used = rb.used = self.seq
            ahead = self.ahead
            hit = ahead is not None and ahead["thr"] == thr and any(x is ra and y is rb for x, y in ahead["pairs"])
            if hit:
                # the GPU may still be matching: build the DMatch objects meanwhile (indices resolve against the array below) -
                # as many as the last matches suggest, not one per keypoint: the unused ones are torn down AFTER the results
                # have arrived, on the frame's critical path (1 400 of 2 048 at ~600 matches: ~30 us)
                shells, src = match_shells(min(ra.n, rb.n, self.shell_hint))
            self._finish_ahead()
            e = self._memo_find(ra, rb, thr)
            if e is not None and not e["asked"]:         # a look-ahead's result, asked for the first time
                self.stats["ahead"] += 1
                if ra.seq < rb.seq - 1:
                    self.stats["ahead_kf"] += 1
            elif e is not None:
                self.stats["memo"] += 1
        else:
            self._finish_ahead()
        if e is None:
            self._enqueue([(dict(xy=a[0], desc=a[1], cnt=a[3]), a[2], dict(xy=b[0], desc=b[1], cnt=b[3]), b[2])], thr)
            shells, src = match_shells(min(a[2], b[2], self.shell_hint))
            mctx.sync()
            self.stats["resident"] += 1
            e = self._harvest([(ra, rb)], thr, self.ransac_thr, asked=True)[0]
            if known:                                  # (edited keypoint lists: this call's result only - it must not push a
                self.memo.append(e)                    #  pair the loop will ask for again out of the memo)
        first_ask = not e["asked"]
        e["asked"] = True
        k = e["k"]
        if k < 0:
            if e in self.memo:
                self.memo.remove(e)
            mt.range_overflow()                # reported here: clear the instance's sticky word
            raise _native.NativeError("feature_matcher: an activation left the fp16 range of the split-precision path "
                                      "(|value| >= 65520); rescale the descriptors or use matcher.set_precision('f32')")
        ij = e["ij"]
        epoch = dmatch_edit_epoch()
        if e.get("objs") is not None and epoch is not None and e["objs_epoch"] == epoch:
            # asked again and no DMatch anywhere has been edited since: a new list of the SAME objects (a list copy instead of
            # hundreds of constructions - the duplicate keyframe pair of the triangulation)
            out = MatchList(e["objs"], ij)
        elif shells is not None:
            out = MatchList(bind_matches(shells, src, ij), ij)     # (fewer or more matches than the hint prepared for: trimmed / made)
        else:
            out = MatchList(matches_from_ij(ij), ij)
        if known and epoch is not None:
            e["objs"], e["objs_epoch"] = list(out), epoch
        self.shell_hint = max(256, k + k // 4 + 64)
        if known:
            self._learn(ra, rb, thr, first_ask)
            self.results.append(dict(matches=out, k=k, ij=ij, a=ra, b=rb, thr=e["filter_thr"], entry=e))
        return out

    def _learn(self, ra, rb, thr, first_ask):
        """What the call says about the loop: prev -> cur switches the look-ahead on; a non-consecutive query against the
        newest frame names the keyframe; the same keyframe pair asked twice means the newest frame was promoted."""
        self.last_thr = thr
        newest = rb is self.last
        if newest and ra.seq == rb.seq - 1:
            self.ahead_on = True
        elif newest and ra.seq < rb.seq - 1:
            if not first_ask and ra is self.kf:
                self.kf_next = rb                    # triangulate_between_kfs_2view on (prev_kf, the new keyframe)
                return
            if ra is not self.kf:
                if ra is self.kf_next or self.kf is None or ra.seq > self.kf.seq:
                    self.kf, self.kf_next = ra, None
                    self.kf_gap = rb.seq - ra.seq    # the first match against a new keyframe: the loop's cooldown + 1
            if ra is self.kf:
                self.kf_asked_seq = rb.seq
        elif not newest:
            self.ahead_on = False                    # not the frame loop

    # ------------------------------------------------------------------ filter
    def filtered(self, kp1, kp2, matches, thresh):
        """`filter_matches_ransac(kp1, kp2, matches, thresh)` on a list this ring handed out for exactly these frames: the
        kept matches if the filter already ran on the device behind the match at this threshold, else None (and the
        threshold is remembered: from now on the filter rides behind the match)."""
        r = None
        for x in self.results:
            if x["matches"] is matches:
                r = x
                break
        if (r is None or len(matches) != r["k"] or not isinstance(matches, MatchList) or matches.pristine_ij() is not r["ij"]
                or not isinstance(kp1, KeyPointList) or not isinstance(kp2, KeyPointList)
                or kp1.pristine_xy() is not r["a"].xy or kp2.pristine_xy() is not r["b"].xy):
            return None
        if r["thr"] is not None and r["thr"] == float(thresh):
            self._resolve_masks()                 # (the filter ran beside the host's work on the match list: its mask is due now)
            e = r["entry"]
            if e["none"]:                         # no model: cv2 returns mask None, the reference returns []
                return []
            return list(compress(matches, e["mask"].tolist()))
        self.ransac_thr = float(thresh)
        return None
