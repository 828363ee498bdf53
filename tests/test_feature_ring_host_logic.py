"""Host logic of the drop-in names' device-resident frame store (opencv-simpleslam_amd/feature_ring.py) on a CPU stand-in
for the native layer (tests/fake_device.py): which calls are answered from where - resident records, the memo, the
look-ahead's batched launch, a re-upload - on the call sequences the reference's loop really produces
(slam/monocular/main_revamped.py:325-343, slam/core/keyframe_utils.py:146-154, slam/core/triangulation_utils.py:131-132),
and that every path returns what the host path returns.  The same scenarios run on the real library in
tests/test_dropin_names_gpu.py."""
from types import SimpleNamespace

import numpy as np
import pytest

import fake_device as fd
import lg_inputs
from conftest import load_pkg

ARGS = SimpleNamespace(use_lightglue=True, min_conf=0.7, max_features=256)
IMG = np.zeros((24, 32, 3), np.uint8)
pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]


@pytest.fixture
def rig(monkeypatch):
    fu = load_pkg("slam.core.features_utils")
    fr = load_pkg("feature_ring")
    ep = load_pkg("epipolar")
    ctx = fd.FakeContext()
    det = fd.FakeAliked(ctx, lg_inputs.make_chain(40, 200, seed=5, period=8), 256)
    mat = fd.FakeLightGlue(ctx, 256, max_pairs=fr.DeviceFeatureRing.PAIRS)
    ring = det._feature_ring = fr.DeviceFeatureRing(det)
    ring.attach_matcher(mat)
    mat._feature_ring = ring
    fake_ep = fd.FakeEpipolar()
    monkeypatch.setattr(ep, "find_fundamental_ransac", fake_ep.find_fundamental_ransac)
    monkeypatch.setattr(ep, "filter_matches_dev", fake_ep.filter_matches_dev)
    return SimpleNamespace(fu=fu, det=det, mat=mat, ring=ring, ep=fake_ep, ctx=ctx)


def host_match(rig, k0, k1, d0, d1, args=ARGS):
    return pairs(rig.fu.feature_matcher(args, list(k0), list(k1), d0.copy(), d1.copy(), rig.mat))


def host_filter(rig, k0, k1, ms, thr):
    return pairs(rig.fu.filter_matches_ransac(list(k0), list(k1), list(ms), thr))


def slam_loop(rig, n_frames, cooldown, promote=lambda f: True, thr=2.5, bootstrap_ref=False):
    """The reference's call sequence; -> list of (k0, k1, d0, d1, raw pairs, kept pairs, what)."""
    fu, det, mat = rig.fu, rig.det, rig.mat
    calls = []

    def match_and_filter(k0, k1, d0, d1, what):
        raw = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
        kept = fu.filter_matches_ransac(k0, k1, raw, thr)
        calls.append((k0, k1, d0, d1, pairs(raw), pairs(kept), what))
        return raw

    kp_prev, des_prev = fu.feature_extractor(ARGS, IMG, det)
    kf, last_kf = (kp_prev, des_prev), 0
    for f in range(1, n_frames):
        kp, des = fu.feature_extractor(ARGS, IMG, det)
        match_and_filter(kp_prev, kp, des_prev, des, "prev->cur")
        if bootstrap_ref:
            match_and_filter(kf[0], kp, kf[1], des, "ref->cur")             # main_revamped.py:343, every frame
        elif f - last_kf > cooldown:
            r1 = match_and_filter(kf[0], kp, kf[1], des, "kf->cur")
            if promote(f):
                r2 = match_and_filter(kf[0], kp, kf[1], des, "kf->cur again")
                assert r2 is not r1 and pairs(r2) == pairs(r1)
                kf, last_kf = (kp, des), f
        kp_prev, des_prev = kp, des
    return calls


def check_against_host(rig, calls, thr=2.5):
    for k0, k1, d0, d1, raw, kept, what in calls:
        assert len(raw) >= 20, what
        ms = rig.fu.feature_matcher(ARGS, list(k0), list(k1), d0.copy(), d1.copy(), rig.mat)
        assert pairs(ms) == raw, what
        assert host_filter(rig, k0, k1, ms, thr) == kept, what


def test_frame_loop_look_ahead_and_filter_behind_the_match(rig):
    calls = slam_loop(rig, 12, cooldown=100)
    st = rig.ring.stats
    # frame 1 -> 2 runs at call time (and teaches the pattern), every later pair is the look-ahead's
    assert st["resident"] == 1 and st["ahead"] == 10 and st["memo"] == 0 and st["wasted"] == 0 and st["reupload"] == 0
    assert rig.mat.host_calls == 0 and rig.mat.dev_pairs == 11
    # the first filter call teaches the threshold on the host; from then on the filter rides behind the match
    assert rig.ep.host_calls == 1 and rig.ep.dev_calls == 10
    check_against_host(rig, calls)


def test_keyframe_pattern_cooldown_5(rig):
    calls = slam_loop(rig, 19, cooldown=5)                 # keyframes at 0, 6, 12, 18
    st = rig.ring.stats
    assert rig.mat.host_calls == 0
    assert st["reupload"] == 0                             # a keyframe is still resident six frames later
    assert st["memo"] == 3                                 # the triangulation's duplicate pairs
    assert st["ahead_kf"] == 2                             # keyframe matches 2 and 3 rode in the look-ahead's launch (learned gap)
    assert st["wasted"] == 0
    assert rig.mat.dev_pairs == 18 + 3                     # every distinct pair computed exactly once
    assert rig.mat.dev_calls == 18 + 1                     # ... and only the first keyframe match cost a launch sequence of its own
    check_against_host(rig, calls)


def test_keyframe_not_promoted_is_asked_again_and_predicted(rig):
    calls = slam_loop(rig, 14, cooldown=5, promote=lambda f: f >= 9)      # asked at 6, 7, 8, 9 (promoted), then 15 > range
    st = rig.ring.stats
    assert rig.mat.host_calls == 0 and st["reupload"] == 0
    assert st["ahead_kf"] == 3 and st["memo"] == 1 and st["wasted"] == 0   # 7, 8, 9 predicted: asked on the previous frame
    check_against_host(rig, calls)


def test_bootstrap_reference_is_matched_on_every_frame(rig):
    calls = slam_loop(rig, 10, cooldown=0, bootstrap_ref=True)
    st = rig.ring.stats
    # frame 1: ref == prev -> the second call is the memo's; frame 2: ref -> cur at call time; from frame 3 both pairs in one launch
    assert rig.mat.host_calls == 0 and st["memo"] == 1 and st["ahead_kf"] == 7 and st["wasted"] == 0
    assert rig.mat.dev_pairs == 9 + 8 and rig.mat.dev_calls == 9 + 1
    check_against_host(rig, calls)


def test_cooldown_longer_than_the_ring_uploads_the_keyframe_again_once(rig):
    calls = slam_loop(rig, 25, cooldown=10)                # keyframes at 0, 11, 22; every frame kept alive by `calls`
    st = rig.ring.stats
    assert rig.mat.host_calls == 0
    assert st["reupload"] == 1                             # keyframe 0 fell out (not yet known as a keyframe); 11 was pinned
    assert st["memo"] == 2 and st["ahead_kf"] == 1
    check_against_host(rig, calls)


def test_a_wrong_guess_costs_time_not_results(rig):
    # cooldown 3 for two keyframes, then the loop stops promoting / asking: one predicted keyframe pair is never asked for
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    calls = slam_loop(rig, 9, cooldown=3)                  # keyframes at 0, 4, 8
    assert ring.stats["wasted"] == 0 and ring.kf_gap == 4
    kp_prev, des_prev = calls[-1][1], calls[-1][3]
    more = []
    for f in range(9, 15):                                 # plain frame loop from here on
        kp, des = fu.feature_extractor(ARGS, IMG, det)
        raw = fu.feature_matcher(ARGS, kp_prev, kp, des_prev, des, mat)
        more.append((kp_prev, kp, des_prev, des, pairs(raw), pairs(fu.filter_matches_ransac(kp_prev, kp, raw, 2.5)), "prev->cur"))
        kp_prev, des_prev = kp, des
    assert ring.stats["wasted"] == 1 and ring.kf_gap is None           # frame 12's keyframe pair: computed, never asked, un-learned
    check_against_host(rig, calls + more)


def test_other_threshold_or_settings_do_not_hit_the_memo(rig):
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    k0, d0 = fu.feature_extractor(ARGS, IMG, det)
    k1, d1 = fu.feature_extractor(ARGS, IMG, det)
    m1 = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    m2 = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    assert ring.stats["memo"] == 1 and mat.dev_pairs == 1 and m2 is not m1 and pairs(m1) == pairs(m2)
    loose = SimpleNamespace(use_lightglue=True, min_conf=0.2)
    m3 = fu.feature_matcher(loose, k0, k1, d0, d1, mat)
    assert mat.dev_pairs == 2 and pairs(m3) == host_match(rig, k0, k1, d0, d1, loose)
    mat.epoch += 1                                         # what set_conf / set_precision do
    fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    assert mat.dev_pairs == 3 and ring.stats["memo"] == 1
    # edited keypoints: uploaded per call, never memoised, answered like the host path
    T = load_pkg("slam.core.types")
    k0e = T.KeyPointList(k0, k0._xy)
    k0e[3] = T.KeyPoint(k0[3].pt[0] + 90.0, k0[3].pt[1], 1)
    me = fu.feature_matcher(ARGS, k0e, k1, d0, d1, mat)
    assert mat.dev_pairs == 4 and pairs(me) == host_match(rig, k0e, k1, d0, d1)
    fu.feature_matcher(ARGS, k0e, k1, d0, d1, mat)
    assert mat.dev_pairs == 5 and mat.host_calls == 2      # (the two host_match calls above)


def test_edited_match_list_is_filtered_from_scratch(rig):
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    k0, d0 = fu.feature_extractor(ARGS, IMG, det)
    k1, d1 = fu.feature_extractor(ARGS, IMG, det)
    ring.ransac_thr = 2.5
    m = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    assert ring.results[-1]["thr"] == 2.5
    n_host = rig.ep.host_calls
    kept = fu.filter_matches_ransac(k0, k1, m, 2.5)
    assert rig.ep.host_calls == n_host and 0 < len(kept) <= len(m)                  # the device's mask
    ms = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    ms.sort(key=lambda x: -x.trainIdx)                                               # the OpenCV idiom: reorder in place
    assert pairs(fu.filter_matches_ransac(k0, k1, ms, 2.5)) == host_filter(rig, k0, k1, ms, 2.5)
    assert rig.ep.host_calls == n_host + 2
    me = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    me[0].trainIdx = me[1].trainIdx                                                  # an element edited in place: spot check
    assert pairs(fu.filter_matches_ransac(k0, k1, me, 2.5)) == host_filter(rig, k0, k1, me, 2.5)
    # ... and the edited OBJECT is never handed out again: the next answer is built from scratch
    fresh = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    assert pairs(fresh) == host_match(rig, k0, k1, d0, d1) and all(a is not b for a, b in zip(fresh, me))      # (`m` shares the edited object)
    # other keypoint lists (copies) for the same match list: host path
    n_host = rig.ep.host_calls
    mc = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    fu.filter_matches_ransac(list(k0), list(k1), mc, 2.5)
    assert rig.ep.host_calls == n_host + 1


def test_no_model_gives_nothing_on_the_fast_path_too(rig):
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    k0, d0 = fu.feature_extractor(ARGS, IMG, det)
    k1, d1 = fu.feature_extractor(ARGS, IMG, det)
    ring.ransac_thr = 1e-6                                 # nothing lies within 40e-6 px of the median displacement
    m = fu.feature_matcher(ARGS, k0, k1, d0, d1, mat)
    n_host = rig.ep.host_calls
    assert fu.filter_matches_ransac(k0, k1, m, 1e-6) == [] and rig.ep.host_calls == n_host     # info[3] == -1, not the raw mask
    assert host_filter(rig, k0, k1, m, 1e-6) == []


def test_dropped_frames_free_their_slots_and_recycled_ids_do_not_alias(rig):
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    kp_prev, des_prev = fu.feature_extractor(ARGS, IMG, det)
    for _ in range(40):                                    # only prev and cur are held, as the reference's loop does
        kp, des = fu.feature_extractor(ARGS, IMG, det)
        assert pairs(fu.feature_matcher(ARGS, kp_prev, kp, des_prev, des, mat)) == host_match(rig, kp_prev, kp, des_prev, des)
        kp_prev, des_prev = kp, des
    assert len(ring.records) <= 3 and ring.stats["reupload"] == 0
    assert sum(sl["rec"] is not None for sl in ring.slots) <= 3
    # a frame that is not the ring's (a copy of the descriptors) takes the host path
    n = mat.host_calls
    fu.feature_matcher(ARGS, kp_prev, kp, des_prev.copy(), des, mat)
    assert mat.host_calls == n + 1


def test_match_objects_prepared_from_the_last_count_are_topped_up_when_more_come_back(rig):
    """The DMatch shells built behind a running match are sized from the last result (tearing down the unused ones is on the
    frame's critical path), so a frame pair with many more matches than the last finds too few: the rest are built when the
    results arrive, against the same index source."""
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    kp_prev, des_prev = fu.feature_extractor(ARGS, IMG, det)
    seen = []
    for f in range(1, 8):
        kp, des = fu.feature_extractor(ARGS, IMG, det)
        if f == 5:
            ring.shell_hint = 3                       # as if the previous pairs had next to no matches
        ms = fu.feature_matcher(ARGS, kp_prev, kp, des_prev, des, mat)
        assert len(ms) >= 20
        assert pairs(ms) == host_match(rig, kp_prev, kp, des_prev, des)
        if not rig.fu.HAVE_CV2:                       # (the duck type resolves lazily: one source, consecutive rows)
            assert [m._i for m in ms if hasattr(m, "_i")] == list(range(len(ms)))
        assert len({id(m) for m in ms}) == len(ms)
        seen.append(ring.shell_hint)
        kp_prev, des_prev = kp, des
    assert ring.stats["ahead"] >= 5                    # (the look-ahead path is the one that prepares shells)
    assert all(h >= 256 for h in seen)                 # the hint follows the results again


def test_filter_verdicts_that_are_never_collected_do_not_leak_into_later_pairs(rig):
    """r06: the filter behind a match runs on a stream of its own and its verdict is fetched when `filter_matches_ransac` asks
    (or before anything reuses the mirror / the device outputs).  A caller that skips the filter for some frames, filters a
    LATER pair, or comes back to an earlier result after other matches have run must get each pair's own mask."""
    fu, det, mat, ring = rig.fu, rig.det, rig.mat, rig.ring
    thr = 2.5
    kp, des = [None] * 9, [None] * 9
    raw = [None] * 9
    kp[0], des[0] = fu.feature_extractor(ARGS, IMG, det)
    for f in range(1, 9):
        kp[f], des[f] = fu.feature_extractor(ARGS, IMG, det)
        raw[f] = fu.feature_matcher(ARGS, kp[f - 1], kp[f], des[f - 1], des[f], mat)
        if f in (1, 2, 5, 8):                              # frames 3, 4, 6, 7: the loop does not filter
            kept = fu.filter_matches_ransac(kp[f - 1], kp[f], raw[f], thr)
            assert pairs(kept) == host_filter(rig, kp[f - 1], kp[f], raw[f], thr), f
        assert len(ring.pending_masks) <= ring.PAIRS
    # an earlier result, asked after later matches overwrote the mirror: its verdict was taken into its entry in time
    n_host = rig.ep.host_calls
    for f in (7, 6):
        kept = fu.filter_matches_ransac(kp[f - 1], kp[f], raw[f], thr)
        assert pairs(kept) == host_filter(rig, kp[f - 1], kp[f], raw[f], thr), f
    assert rig.ep.host_calls == n_host + 2                 # (+2: the two host_filter reference calls; the ring answered from its entries)
