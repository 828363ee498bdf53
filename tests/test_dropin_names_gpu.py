"""The names slam/monocular/main_revamped.py:48-52 imports, driven the way the reference drives them
(reference slam/core/features_utils.py:18-30, :85-124, :136-171), on the GPU."""
from types import SimpleNamespace

import numpy as np
import pytest

import frames
from conftest import load_pkg
from oracle import aliked_ref, lightglue_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fu():
    return load_pkg("slam.core.features_utils")


SD_L = dict(seed=1, match_gain=4.0, match_bias=3.0)      # random-init weights that produce matches (not vacuous)


_SD_CACHE = {}


def _sd_l():
    if "sd" in _SD_CACHE:
        return _SD_CACHE["sd"]
    W = load_pkg("weights")
    _SD_CACHE["sd"] = W.random_lightglue_state_dict(SD_L["seed"], match_gain=SD_L["match_gain"], match_bias=SD_L["match_bias"])
    return _SD_CACHE["sd"]


@pytest.fixture(scope="module")
def pipeline(fu, gpu_ctx):
    """init_feature_pipeline(args) with the reference's default max_features (4000, features_utils.py:25).
    No checkpoint exists in the image, so the overlay falls back to seeded random weights; the seed is
    steered to a set that yields matches on the synthetic frames."""
    mp = pytest.MonkeyPatch()
    sd = _sd_l()                                                      # built BEFORE the function is patched
    mp.setattr(fu._weights, "random_lightglue_state_dict", lambda seed=0: sd)
    mp.setenv(fu.ENV_ALLOW_RANDOM, "1")
    mp.delenv(fu.ENV_ALIKED, raising=False); mp.delenv(fu.ENV_LIGHTGLUE, raising=False)
    args = SimpleNamespace(use_lightglue=True, min_conf=0.05)            # no max_features: the default applies
    det, mat = fu.init_feature_pipeline(args)
    mp.undo()
    yield args, det, mat
    det.close(); mat.close()


def test_init_feature_pipeline_defaults(pipeline):
    args, det, mat = pipeline
    assert det.max_num_keypoints == 4000 and mat.max_kpts == 4000 and mat.capacity == 4096
    assert list(mat.parameters()) == []                                  # the reference probes matcher.parameters()


def test_extractor_and_matcher_conventions(fu, pipeline):
    args, det, mat = pipeline
    W = load_pkg("weights")
    T = load_pkg("slam.core.types")
    img0, img1 = frames.structured_frame(0), frames.structured_frame(1)
    kp0, des0 = fu.feature_extractor(args, img0, det)
    kp1, des1 = fu.feature_extractor(args, img1, det)
    assert isinstance(kp0, list) and des0.dtype == np.float32 and des0.shape == (len(kp0), 128)
    assert all(hasattr(k, "pt") and isinstance(k.pt[0], float) for k in kp0[:5])
    np.testing.assert_allclose(np.linalg.norm(des0, axis=1), 1.0, atol=1e-5)      # unit rows (:100)
    ref = aliked_ref.aliked_extract(W.random_aliked_state_dict(0), img0, 4000)
    assert abs(len(kp0) - len(ref["keypoints"])) <= 0.005 * len(ref["keypoints"]) + 2
    # returned arrays are the caller's: a second extraction must not alias them
    keep = des0.copy()
    fu.feature_extractor(args, img1, det)
    np.testing.assert_array_equal(des0, keep)

    m = fu.feature_matcher(args, kp0, kp1, des0, des1, mat)
    assert isinstance(m, list) and len(m) > 0 and all(hasattr(x, "queryIdx") and hasattr(x, "trainIdx") for x in m[:5])
    q = [x.queryIdx for x in m]
    assert q == sorted(q) and len(set(q)) == len(q)                       # ascending queryIdx, one match per query
    xy0 = np.array([k.pt for k in kp0], np.float32); xy1 = np.array([k.pt for k in kp1], np.float32)
    rij, _, _ = lightglue_ref.reference_feature_matcher(_sd_l(), xy0, xy1, des0, des1, args.min_conf)
    np.testing.assert_array_equal(np.array([(x.queryIdx, x.trainIdx) for x in m], np.int64).reshape(-1, 2), rij)

    # None / empty inputs -> [] (features_utils.py:118-124); ([], []) is what the ORB branch returns (:105-106)
    assert fu.feature_matcher(args, None, kp1, des0, des1, mat) == []
    assert fu.feature_matcher(args, kp0, kp1, None, des1, mat) == []
    assert fu.feature_matcher(args, [], kp1, des0[:0], des1, mat) == []
    assert fu.feature_matcher(args, kp0, kp1, des0, np.zeros((0, 128), np.float32), mat) == []

    # torch-tensor descriptors are accepted like numpy ones (features_utils.py:136-154)
    torch = pytest.importorskip("torch")
    m_t = fu.feature_matcher(args, kp0, kp1, torch.tensor(des0), torch.tensor(des1), mat)      # (copies: the arrays are read-only)
    assert [(x.queryIdx, x.trainIdx) for x in m_t] == [(x.queryIdx, x.trainIdx) for x in m]

    # the keypoint list is the caller's: an in-place edit of an INTERIOR keypoint must be seen
    kp0_mut = list(kp0)
    i = len(kp0_mut) // 2
    kp0_mut[i] = T.KeyPoint(kp0[i].pt[0] + 200.0, kp0[i].pt[1], 1)
    xy0m = np.array([k.pt for k in kp0_mut], np.float32)
    rij_m, _, _ = lightglue_ref.reference_feature_matcher(_sd_l(), xy0m, xy1, des0, des1, args.min_conf)
    m_m = fu.feature_matcher(args, kp0_mut, kp1, des0, des1, mat)
    np.testing.assert_array_equal(np.array([(x.queryIdx, x.trainIdx) for x in m_m], np.int64).reshape(-1, 2), rij_m)

    # min_conf is read from args on every call, default 0.7 (features_utils.py:168)
    strict = fu.feature_matcher(SimpleNamespace(use_lightglue=True), kp0, kp1, des0, des1, mat)
    r7, _, _ = lightglue_ref.reference_feature_matcher(_sd_l(), xy0, xy1, des0, des1, 0.7)
    assert len(strict) == len(r7) <= len(m)


def test_grayscale_and_float_images(fu, pipeline):
    args, det, mat = pipeline
    g = frames.structured_frame(2, c=1)
    kp, des = fu.feature_extractor(args, g, det)
    assert len(kp) > 0 and des.shape[1] == 128
    with pytest.raises(TypeError):
        fu.feature_extractor(args, g.astype(np.float32), det)


def test_matcher_reads_the_extractor_device_records_when_it_can(fu, pipeline, monkeypatch):
    """r03: `feature_matcher` on the arrays `feature_extractor` returned reads both frames where the extractor
    left them on the GPU (no keypoint rebuild, no descriptor upload); a copied / edited keypoint list, descriptors
    that are not the returned array, or a frame that fell out of the ring take the host path - with the same
    matches either way (reference contract: features_utils.py:109-171, ownership SURVEY 8(b))."""
    args, det, mat = pipeline
    T = load_pkg("slam.core.types")
    frames_ = [frames.structured_frame(i) for i in range(7)]
    kp0, des0 = fu.feature_extractor(args, frames_[0], det)
    kp1, des1 = fu.feature_extractor(args, frames_[1], det)
    assert isinstance(kp0, T.KeyPointList) and not des0.flags.writeable
    with pytest.raises(ValueError):
        des0[0, 0] = 1.0                                  # read-only: the device copy cannot go stale silently
    host_calls = []
    real_match = mat.match
    monkeypatch.setattr(mat, "match", lambda *a, **k: (host_calls.append(1), real_match(*a, **k))[1])
    pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
    m_res = fu.feature_matcher(args, kp0, kp1, des0, des1, mat)
    assert host_calls == [] and len(m_res) > 0            # resident path
    m_host = fu.feature_matcher(args, list(kp0), list(kp1), des0.copy(), des1.copy(), mat)
    assert len(host_calls) == 1 and pairs(m_host) == pairs(m_res)      # same matches through the host path
    # copied lists with the SAME keypoints still hit (the keypoints are rebuilt and compared)
    fu.feature_matcher(args, list(kp0), list(kp1), des0, des1, mat)
    assert len(host_calls) == 1
    # an edited list: its keypoints are uploaded, the descriptors stay resident; result = host path on the same inputs
    kp0e = T.KeyPointList(kp0, kp0._xy)
    i = len(kp0e) // 3
    kp0e[i] = T.KeyPoint(kp0[i].pt[0] + 150.0, kp0[i].pt[1] + 3.0, 1)
    m_e = fu.feature_matcher(args, kp0e, kp1, des0, des1, mat)
    assert len(host_calls) == 1
    m_eh = fu.feature_matcher(args, list(kp0e), kp1, des0.copy(), des1, mat)
    assert len(host_calls) == 2 and pairs(m_e) == pairs(m_eh)
    # an element edited in place is caught by the spot check or by the rebuild: never a stale result
    orig = kp1[0].pt
    kp1[0].pt = (orig[0] + 80.0, orig[1])                # (the list itself is untouched and still "pristine")
    m_p = fu.feature_matcher(args, kp0, kp1, des0, des1, mat)
    m_ph = fu.feature_matcher(args, list(kp0), list(kp1), des0.copy(), des1.copy(), mat)
    assert pairs(m_p) == pairs(m_ph)
    kp1[0].pt = orig
    # a frame held longer than the ring (8 slots, least recently used first) is uploaded again on its next use - the
    # descriptor array is read-only, so the host copy is still the truth - and answered on the device like any other
    held = [fu.feature_extractor(args, im, det) for im in (frames_[2:7] + frames_[2:7])]      # all kept alive: frame 0 falls out
    ring = fu._ring_of(det)
    assert ring.records[id(des0)].slot is None
    n_before, up_before = len(host_calls), ring.stats["reupload"]
    m_old = fu.feature_matcher(args, kp0, held[-1][0], des0, held[-1][1], mat)
    assert len(host_calls) == n_before and ring.stats["reupload"] == up_before + 1 and ring.records[id(des0)].slot is not None
    m_oldh = fu.feature_matcher(args, list(kp0), list(held[-1][0]), des0.copy(), held[-1][1].copy(), mat)
    assert pairs(m_old) == pairs(m_oldh)
    # a dropped descriptor array frees its frame at once (it can never be asked for again)
    key = id(held[3][1])
    assert key in ring.records
    held[3] = None
    assert key not in ring.records


def test_legacy_pair_entry_normalises_by_the_image_size(fu, pipeline):
    """`detect_and_match` / `_lightglue_detect_and_match` (reference features_utils.py:233-256), the form the
    reference's own tests/test_lightglue_vs_manual.py drives, on that test's disc pair: same keypoints and
    descriptors as the split API (its assertions a, b), and the matches of the oracle run WITH 'image_size'
    (upstream normalize_keypoints(kpts, size): the feature dicts of `extractor.extract` carry it) and without
    the min_conf cut."""
    args, det, mat = pipeline
    img1, img2 = frames.disc_pair()
    kp1, kp2, d1, d2, m = fu.detect_and_match(img1, img2, det, mat, args)
    kp1s, d1s = fu.feature_extractor(args, img1, det)
    kp2s, d2s = fu.feature_extractor(args, img2, det)
    assert [k.pt for k in kp1] == [k.pt for k in kp1s] and [k.pt for k in kp2] == [k.pt for k in kp2s]
    np.testing.assert_array_equal(d1, d1s); np.testing.assert_array_equal(d2, d2s)
    if len(kp1) == 0 or len(kp2) == 0:
        pytest.skip("no keypoints on the disc pair with these weights")
    xy1 = np.array([k.pt for k in kp1], np.float32); xy2 = np.array([k.pt for k in kp2], np.float32)
    h1, w1 = img1.shape[:2]; h2, w2 = img2.shape[:2]
    ref = lightglue_ref.lightglue_forward(_sd_l(), xy1, d1, xy2, d2, size0=(w1, h1), size1=(w2, h2))
    np.testing.assert_array_equal(np.array([(x.queryIdx, x.trainIdx) for x in m], np.int64).reshape(-1, 2), ref["matches"].numpy())
    # the bounding-box normalisation of the split API is a different function of the same keypoints
    nob = lightglue_ref.lightglue_forward(_sd_l(), xy1, d1, xy2, d2)
    kn_b = lightglue_ref.normalize_keypoints(__import__("torch").as_tensor(xy1)[None])
    kn_s = lightglue_ref.normalize_keypoints(__import__("torch").as_tensor(xy1)[None], (w1, h1))
    assert not np.allclose(kn_b.numpy(), kn_s.numpy())
    assert ref["matches"].shape[1] == 2 and nob["matches"].shape[1] == 2


def test_look_ahead_match_of_the_frame_loop_gives_the_same_matches(fu, pipeline):
    """r03: in the reference's frame loop every `feature_extractor(cur)` is followed by `feature_matcher(prev, cur)`
    (main_revamped.py:325-330).  Once the ring has seen that pattern, `feature_extractor` itself enqueues that match
    behind the extraction and the matcher call only collects it.  Same matches as the plain path (a fresh pipeline's
    first pair, the host path); a call with other arguments (keyframe -> cur, another threshold) is answered
    correctly while a look-ahead is outstanding; extractions that never collect it switch it off."""
    args, det, mat = pipeline
    ring = fu._ring_of(det)
    pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
    host = lambda k0, k1, d0, d1, a=args: pairs(fu.feature_matcher(a, list(k0), list(k1), d0.copy(), d1.copy(), mat))
    fr = [frames.structured_frame(20 + i) for i in range(8)]
    kp = [None] * 8; des = [None] * 8
    kp[0], des[0] = fu.feature_extractor(args, fr[0], det)
    kp[1], des[1] = fu.feature_extractor(args, fr[1], det)
    m01 = pairs(fu.feature_matcher(args, kp[0], kp[1], des[0], des[1], mat))       # plain resident path; sets the pattern
    assert ring.ahead_on and ring.ahead is None and len(m01) > 0
    got = []
    for i in (2, 3, 4):                                                           # look-ahead enqueued by the extractor
        kp[i], des[i] = fu.feature_extractor(args, fr[i], det)
        assert ring.ahead is not None
        got.append(pairs(fu.feature_matcher(args, kp[i - 1], kp[i], des[i - 1], des[i], mat)))
        assert ring.ahead is None and ring.ahead_on
    for i, g in zip((2, 3, 4), got):
        assert g == host(kp[i - 1], kp[i], des[i - 1], des[i]) and len(g) > 0
    # other arguments while a look-ahead (4 -> 5) is outstanding: keyframe 3 -> 5, then 4 -> 5 at another threshold
    kp[5], des[5] = fu.feature_extractor(args, fr[5], det)
    assert ring.ahead is not None
    m35 = pairs(fu.feature_matcher(args, kp[3], kp[5], des[3], des[5], mat))
    assert ring.ahead is None and ring.kf is ring.records[id(des[3])]              # frame 3 is in the keyframe role now
    assert m35 == host(kp[3], kp[5], des[3], des[5])
    loose = SimpleNamespace(use_lightglue=True, min_conf=0.2)
    m45 = pairs(fu.feature_matcher(loose, kp[4], kp[5], des[4], des[5], mat))
    assert ring.ahead_on and m45 == host(kp[4], kp[5], des[4], des[5], loose)
    kp[6], des[6] = fu.feature_extractor(args, fr[6], det)                         # look-ahead at 0.2 ...
    assert ring.ahead is not None and ring.ahead["thr"] == 0.2
    m56 = pairs(fu.feature_matcher(args, kp[5], kp[6], des[5], des[6], mat))       # ... but the call asks for 0.7
    assert m56 == host(kp[5], kp[6], des[5], des[6])
    # extractions that never collect their look-ahead switch it off
    kp[7], des[7] = fu.feature_extractor(args, fr[7], det)
    assert ring.ahead is not None
    fu.feature_extractor(args, fr[0], det)
    assert ring.ahead is None and not ring.ahead_on


def test_filter_matches_ransac_behind_the_match_equals_the_host_filter(fu, pipeline):
    """r04: the reference's loop filters every match with F-matrix RANSAC right away (main_revamped.py:118-126).  Once
    `filter_matches_ransac` has been called on a resident match, the filter is enqueued on the device behind every such
    match (sslam_fmat_ransac_dev on the matcher's own output) and the call only applies the mask that came back with
    the matches: same kept matches - the same OBJECTS - as the host filter on the same lists; another threshold or
    other lists take the host path.  Random-init networks do not produce enough matches on extracted features, so the
    device records of the extracted frames are overwritten with the synthetic matched features of the parity tests
    (white box: slot contents, entry["xy"], a KeyPointList built from the planted keypoints)."""
    import lg_inputs
    args, det, mat = pipeline
    ring = fu._ring_of(det)
    ring.ransac_thr = None
    pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
    host = lambda k0, k1, ms, thr: pairs(fu.filter_matches_ransac(list(k0), list(k1), list(ms), thr))   # plain lists: host path
    strict = SimpleNamespace(use_lightglue=True, min_conf=0.7)
    img = frames.structured_frame(40)
    fast = []
    for i in range(1, 6):
        k0, d0, k1, d1 = lg_inputs.make_pair(900, 860, seed=50 + i)
        planter = lg_inputs.PlantedExtractor(det, [(k0, d0), (k1, d1)])
        try:
            kp0, des0 = fu.feature_extractor(args, img, det)
            kp1, des1 = fu.feature_extractor(args, img, det)
        finally:
            planter.restore()
        np.testing.assert_array_equal(des0, d0); np.testing.assert_array_equal(kp1._xy, k1)      # the planted frames came back
        m = fu.feature_matcher(strict, kp0, kp1, des0, des1, mat)
        assert len(m) >= 100                                           # the RANSAC branch with real work
        thr = 2.5 if i == 3 else 1.0
        on_device = ring.results[-1]["matches"] is m and ring.results[-1]["thr"] == thr
        fast.append(on_device)
        f = fu.filter_matches_ransac(kp0, kp1, m, thr)
        assert pairs(f) == host(kp0, kp1, m, thr), f"pair {i}"
        ids = {id(x) for x in m}
        assert all(id(x) in ids for x in f) and 0 < len(f) <= len(m)
        if on_device:
            # an edited match list must not get the mask of the list that was handed out: same answer as the
            # host filter on the edited list
            m2 = fu.feature_matcher(strict, kp0, kp1, des0, des1, mat)            # (the memo: a new list, same pairs)
            assert m2 is not m and pairs(m2) == pairs(m)
            m2.reverse()
            assert pairs(fu.filter_matches_ransac(kp0, kp1, m2, thr)) == host(kp0, kp1, m2, thr)
    # call 1 teaches the threshold (host), 2 rides behind the match, 3 asks another threshold (host, teaches 2.5), 4 asks 1.0
    # again while 2.5 was enqueued (host), 5 rides again
    assert fast == [False, True, False, False, True], fast
    ring.ransac_thr = None


def test_few_matches_with_no_model_give_nothing_on_both_paths(fu, pipeline):
    """8 - 14 matches take OpenCV's LMedS branch, and a best model with fewer than 7 inliers means `mask is None`
    -> the reference returns [] (features_utils.py:196-198).  The device-side filter signals that as info[3] == -1; the
    fast path must honour it like the host path does."""
    import lg_inputs
    args, det, mat = pipeline
    ring = fu._ring_of(det)
    pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
    strict = SimpleNamespace(use_lightglue=True, min_conf=0.7)
    img = frames.structured_frame(41)
    ring.ransac_thr = 0.01                               # (as if the loop had already asked for this threshold)
    seen = 0
    for seed in range(12):
        # ~10 true correspondences whose second view is scattered by tens of pixels: no epipolar model fits 7 of them at 0.01 px
        k0, d0, k1, d1 = lg_inputs.make_pair(300, 300, seed=900 + seed, drop=0.965)
        rng = np.random.default_rng(seed)
        k1 = (k1 + rng.uniform(-40, 40, k1.shape)).astype(np.float32)
        planter = lg_inputs.PlantedExtractor(det, [(k0, d0), (k1, d1)])
        try:
            kp0, des0 = fu.feature_extractor(args, img, det)
            kp1, des1 = fu.feature_extractor(args, img, det)
        finally:
            planter.restore()
        m = fu.feature_matcher(strict, kp0, kp1, des0, des1, mat)
        if not 8 <= len(m) <= 14:
            continue
        seen += 1
        assert ring.results[-1]["matches"] is m and ring.results[-1]["thr"] == 0.01
        got = pairs(fu.filter_matches_ransac(kp0, kp1, m, 0.01))
        want = pairs(fu.filter_matches_ransac(list(kp0), list(kp1), list(m), 0.01))
        assert got == want
    ring.ransac_thr = None
    assert seen >= 1, "no pair with 8..14 matches among the seeds - retune the drop rate"


def test_keyframe_pattern_of_the_frame_loop_stays_on_the_device(fu, pipeline, monkeypatch):
    """the reference's loop is not one match per frame.  Beyond the cooldown `select_keyframe` matches
    the KEYFRAME against the current frame (keyframe_utils.py:153-154), a promoted frame makes
    `triangulate_between_kfs_2view` match THE SAME pair again (triangulation_utils.py:131-132), and every match is
    filtered (main_revamped.py:118-126).  Replayed here with cooldown 5 over 13 frames (keyframes at 0, 6, 12): every call
    is answered from the device-resident records - the keyframe is still resident six frames later, the duplicate pair
    comes from the memo, the second keyframe match rides in the look-ahead's launch - with the matches and the kept
    matches of the host path."""
    import lg_inputs
    args, det, mat = pipeline
    ring = fu._ring_of(det)
    ring.forget_patterns()
    strict = SimpleNamespace(use_lightglue=True, min_conf=0.7)
    pairs = lambda ms: [(m.queryIdx, m.trainIdx) for m in ms]
    host_calls = []
    real_match = mat.match
    monkeypatch.setattr(mat, "match", lambda *a, **k: (host_calls.append(1), real_match(*a, **k))[1])
    chain = lg_inputs.make_chain(13, 1500, seed=3)
    img = frames.structured_frame(42)
    stats0 = dict(ring.stats)
    calls = []                                             # (kp0, kp1, des0, des1, raw pairs, kept pairs, what)

    def match_and_filter(k0, k1, d0, d1, what):
        raw = fu.feature_matcher(strict, k0, k1, d0, d1, mat)
        kept = fu.filter_matches_ransac(k0, k1, raw, 2.5)
        calls.append((k0, k1, d0, d1, pairs(raw), pairs(kept), what))
        return raw

    planter = lg_inputs.PlantedExtractor(det, chain)
    try:
        kp_prev, des_prev = fu.feature_extractor(args, img, det)
        kf, last_kf = (kp_prev, des_prev), 0
        for f in range(1, 13):
            kp, des = fu.feature_extractor(args, img, det)
            match_and_filter(kp_prev, kp, des_prev, des, "prev->cur")
            if f - last_kf > 5:                            # (frame_no - last_kf_frame_no) > kf_cooldown
                r1 = match_and_filter(kf[0], kp, kf[1], des, "kf->cur")
                r2 = match_and_filter(kf[0], kp, kf[1], des, "kf->cur again")      # promoted: the triangulation's match
                assert r2 is not r1 and pairs(r2) == pairs(r1)
                kf, last_kf = (kp, des), f
            kp_prev, des_prev = kp, des
    finally:
        planter.restore()
    assert host_calls == []                                # nothing took the host path
    d = {k: ring.stats[k] - stats0[k] for k in ring.stats}
    assert d["reupload"] == 0                              # the keyframes were still resident six frames later
    assert d["memo"] == 2                                  # both duplicate pairs
    assert d["ahead_kf"] == 1                              # the second keyframe match was predicted (the learned gap)
    assert d["ahead"] >= 10 and d["wasted"] == 0
    assert sum(1 for c in calls if c[6] != "prev->cur") == 4
    for k0, k1, d0, d1, raw, kept, what in calls:
        ms = fu.feature_matcher(strict, list(k0), list(k1), d0.copy(), d1.copy(), mat)          # host path
        assert pairs(ms) == raw and len(raw) >= 100, what
        assert pairs(fu.filter_matches_ransac(list(k0), list(k1), ms, 2.5)) == kept, what
    assert len(host_calls) == len(calls)
    ring.ransac_thr = None


def test_ring_survives_recycled_array_ids(fu, pipeline):
    """A long run of the frame loop, holding only the previous and the current frame as the reference's loop does
    (main_revamped.py:708): the allocator hands the ids of dropped descriptor arrays to new ones, and the ring, which
    keys its records by id, must neither alias a dead frame nor lose a live one - every prev -> cur match stays on the
    device-resident path."""
    args, det, mat = pipeline
    calls = []
    real = mat.match
    mat.match = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        kp_prev, des_prev = fu.feature_extractor(args, frames.structured_frame(40), det)
        for i in range(1, 25):
            kp, des = fu.feature_extractor(args, frames.structured_frame(40 + i % 6), det)
            fu.feature_matcher(args, kp_prev, kp, des_prev, des, mat)
            kp_prev, des_prev = kp, des
    finally:
        del mat.match
    assert calls == []
