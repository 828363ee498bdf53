"""The configuration bench.py TIMES, held to the oracle at its own size: `FrameStreamPipeline` built as
bench.py builds it for `value` - 1241 x 376 frames, 24 frames per round, ONE extractor stream taking 8 frames per call, three
matcher streams taking 8 pairs per launch, 2048 keypoints, cached hipGraphs - run for whole rounds and compared with

  * `oracle.aliked_ref` on the frames themselves: the single-frame entry is held STAGE-exact on the same frame (every dense
    stage within 1e-3, the detector's decisions bit-exact on the GPU's own score map: tests/test_aliked_gpu.py::_check) and
    the batched F = 8 entry must equal the single-frame entry bit for bit at this size, so the records the timed pipeline
    writes are the stage-exact ones; end to end they meet the oracle on the common keypoint set like every other ALIKED test;
  * `oracle.lightglue_ref` on the pipeline's own feature records: index arrays `assert_array_equal`, for pairs inside a
    batch, across an extractor chunk boundary and across a matcher batch boundary - once on the extracted features
    (random-init ALIKED descriptors are nearly all alike: a handful of matches at a low threshold) and once on a planted MATCHED
    chain (tests/lg_inputs.py::PlantedBatchExtractor - the form bench.py's `planted_matches` leg times): hundreds of matches
    per pair through emit / compaction / read-back of the batched path.

Reference call chain: slam/monocular/main_revamped.py:321-328 (feature_extractor(frame t), feature_matcher(t-1 -> t))."""
import numpy as np
import pytest

import frames
import lg_inputs
from conftest import load_pkg
from oracle import lightglue_ref

pytestmark = pytest.mark.gpu

H, WD, K, B, EF, P, NM = 376, 1241, 2048, 24, 8, 8, 3          # bench.py: H_IMG, W_IMG, MAX_KPTS, FRAMES_PER_RANK, EXT_FRAMES, BATCH_PAIRS, N_MAT


def _records(pipe, fs):
    pipe.sync()
    slab = np.empty((B, pipe.REC), np.float32)
    pipe.ctx.d2h(slab, pipe.rec_ptr(pipe.last_set * B))
    return [fs.unpack_record(slab[s], K) for s in range(B)]


def test_the_timed_pipeline_configuration_against_the_oracle(native):
    from test_aliked_gpu import _check
    W = load_pkg("weights"); fs = load_pkg("frame_shard")
    AL = load_pkg("aliked").AlikedHIP; LG = load_pkg("lightglue").LightGlueHIP
    sd_a = W.random_aliked_state_dict(0)
    sd_l = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)          # (bench's `value` weights never emit a match)
    dets = [AL(sd_a, max_num_keypoints=K, max_h=H, max_w=WD, ctx=native.Context(0), max_frames=EF)]
    mats = [LG(sd_l, max_kpts=K, ctx=native.Context(0), max_pairs=P) for _ in range(NM)]
    plan = fs.ShardPlan(1, 0, B)
    imgs = [frames.structured_frame(i) if i % 3 else frames.noise_frame(i) for i in range(B)]
    ctx = dets[0].ctx
    chunk = ctx.upload(np.stack(imgs))

    # ---- round(s) on the frames themselves; min_conf 0.05 so that the few matches of untrained descriptors are kept
    pipe = fs.FrameStreamPipeline(dets, mats, plan, K, 0.05, batch_pairs=P, use_graphs=True)
    assert pipe.EF == EF and pipe.P == P
    for _ in range(3):                                   # (the third round replays every graph the first two captured)
        pipe.round(chunk, H, WD, 3)
    recs = _records(pipe, fs)
    res, info = pipe.results(), pipe.infos()
    # (a) batched F = 8 == the single-frame entry, bit for bit, at 1241 x 376 - and the single-frame entry stage-exact on frame 0
    single = AL(sd_a, max_num_keypoints=K, max_h=H, max_w=WD, ctx=native.default_context(0))
    for s in (0, 3, 7, 8, 13, 23):                       # both kinds of frame, all three extractor chunks
        xy1, de1 = single.extract(imgs[s], K)
        n, xy, de = recs[s]
        assert n == len(xy1) and n > 0, (s, n, len(xy1))
        np.testing.assert_array_equal(xy, xy1, err_msg=f"frame {s}")
        np.testing.assert_array_equal(de, de1, err_msg=f"frame {s}")
    for s in (0, 13):
        xy1, de1, _ = _check(single, sd_a, imgs[s], K)   # every stage of THIS frame against oracle.aliked_ref
        np.testing.assert_array_equal(recs[s][1], xy1)
    single.close()
    # (b) index arrays of the pipeline == oracle.lightglue_ref on the pipeline's own records
    total = 0
    for s in (1, 8, 16, 23):                             # inside a batch, chunk boundary 7|8, batch boundary 15|16, the last pair
        (n0, xy0, d0), (n1, xy1, d1) = recs[s - 1], recs[s]
        rij, _, stop = lightglue_ref.reference_feature_matcher(sd_l, xy0, xy1, d0, d1, 0.05)
        np.testing.assert_array_equal(res[s][0].astype(np.int64), rij, err_msg=f"pair {s - 1} -> {s}")
        assert info[s, 0] == len(rij) and info[s, 2] == n0 and info[s, 3] == n1
        total += len(rij)
    assert total > 0, "vacuous: no match on the extracted frames at min_conf 0.05"

    # ---- the same instances, bench.py's threshold, records overwritten behind every batched extraction with a MATCHED chain
    chain = lg_inputs.make_chain(B, K, seed=7, noise=0.035, drop=0.1)
    pipe_p = fs.FrameStreamPipeline(dets, mats, plan, K, 0.7, batch_pairs=P, use_graphs=True)
    planter = lg_inputs.PlantedBatchExtractor(dets[0], chain, K)
    try:
        for _ in range(2):
            planter.i = 0
            pipe_p.round(chunk, H, WD, 3)
        recs_p = _records(pipe_p, fs)
        res_p, info_p = pipe_p.results(), pipe_p.infos()
    finally:
        planter.restore()
    for s in range(B):
        np.testing.assert_array_equal(recs_p[s][1], chain[s][0]); np.testing.assert_array_equal(recs_p[s][2], chain[s][1])
    assert info_p[1:, 0].min() >= 300, info_p[:, 0]      # every pair emits hundreds of matches (pair 0 is against the previous round's last frame)
    for s in (1, 8, 16):
        rij, _, stop = lightglue_ref.reference_feature_matcher(sd_l, chain[s - 1][0], chain[s][0], chain[s - 1][1], chain[s][1], 0.7)
        np.testing.assert_array_equal(res_p[s][0].astype(np.int64), rij, err_msg=f"planted pair {s - 1} -> {s}")
        assert len(rij) >= 300
    ctx.free(chunk)
    for x in dets + mats:
        x.close()


def test_batched_extractor_at_the_c5_size_equals_the_single_frame_entry(native):
    """BASELINE config C5 (1920 x 1080): the batched extractor entry at that size - what `bench.py`'s `c5` leg times, F = 8 per
    call - gives the single-frame entry's keypoints, descriptors and scores bit for bit (that entry is held to the oracle at this
    size by tests/test_aliked_gpu.py::test_c5_size_downscale_with_real_blur)."""
    W = load_pkg("weights"); AL = load_pkg("aliked").AlikedHIP
    Hh, Ww, Kk, F = 1080, 1920, 2048, 8
    sd_a = W.random_aliked_state_dict(3)
    ctx = native.Context(0)
    imgs = [frames.structured_frame(i, h=Hh, w=Ww) if i % 2 else frames.noise_frame(i, h=Hh, w=Ww) for i in range(F)]
    single = AL(sd_a, max_num_keypoints=Kk, max_h=Hh, max_w=Ww, ctx=ctx)
    want = [single.extract(im, Kk, return_scores=True) for im in imgs[:3] + imgs[-1:]]
    single.close()
    al = AL(sd_a, max_num_keypoints=Kk, max_h=Hh, max_w=Ww, ctx=ctx, max_frames=F)
    al.use_graphs(True)
    dev = [ctx.upload(im) for im in imgs]
    xy = [ctx.malloc(Kk * 8) for _ in imgs]; de = [ctx.malloc(Kk * 512) for _ in imgs]
    sc = [ctx.malloc(Kk * 4) for _ in imgs]; nn = [ctx.malloc(16) for _ in imgs]
    for _ in range(2):                                   # (the second call replays the cached graph pieces)
        al.extract_batch_dev(dev, Hh, Ww, 3, xy, de, sc, nn, Kk)
    ctx.sync()
    for slot, w in zip((0, 1, 2, F - 1), want):
        n = np.empty(4, np.int32); ctx.d2h(n, nn[slot])
        k = int(n[0])
        assert k == len(w[0]) > 0
        a = np.empty((Kk, 2), np.float32); d = np.empty((Kk, 128), np.float32); s = np.empty(Kk, np.float32)
        ctx.d2h(a, xy[slot]); ctx.d2h(d, de[slot]); ctx.d2h(s, sc[slot])
        np.testing.assert_array_equal(a[:k], w[0]); np.testing.assert_array_equal(d[:k], w[1]); np.testing.assert_array_equal(s[:k], w[2])
    for p in dev + xy + de + sc + nn:
        ctx.free(p)
    al.close()
