"""GPU: the batched LightGlue entry point (`sslam_lightglue_match_batch_dev`).

A batch is n independent pairs in one enqueue (every launch covers all pairs).  Bar: every pair's
result equals the single-pair call's - bit for bit when both run the same key split of the
attention launches, and index-identical to the torch-CPU oracle in every configuration (ragged
sizes, an empty image, early stop and point pruning decided per pair, a partly filled batch)."""
import numpy as np
import pytest

import lg_inputs
from conftest import load_pkg
from oracle import lightglue_ref as R

pytestmark = pytest.mark.gpu


class DevBatch:
    """Device-resident inputs / outputs of one batch, on raw context allocations (no torch)."""

    def __init__(self, ctx, pairs, stride):
        self.ctx, self.stride, self.n = ctx, stride, len(pairs)
        self.ptrs, self.args = [], []
        for k0, d0, k1, d1 in pairs:
            a = [ctx.upload(np.ascontiguousarray(v, np.float32)) for v in (k0, d0, k1, d1)]
            self.ptrs += a
            self.args.append((a[0], a[1], len(k0), a[2], a[3], len(k1)))
        self.ij = ctx.malloc(self.n * stride * 8)
        self.sc = ctx.malloc(self.n * stride * 4)
        self.info = ctx.malloc(self.n * 16)
        self.ptrs += [self.ij, self.sc, self.info]

    def run(self, lg, min_conf):
        lg.match_batch_dev(self.args, self.ij, self.sc, self.info, self.stride, min_conf=min_conf)
        self.ctx.sync()
        ij = np.empty((self.n, self.stride, 2), np.int32)
        sc = np.empty((self.n, self.stride), np.float32)
        info = np.empty((self.n, 4), np.int32)
        self.ctx.d2h(ij, self.ij); self.ctx.d2h(sc, self.sc); self.ctx.d2h(info, self.info)
        return [(ij[p, :info[p, 0]].copy(), sc[p, :info[p, 0]].copy(), info[p].copy()) for p in range(self.n)]

    def free(self):
        for p in self.ptrs:
            self.ctx.free(p)


def _oracle(sd, pair, min_conf, conf=None):
    ref = R.lightglue_forward(sd, *pair, conf)
    keep = ref["scores"] > min_conf
    return ref["matches"][keep].numpy(), ref["scores"][keep].numpy(), ref["stop"]


def test_batch_equals_single_pair_calls_and_oracle(gpu_ctx):
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    sizes = [(512, 512), (300, 417), (64, 33), (640, 1), (129, 128)]
    pairs = [lg_inputs.make_pair(m, n, seed=m + n) for m, n in sizes]
    batch = LG(sd, max_kpts=640, max_pairs=6, ctx=gpu_ctx)
    single = LG(sd, max_kpts=640, ctx=gpu_ctx)
    single.debug_big_gemm(1)               # the batch's form of the linears (by size a 640-keypoint pair would take the
    batch.debug_big_gemm(1)                # ring kernels, whose LayerNorm sums in another order): same arithmetic both sides
    dev = DevBatch(gpu_ctx, pairs, 640)
    got = dev.run(batch, 0.7)
    for ks in (0, 1):                      # the single-pair default (key split + merge), then the batch's own
        single.debug_key_split(ks)
        batch.debug_key_split(ks)
        got = dev.run(batch, 0.7)
        for pr, (ij, sc, info) in zip(pairs, got):
            s_ij, s_sc, s_stop = single.match(*pr, min_conf=0.7)
            np.testing.assert_array_equal(ij, s_ij)
            assert info[1] == s_stop
            np.testing.assert_array_equal(sc, s_sc)          # same arithmetic -> bit-identical scores
            o_ij, o_sc, o_stop = _oracle(sd, pr, 0.7)
            np.testing.assert_array_equal(ij, o_ij)
            np.testing.assert_allclose(sc, o_sc, atol=1e-3, rtol=1e-3)
            assert info[1] == o_stop
    # the two attention kernels without key split - the r02 4-wave kernel (-1) and the hand-scheduled assembly kernel
    # batched launches run by default (-3) - multiply the same products in the same order: bit-identical results,
    # ragged key counts, the 1-keypoint image and all
    for mode in (-1, -3):
        batch.debug_key_split(mode)
        for (ij, sc, info), (p_ij, p_sc, p_info) in zip(got, dev.run(batch, 0.7)):
            np.testing.assert_array_equal(ij, p_ij)
            np.testing.assert_array_equal(sc, p_sc)
            np.testing.assert_array_equal(info, p_info)
    # default key-split policy of a batch differs from the single-pair one: indices still identical
    batch.debug_key_split(0); single.debug_key_split(0)
    assert sum(len(g[0]) for g in got) > 300
    dev.free(); batch.close(); single.close()


def test_batched_linear_forms_agree_with_the_ring_kernels_and_the_oracle(gpu_ctx):
    """The batched form of the linears (128 x 128 projections + the whole FFN as one kernel, ffn_fused.hpp) against
    the 64-row ring kernels of the single-pair path on the same batch: the projections accumulate every output in
    the same k order (bit-identical), the fused FFN sums the LayerNorm statistics in another order (registers of a
    lane, then lanes, then waves), so indices and control flow must be identical and scores equal to fp32
    rounding - including ragged row counts, pruning and an early stop; both must give the oracle's indices."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    for seed, kw in ((1, dict(match_gain=4.0, match_bias=3.0)), (4, dict(match_gain=4.0, match_bias=-4.6, conf_bias=2.3))):
        sd = W.random_lightglue_state_dict(seed, **kw)
        pairs = [lg_inputs.make_pair(m, n, seed=m + n) for m, n in [(512, 512), (300, 417), (640, 77), (129, 128)]]
        batch = LG(sd, max_kpts=640, max_pairs=4, ctx=gpu_ctx)
        dev = DevBatch(gpu_ctx, pairs, 640)
        batch.debug_big_gemm(0)
        ring = dev.run(batch, 0.0)
        batch.debug_big_gemm(1)
        big = dev.run(batch, 0.0)
        for (a_ij, a_sc, a_info), (b_ij, b_sc, b_info), pr in zip(ring, big, pairs):
            np.testing.assert_array_equal(a_ij, b_ij)
            np.testing.assert_array_equal(a_info, b_info)
            np.testing.assert_allclose(a_sc, b_sc, rtol=0, atol=2e-4)      # (vs the oracle the bar is 1e-3)
            o_ij, o_sc, o_stop = _oracle(sd, pr, 0.0)
            np.testing.assert_array_equal(b_ij, o_ij)
            assert b_info[1] == o_stop
        if seed == 1:
            assert sum(len(r[0]) for r in big) > 100
        dev.free(); batch.close()


def test_batch_with_an_empty_image_early_stop_and_pruning_per_pair(gpu_ctx):
    """Control flow is per pair: one pair stops after layer 1, one prunes, one has an empty image."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=-4.6, conf_bias=2.3)
    pairs = [lg_inputs.make_pair(400, 350, seed=6), lg_inputs.make_pair(256, seed=7),
             lg_inputs.make_pair(128, 200, seed=8)]
    k0, d0, k1, d1 = lg_inputs.make_pair(64, seed=9)
    pairs.append((k0[:0], d0[:0], k1, d1))                     # empty query image -> no matches, no fault
    batch = LG(sd, max_kpts=512, max_pairs=4, ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 512)
    got = dev.run(batch, 0.0)
    stops = set()
    for pr, (ij, sc, info) in zip(pairs[:3], got[:3]):
        o_ij, o_sc, o_stop = _oracle(sd, pr, 0.0)
        np.testing.assert_array_equal(ij, o_ij)
        np.testing.assert_allclose(sc, o_sc, atol=1e-3, rtol=1e-3)
        assert info[1] == o_stop
        stops.add(int(info[1]))
    assert got[0][2][2] < 400 or got[0][2][3] < 350            # pruning really happened in pair 0
    assert len(got[3][0]) == 0 and got[3][2][0] == 0
    # a second, smaller batch on the same instance (stale state of pairs 2, 3 must not leak)
    dev2 = DevBatch(gpu_ctx, pairs[1:3], 512)
    got2 = dev2.run(batch, 0.0)
    for a, b in zip(got2, got[1:3]):
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    dev.free(); dev2.free(); batch.close()


def test_batch_at_c2_size_no_key_split(gpu_ctx):
    """The bench configuration itself: 8 pairs of 2048 x 2048 keypoints (BASELINE C2 size) per enqueue, replayed as a
    cached hipGraph - the attention launches run un-split on the assembly kernel (one workgroup sees every key of its
    queries, context written straight to the split planes), the FFN in 64-token tiles.  Plain enqueue and graph
    replay must agree bit for bit; four of the eight pairs (one of them ragged) are held to the oracle."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0)
    pairs = [lg_inputs.make_pair(2048, seed=11 + i) for i in range(7)] + [lg_inputs.make_pair(2048, 1900, seed=20)]
    batch = LG(sd, max_kpts=2048, max_pairs=8, ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 2048)
    plain = dev.run(batch, 0.7)
    batch.use_graphs(True)
    dev.run(batch, 0.7)                                   # capture
    got = dev.run(batch, 0.7)                             # replay
    for (ij, sc, info), (p_ij, p_sc, p_info) in zip(got, plain):
        np.testing.assert_array_equal(ij, p_ij)
        np.testing.assert_array_equal(sc, p_sc)
        np.testing.assert_array_equal(info, p_info)
    for k in (0, 3, 6, 7):                                # the oracle at 2048 is slow: 4 of 8
        ij, sc, info = got[k]
        o_ij, o_sc, o_stop = _oracle(sd, pairs[k], 0.7)
        np.testing.assert_array_equal(ij, o_ij)
        np.testing.assert_allclose(sc, o_sc, atol=1e-3, rtol=1e-3)
        assert info[1] == o_stop == 9 and len(ij) > 100
    assert not batch.range_overflow()
    dev.free(); batch.close()


def test_batch_argument_errors(gpu_ctx, native):
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    lg = LG(W.random_lightglue_state_dict(0), max_kpts=128, max_pairs=2, ctx=gpu_ctx)
    pairs = [lg_inputs.make_pair(16, seed=1)] * 3
    dev = DevBatch(gpu_ctx, pairs, 128)
    with pytest.raises(ValueError, match="capacity"):
        dev.run(lg, 0.5)
    dev.args = dev.args[:2]; dev.n = 2
    dev.stride = 8                                           # fewer rows than a pair can emit
    with pytest.raises(native.NativeError, match="out_stride"):
        lg.match_batch_dev(dev.args, dev.ij, dev.sc, dev.info, 8)
    with pytest.raises(native.NativeError, match="max_pairs"):
        LG(W.random_lightglue_state_dict(0), max_kpts=128, max_pairs=17, ctx=gpu_ctx)
    dev.free(); lg.close()


@pytest.mark.parametrize("max_kpts,sizes", [(128, [(128, 97), (64, 128), (5, 31)]), (256, [(256, 130), (129, 256)]),
                                            (384, [(384, 384), (300, 77), (1, 384), (260, 259)])])
def test_assembly_attention_at_small_capacities(gpu_ctx, max_kpts, sizes):
    """The hand-scheduled attention kernel (csrc/gen_lg_attention_asm.py) at the capacities the other tests do not
    reach: one, two and three query blocks per (image, head) (the single-block case skips the magic-number division
    of the XCD remap), one-tile and ragged key counts, 1 / 5 keypoints (workgroups beyond n exit, whole waves without
    a valid query), forced on at any batch size (debug_key_split(-3)).  Bit-identical to the r02 4-wave kernel
    (-1), index-identical to the oracle."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(3, match_gain=4.0, match_bias=3.0)
    pairs = [lg_inputs.make_pair(m, n, seed=7 * m + n) for m, n in sizes]
    batch = LG(sd, max_kpts=max_kpts, max_pairs=len(pairs), ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, max_kpts)
    batch.debug_key_split(-1)
    ref = dev.run(batch, 0.5)
    batch.debug_key_split(-3)
    got = dev.run(batch, 0.5)
    for pr, (ij, sc, info), (r_ij, r_sc, r_info) in zip(pairs, got, ref):
        np.testing.assert_array_equal(ij, r_ij)
        np.testing.assert_array_equal(sc, r_sc)
        np.testing.assert_array_equal(info, r_info)
        o_ij, o_sc, o_stop = _oracle(sd, pr, 0.5)
        np.testing.assert_array_equal(ij, o_ij)
        np.testing.assert_allclose(sc, o_sc, atol=1e-3, rtol=1e-3)
    assert sum(len(g[0]) for g in got) > 20
    # the fused FFN in 64-token and in 32-token tiles (the form small token sets take): a token's arithmetic is the
    # same in both - bit-identical results
    for mode in (2, 3):
        batch.debug_big_gemm(mode)
        for (ij, sc, info), (t_ij, t_sc, t_info) in zip(got if mode == 3 else dev.run(batch, 0.5), dev.run(batch, 0.5)):
            np.testing.assert_array_equal(ij, t_ij)
            np.testing.assert_array_equal(sc, t_sc)
            np.testing.assert_array_equal(info, t_info)
        if mode == 2:
            got = dev.run(batch, 0.5)
    assert not batch.range_overflow()
    dev.free(); batch.close()


@pytest.mark.parametrize("seed,kw", [(3, dict(match_gain=4.0, match_bias=3.0)),
                                     (4, dict(match_gain=4.0, match_bias=-4.6, conf_bias=2.3))])
def test_assembly_attention_key_ranges_give_the_4_wave_kernel_partials(gpu_ctx, seed, kw):
    """Key ranges on the assembly kernel (what a single pair runs: 2 or 4 ranges of the keys + the merge launch): the
    same (o, m, l) partials as lg_attention_p_kernel with the same split, so everything downstream is bit-identical -
    full and ragged key counts, a range that holds the image's ragged last tile, fewer key tiles than ranges (empty
    ranges: o = 0, m = -inf, l = 0), a 1-keypoint image, and (second weight set) pruning and an early stop that shrink
    the key counts from layer to layer.  The default policy must also give the oracle's indices."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(seed, **kw)
    sizes = [(1024, 1024), (700, 900), (130, 64), (1000, 1), (257, 511)]
    pairs = [lg_inputs.make_pair(m, n, seed=5 * m + n) for m, n in sizes]
    batch = LG(sd, max_kpts=1024, max_pairs=len(pairs), ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 1024)
    for ks in (2, 4):
        batch.debug_key_split(ks)
        ref = dev.run(batch, 0.5)
        batch.debug_key_split(100 + ks)
        got = dev.run(batch, 0.5)
        for (ij, sc, info), (r_ij, r_sc, r_info) in zip(got, ref):
            np.testing.assert_array_equal(ij, r_ij)
            np.testing.assert_array_equal(sc, r_sc)
            np.testing.assert_array_equal(info, r_info)
    if seed == 3:
        assert sum(len(g[0]) for g in got) > 20
    else:
        assert any(info[1] < 9 for _, _, info in got)          # (this weight set stops early and prunes)
    # one pair per call: the default policy (key ranges on the assembly kernel) against the same policy on the 4-wave
    # kernel, and against the oracle
    single = LG(sd, max_kpts=1024, ctx=gpu_ctx)
    for pr in pairs:
        single.debug_key_split(-4)
        r_ij, r_sc, r_stop = single.match(*pr, min_conf=0.5)
        single.debug_key_split(0)
        ij, sc, stop = single.match(*pr, min_conf=0.5)
        np.testing.assert_array_equal(ij, r_ij)
        np.testing.assert_array_equal(sc, r_sc)
        assert stop == r_stop
        o_ij, o_sc, o_stop = _oracle(sd, pr, 0.5)
        np.testing.assert_array_equal(ij, o_ij)
        # (an image pruned down to no keypoints at all - the 1-keypoint image under this weight set - ends the pair on
        # the device at that layer with no matches; upstream keeps stepping the other image until its confidence test
        # fires and reports that later layer with the same empty result: the layer number is compared elsewhere)
        if len(pr[2]) > 1:
            assert stop == o_stop
    assert not batch.range_overflow() and not single.range_overflow()
    dev.free(); batch.close(); single.close()


@pytest.mark.parametrize("seed,kw", [(1, dict(match_gain=4.0, match_bias=3.0)),
                                     (4, dict(match_gain=4.0, match_bias=-4.6, conf_bias=2.3)),
                                     (7, dict(match_gain=4.0, match_bias=-1.0, conf_bias=1.2))])
def test_token_heads_in_the_fused_ffn_decide_like_the_separate_kernel(gpu_ctx, seed, kw):
    """The token-confidence / matchability heads evaluated in the cross block's fused FFN (default for batched token
    sets) against the lane-per-token kernel (debug_big_gemm(5)): the dot products sum in another order, so the control
    flow they feed - stop layer, the keypoints kept by every pruning step - and the matches must be identical and the
    scores equal to fp32 rounding; weight sets that never stop, stop at once, and stop / prune midway; ragged sizes."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(seed, **kw)
    sizes = [(1024, 1024), (700, 900), (130, 64), (1000, 1), (257, 511)]
    pairs = [lg_inputs.make_pair(m, n, seed=11 * m + n) for m, n in sizes]
    batch = LG(sd, max_kpts=1024, max_pairs=len(pairs), ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 1024)
    batch.debug_big_gemm(5)
    ref = dev.run(batch, 0.3)
    batch.debug_big_gemm(-1)
    got = dev.run(batch, 0.3)
    for (ij, sc, info), (r_ij, r_sc, r_info) in zip(got, ref):
        np.testing.assert_array_equal(info, r_info)
        np.testing.assert_array_equal(ij, r_ij)
        np.testing.assert_allclose(sc, r_sc, rtol=2e-5, atol=2e-6)
    single = LG(sd, max_kpts=1024, ctx=gpu_ctx)                  # one pair per call: 32-token FFN tiles
    for pr, (ij, sc, info) in zip(pairs, got):
        s_ij, s_sc, s_stop = single.match(*pr, min_conf=0.3)
        np.testing.assert_array_equal(ij, s_ij)
        assert info[1] == s_stop
    assert not batch.range_overflow()
    dev.free(); batch.close(); single.close()


def test_precision_f16x3p1_assembly_kernel_equals_the_4_wave_kernel_and_keeps_the_matches(gpu_ctx):
    """Precision "f16x3p1" (r04 opt-in, the default since r05): P as ONE fp16 plane in P.V (row sums over the rounded weights).
    Its hand-scheduled kernel (gen_lg_attention_asm_p1.py) gives what the 4-wave kernel's `p_single` branch gives, bit for bit -
    un-split (batch) and in key ranges (single-pair policy) - and the matches of "f16x3" and of the oracle."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    sizes = [(512, 512), (300, 417), (64, 33), (640, 1), (129, 128), (640, 640)]
    pairs = [lg_inputs.make_pair(m, n, seed=m + n) for m, n in sizes]
    batch = LG(sd, max_kpts=640, max_pairs=6, ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 640)
    assert batch.precision == 2                                 # "f16x3p1" is what an instance starts in (r05)
    batch.set_precision("f16x3")
    ref = dev.run(batch, 0.0)                                   # three MFMAs per product everywhere
    batch.set_precision("f16x3p1")
    got = {}
    for ks in (-3, -1, 102, 2):                                 # assembly / 4-wave kernel: no split, two key ranges
        batch.debug_key_split(ks)
        got[ks] = dev.run(batch, 0.0)
    for a, b in ((-3, -1), (102, 2)):
        for p, (x, y) in enumerate(zip(got[a], got[b])):
            np.testing.assert_array_equal(x[0], y[0], err_msg=f"pair {p} ks {a}/{b}")
            np.testing.assert_array_equal(x[1], y[1], err_msg=f"pair {p} ks {a}/{b}")
            np.testing.assert_array_equal(x[2], y[2])
    assert any(not np.array_equal(a[1], b[1]) for a, b in zip(ref, got[-3]))      # another arithmetic: the scores move ...
    for p, (a, b) in enumerate(zip(ref, got[-3])):                                 # ... the matches do not
        np.testing.assert_array_equal(a[0], b[0], err_msg=f"pair {p}")
        np.testing.assert_allclose(a[1], b[1], atol=2e-4)
    for p in (0, 1, 5):
        o_ij, o_sc, o_stop = _oracle(sd, pairs[p], 0.0)
        np.testing.assert_array_equal(got[-3][p][0], o_ij)
    batch.debug_key_split(0)
    dev.free(); batch.close()


@pytest.mark.parametrize("seed,kw", [(3, dict(match_gain=4.0, match_bias=3.0)),
                                     (4, dict(match_gain=4.0, match_bias=-4.6, conf_bias=2.3))])
def test_key_range_merge_inside_the_fused_ffn_equals_the_merge_launch(gpu_ctx, seed, kw):
    """r05, one pair at the bench capacity (2048): the attention's key-range partials are merged by the fused FFN's
    tiles in their prologue (ffn_fused.hpp FOLD) instead of by lg_attn_merge_h_kernel.  Same arithmetic, expression for
    expression: indices, scores and stop layer identical to the forms with the merge launch (debug_key_split(-5): the same
    assembly kernel's partials + the launch; -4: the 4-wave kernel's) - full and ragged sizes, a tile with fewer than 32 live tokens, an image of one
    keypoint, and (second weight set) pruning and an early stop that shrink the token sets from layer to layer."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(seed, **kw)
    single = LG(sd, max_kpts=2048, ctx=gpu_ctx)
    n_matches, stops = 0, []
    for m, n in [(2048, 2048), (1999, 1411), (2048, 1), (33, 2048), (1300, 1300)]:
        pr = lg_inputs.make_pair(m, n, seed=7 * m + n)
        single.debug_key_split(0)
        ij, sc, stop = single.match(*pr, min_conf=0.0)
        for form in (-5, -4):
            single.debug_key_split(form)
            r_ij, r_sc, r_stop = single.match(*pr, min_conf=0.0)
            np.testing.assert_array_equal(ij, r_ij)
            np.testing.assert_array_equal(sc, r_sc)
            assert stop == r_stop
        n_matches += len(ij); stops.append(stop)
    if seed == 3:
        assert n_matches > 100
    else:
        assert any(s_ < 9 for s_ in stops)                      # (this weight set stops early and prunes)
    assert not single.range_overflow()
    single.close()


@pytest.mark.parametrize("cap,m,n", [(2100, 2100, 1977), (3000, 2950, 3000), (2300, 33, 2300)])
def test_key_range_merge_inside_the_fused_ffn_at_other_capacities(gpu_ctx, cap, m, n):
    """The same equality at capacities whose row count is not 2048 (Kc = 2176, 3072, 2304: other tile counts per image, other
    partial strides), and the oracle's indices there."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(3, match_gain=4.0, match_bias=3.0)
    single = LG(sd, max_kpts=cap, ctx=gpu_ctx)
    pr = lg_inputs.make_pair(m, n, seed=3 * m + n)
    ij, sc, stop = single.match(*pr, min_conf=0.5)
    single.debug_key_split(-5)
    r_ij, r_sc, r_stop = single.match(*pr, min_conf=0.5)
    np.testing.assert_array_equal(ij, r_ij)
    np.testing.assert_array_equal(sc, r_sc)
    assert stop == r_stop
    o_ij, o_sc, o_stop = _oracle(sd, pr, 0.5)
    np.testing.assert_array_equal(ij, o_ij)
    assert stop == o_stop and len(ij) > 10
    assert not single.range_overflow()
    single.close()


def test_four_pair_batch_at_the_reference_default_of_4000_keypoints_equals_the_oracle(gpu_ctx):
    """`main_revamped.py:206`: --max_features defaults to 4000, and `bench.py`'s `kpts4000` leg times batches of 4 pairs at that
    size.  The batched entry at 4000 keypoints (ragged: 4000 x 3700 and 3811 x 4000 in the slots the oracle is run on; the oracle
    needs ~10 s per pair at this size, so two of the four pairs are held to it and the other two to the single-pair entry):
    index arrays equal, scores within 1e-4."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    sizes = [(4000, 3700), (4000, 4000), (2900, 3333), (3811, 4000)]
    pairs = [lg_inputs.make_pair(m, n, seed=40 + i) for i, (m, n) in enumerate(sizes)]
    batch = LG(sd, max_kpts=4000, max_pairs=4, ctx=gpu_ctx)
    dev = DevBatch(gpu_ctx, pairs, 4000)
    got = dev.run(batch, 0.1)
    for p in (0, 3):
        rij, rsc, stop = _oracle(sd, pairs[p], 0.1)
        np.testing.assert_array_equal(got[p][0].astype(np.int64), rij, err_msg=f"pair {p}")
        np.testing.assert_allclose(got[p][1], rsc, atol=1e-4)
        assert got[p][2][1] == stop and len(rij) > 500
    single = LG(sd, max_kpts=4000, ctx=gpu_ctx)
    for p in (1, 2):
        ij, sc, stop = single.match(*pairs[p], min_conf=0.1)
        np.testing.assert_array_equal(got[p][0], ij, err_msg=f"pair {p}")
        assert got[p][2][1] == stop and len(ij) > 500
    dev.free(); batch.close(); single.close()
