"""GPU parity: HIP LightGlue vs the torch-CPU oracle, through the C-ABI.

Bar (BASELINE.json north_star): match-index arrays identical to the oracle's;
floating-point intermediates (token states, similarity, scores) within 1e-3."""
import numpy as np
import pytest

import lg_inputs
from conftest import load_pkg
from oracle import lightglue_ref as R

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def W():
    return load_pkg("weights")


@pytest.fixture(scope="module")
def LG():
    return load_pkg("lightglue").LightGlueHIP


def _compare(lg, sd, k0, d0, k1, d1, min_conf, conf=None, check_state=True):
    ij, sc, stop = lg.match(k0, d0, k1, d1, min_conf=min_conf)
    ref = R.lightglue_forward(sd, k0, d0, k1, d1, conf, return_debug=True)
    keep = ref["scores"] > min_conf
    ref_ij = ref["matches"][keep].numpy()
    ref_sc = ref["scores"][keep].numpy()
    assert stop == ref["stop"]
    np.testing.assert_array_equal(ij, ref_ij)            # bit-exact indices
    np.testing.assert_allclose(sc, ref_sc, atol=TOL, rtol=TOL)
    assert np.all(np.diff(ij[:, 0]) > 0) if len(ij) > 1 else True
    if check_state and "x_out0" in ref["debug"]:
        Kc = lg.capacity
        x = lg.debug_read(0, (2, Kc, 256))
        n0, n1 = ref["debug"]["x_out0"].shape[0], ref["debug"]["x_out1"].shape[0]
        info = lg.debug_read(4, (4,), np.int32)
        assert (info[2], info[3]) == (n0, n1)
        np.testing.assert_allclose(x[0, :n0], ref["debug"]["x_out0"].numpy(), atol=TOL, rtol=TOL)
        np.testing.assert_allclose(x[1, :n1], ref["debug"]["x_out1"].numpy(), atol=TOL, rtol=TOL)
        sim = lg.debug_read(1, (Kc, Kc))[:n0, :n1]
        np.testing.assert_allclose(sim, ref["debug"]["sim"].numpy(), atol=TOL, rtol=TOL)
    return ij, ref


@pytest.mark.parametrize("m,n", [(512, 512), (300, 417), (64, 33), (1, 5), (129, 128)])
def test_matches_bit_exact_vs_oracle(W, LG, m, n):
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=640)
    k0, d0, k1, d1 = lg_inputs.make_pair(m, n, seed=m + n)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.7)
    if min(m, n) >= 64:
        assert len(ij) > 0.2 * min(m, n)       # the test is not vacuous
    # min_conf = 0 keeps everything above LightGlue's own 0.1 filter
    _compare(lg, sd, k0, d0, k1, d1, min_conf=0.0, check_state=False)
    lg.close()


def test_c2_size_2048(W, LG):
    """BASELINE config C2: 2048 keypoints per image, all 9 layers."""
    sd = W.random_lightglue_state_dict(2, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=2048)
    k0, d0, k1, d1 = lg_inputs.make_pair(2048, seed=11)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.7)
    assert ref["stop"] == 9 and len(ij) > 100
    lg.close()


def test_early_stop_layer_is_respected(W, LG):
    # confident tokens everywhere -> upstream stops after the first layer (stop == 1)
    sd = W.random_lightglue_state_dict(3, match_gain=4.0, match_bias=3.0, conf_bias=12.0)
    lg = LG(sd, max_kpts=512)
    k0, d0, k1, d1 = lg_inputs.make_pair(384, seed=5)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.3)
    assert ref["stop"] == 1
    lg.close()


def test_point_pruning_matches_oracle(W, LG):
    # matchability logits centred below the 0.01 keep threshold and confident tokens just
    # under the stop ratio -> some points are pruned after each layer
    sd = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=-4.6, conf_bias=2.3)
    lg = LG(sd, max_kpts=512)
    k0, d0, k1, d1 = lg_inputs.make_pair(400, 350, seed=6)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.0)
    n0, n1 = ref["debug"]["x_out0"].shape[0], ref["debug"]["x_out1"].shape[0]
    assert (n0 < 400 or n1 < 350), "pruning did not trigger in the oracle - test is vacuous"
    Kc = lg.capacity
    ind = lg.debug_read(2, (2, Kc), np.int32)
    np.testing.assert_array_equal(ind[0, :n0], ref["debug"]["ind0"].numpy())
    np.testing.assert_array_equal(ind[1, :n1], ref["debug"]["ind1"].numpy())
    pr = lg.debug_read(3, (2, Kc), np.int32)
    np.testing.assert_array_equal(pr[0, :400], ref["prune0"].numpy())
    np.testing.assert_array_equal(pr[1, :350], ref["prune1"].numpy())
    lg.close()


def test_pruning_switch_off(W, LG):
    sd = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=-4.6, conf_bias=2.3)
    lg = LG(sd, max_kpts=512, prune_min_kpts=100000)
    k0, d0, k1, d1 = lg_inputs.make_pair(256, seed=7)
    _compare(lg, sd, k0, d0, k1, d1, min_conf=0.0, conf={"prune_min_kpts": 100000})
    lg.close()


def test_empty_inputs_and_capacity_errors(W, LG, native):
    lg = LG(W.random_lightglue_state_dict(0), max_kpts=128)
    k0, d0, k1, d1 = lg_inputs.make_pair(16, seed=1)
    ij, sc, stop = lg.match(k0[:0], d0[:0], k1, d1)
    assert ij.shape == (0, 2) and sc.shape == (0,)
    big = lg_inputs.make_pair(300, seed=2)
    with pytest.raises(native.NativeError, match="exceed"):
        lg.match(*big)
    lg.close()


def test_full_size_properties_self_match_and_symmetry(W, LG):
    """Size-independent properties at the C2 size (2048 x 2048), no oracle involved:
    (1) an image matched against itself gives the identity assignment;
    (2) LightGlue is symmetric in its two inputs (shared weights): match(B, A) is match(A, B)
        with the index columns swapped."""
    sd = W.random_lightglue_state_dict(5, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=2048)
    k0, d0, k1, d1 = lg_inputs.make_pair(2048, seed=21)
    ij, sc, stop = lg.match(k0, d0, k0, d0, min_conf=0.0)
    assert len(ij) > 1500 and np.array_equal(ij[:, 0], ij[:, 1])
    ab, sab, _ = lg.match(k0, d0, k1, d1, min_conf=0.0)
    ba, sba, _ = lg.match(k1, d1, k0, d0, min_conf=0.0)
    ba_sw = ba[:, ::-1]
    order = np.argsort(ba_sw[:, 0], kind="stable")
    np.testing.assert_array_equal(ab, ba_sw[order])
    np.testing.assert_allclose(sab, sba[order], atol=1e-5)
    # both arithmetic paths agree on the indices
    lg.set_precision("f32")
    ab32, sab32, _ = lg.match(k0, d0, k1, d1, min_conf=0.0)
    np.testing.assert_array_equal(ab, ab32)
    np.testing.assert_allclose(sab, sab32, atol=1e-4)
    lg.close()


def test_capacity_edge_and_repeatability(W, LG):
    """M = N = capacity exactly (not a multiple of the 128-row tiles' interior), repeated calls on
    one instance give bit-identical output (no stale state between calls)."""
    sd = W.random_lightglue_state_dict(6, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=300)
    assert lg.capacity == 384
    k0, d0, k1, d1 = lg_inputs.make_pair(384, seed=22)
    a = lg.match(k0, d0, k1, d1, min_conf=0.1)
    big = lg_inputs.make_pair(300, 77, seed=23)
    lg.match(*big, min_conf=0.1)                                   # a different, smaller problem in between
    b = lg.match(k0, d0, k1, d1, min_conf=0.1)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    _compare(lg, sd, k0, d0, k1, d1, min_conf=0.1, check_state=False)
    lg.close()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_token_state_per_layer_is_fp32_grade(W, LG, precision):
    """Token state after each half layer against the oracle, at fp32-rounding tolerance (2e-5
    absolute on O(1) values; the 1e-3 north-star bar is 50x looser).  Guards the split path's
    operand handling: fp16 subnormal MFMA operands are flushed by the hardware, which once cost
    two decimal digits here (4e-4 after nine layers)."""
    sd = W.random_lightglue_state_dict(5, match_gain=4.0, match_bias=3.0)
    n = 512
    k0, d0, k1, d1 = lg_inputs.make_pair(n, seed=21)
    ref = R.lightglue_forward(sd, k0, d0, k1, d1, {"depth_confidence": -1, "width_confidence": -1},
                              return_debug=True)
    lg = LG(sd, max_kpts=n, depth_confidence=-1.0, width_confidence=-1.0)
    lg.set_precision(precision)
    for layer in (1, 4, 9):
        for self_only in (True, False):
            lg.debug_layers(layer, self_only)
            lg.match(k0, d0, k1, d1, min_conf=0.0)
            x = lg.debug_read(0, (2, lg.capacity, 256))
            key = "self" if self_only else "cross"
            for img in (0, 1):
                want = ref["debug"]["layers"][layer - 1][f"{key}{img}"].numpy()
                np.testing.assert_allclose(x[img, :n], want, atol=2e-5, rtol=0,
                                           err_msg=f"{precision} layer {layer} {key} image {img}")
    lg.close()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_non_finite_input_gives_no_fault(W, LG, precision):
    """NaN / Inf descriptors or keypoints must not crash the device: rows without a finite
    arg-max simply produce no match, and the instance keeps working afterwards."""
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=256)
    lg.set_precision(precision)
    k0, d0, k1, d1 = lg_inputs.make_pair(200, 180, seed=3)
    good = lg.match(k0, d0, k1, d1, min_conf=0.1)
    bad_d = d0.copy(); bad_d[:] = np.nan
    ij, sc, _ = lg.match(k0, bad_d, k1, d1, min_conf=0.1)
    assert len(ij) == 0
    bad_k = k1.copy(); bad_k[5] = np.inf
    ij, sc, _ = lg.match(k0, d0, bad_k, d1, min_conf=0.1)
    assert ij.shape[1] == 2 and (len(ij) == 0 or (ij[:, 0].max() < 200 and ij[:, 1].max() < 180))
    again = lg.match(k0, d0, k1, d1, min_conf=0.1)
    np.testing.assert_array_equal(good[0], again[0])
    np.testing.assert_array_equal(good[1], again[1])
    # stale non-finite rows of a larger, poisoned problem must not leak into a smaller one
    small = lg.match(k0[:70], d0[:70], k1[:90], d1[:90], min_conf=0.1)
    lg.match(k0, bad_d, k1, bad_d[:180], min_conf=0.1)
    small2 = lg.match(k0[:70], d0[:70], k1[:90], d1[:90], min_conf=0.1)
    np.testing.assert_array_equal(small[0], small2[0])
    np.testing.assert_array_equal(small[1], small2[1])
    lg.close()


def test_reference_default_size_4000_keypoints(W, LG):
    """`max_features` defaults to 4000 in the reference (features_utils.py:25): capacity 4096, ragged
    3 700 x 4 000 pair, index arrays against the oracle."""
    sd = W.random_lightglue_state_dict(9, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=4000)
    assert lg.capacity == 4096
    k0, d0, k1, d1 = lg_inputs.make_pair(3700, 4000, seed=31)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.7, check_state=False)
    assert len(ij) > 500
    lg.close()


def test_c2_size_with_pruning_active(W, LG):
    """2048 x 1900 keypoints with point pruning shrinking both token sets layer by layer (ragged,
    changing row counts through every GEMM / attention launch of the split path): compaction
    order, prune counters and match indices against the oracle."""
    sd = W.random_lightglue_state_dict(4, match_gain=4.0, match_bias=-4.6, conf_bias=2.3)
    lg = LG(sd, max_kpts=2048)
    k0, d0, k1, d1 = lg_inputs.make_pair(2048, 1900, seed=8)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.0, check_state=False)
    n0, n1 = ref["debug"]["x_out0"].shape[0], ref["debug"]["x_out1"].shape[0]
    assert n0 < 2048 and n1 < 1900, "pruning did not trigger in the oracle - test is vacuous"
    Kc = lg.capacity
    ind = lg.debug_read(2, (2, Kc), np.int32)
    np.testing.assert_array_equal(ind[0, :n0], ref["debug"]["ind0"].numpy())
    np.testing.assert_array_equal(ind[1, :n1], ref["debug"]["ind1"].numpy())
    pr = lg.debug_read(3, (2, Kc), np.int32)
    np.testing.assert_array_equal(pr[0, :2048], ref["prune0"].numpy())
    np.testing.assert_array_equal(pr[1, :1900], ref["prune1"].numpy())
    lg.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_index_parity_over_seeds(W, LG, seed):
    """Breadth: different random weights AND inputs at 1024 x 960 keypoints; the match-index arrays
    must equal the oracle's every time (near-ties are where a non-fp32-grade path would flip)."""
    sd = W.random_lightglue_state_dict(seed, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=1024)
    k0, d0, k1, d1 = lg_inputs.make_pair(1024, 960, seed=seed)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.2, check_state=False)
    assert len(ij) > 100
    lg.close()


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_index_parity_over_seeds_f16x3(W, LG, seed):
    """The same six-seed set in the other split form, "f16x3" (three MFMAs per product in P.V too; the default until r04):
    same match indices as the oracle.  (The test above runs the shipped default, "f16x3p1".)"""
    sd = W.random_lightglue_state_dict(seed, match_gain=4.0, match_bias=3.0)
    lg = LG(sd, max_kpts=1024)
    assert lg.precision == 2
    lg.set_precision("f16x3")
    k0, d0, k1, d1 = lg_inputs.make_pair(1024, 960, seed=seed)
    ij, ref = _compare(lg, sd, k0, d0, k1, d1, min_conf=0.2, check_state=False)
    assert len(ij) > 100
    lg.close()


def test_token_state_f16x3p1_is_within_3e_5(W, LG):
    """What "f16x3p1" costs: the token state after 9 layers against the oracle stays within 3e-5 ("f16x3": 2e-5 bound,
    ~8e-6 measured; profiles/r04_split_study.md: 2.4e-5 for this form)."""
    sd = W.random_lightglue_state_dict(5, match_gain=4.0, match_bias=3.0)
    n = 512
    k0, d0, k1, d1 = lg_inputs.make_pair(n, seed=21)
    ref = R.lightglue_forward(sd, k0, d0, k1, d1, {"depth_confidence": -1, "width_confidence": -1}, return_debug=True)
    lg = LG(sd, max_kpts=n, depth_confidence=-1.0, width_confidence=-1.0)
    lg.set_precision("f16x3p1")
    ij, sc, stop = lg.match(k0, d0, k1, d1, min_conf=0.0)
    x = lg.debug_read(0, (2, lg.capacity, 256))
    for img in (0, 1):
        np.testing.assert_allclose(x[img, :n], ref["debug"]["layers"][8][f"cross{img}"].numpy(), atol=3e-5, rtol=0)
    np.testing.assert_array_equal(ij, ref["matches"].numpy())
    lg.close()
