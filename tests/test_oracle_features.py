"""CPU checks of the ALIKED / LightGlue oracles (parity unpinned: no upstream package or
checkpoint exists in this image) - structural invariants and a cross-check of the LightGlue
arithmetic against the independent HuggingFace port that ships with `transformers`."""
import numpy as np
import pytest
import torch

import frames
import lg_inputs
from conftest import load_pkg
from oracle import aliked_ref as A
from oracle import lightglue_ref as L


@pytest.fixture(scope="module")
def W():
    return load_pkg("weights")


def test_lightglue_flops_formula_matches_survey():
    assert abs(L.flops(2048, 9) / 1e9 - 249.4) < 0.1
    assert abs(L.flops(1024, 9) / 1e9 - 85.5) < 0.1


def test_lightglue_invariants(W):
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    k0, d0, k1, d1 = lg_inputs.make_pair(256, 200, seed=3)
    out = L.lightglue_forward(sd, k0, d0, k1, d1, return_debug=True)
    m = out["matches"].numpy()
    assert len(m) > 20
    assert np.all(np.diff(m[:, 0]) > 0)                          # ascending in index 0
    assert len(set(m[:, 1])) == len(m)                           # mutual NN -> injective
    assert np.all(out["scores"].numpy() > 0.1)
    sc = out["debug"]["log_scores"].numpy()
    assert np.all(sc[:-1, :-1] <= 1e-6)                          # log-probabilities
    # keypoint normalisation without image_size (features_utils.py:158-161 passes none):
    # size = 1 + max - min of the keypoints themselves, shift = size / 2, scale = max(size) / 2
    kn = out["debug"]["kn0"].numpy()
    size = 1 + k0.max(0) - k0.min(0)
    np.testing.assert_allclose(kn, (k0 - size / 2) / (size.max() / 2), rtol=1e-6, atol=1e-6)


def test_lightglue_early_stop_and_pruning_paths(W):
    k0, d0, k1, d1 = lg_inputs.make_pair(128, seed=4)
    sd = W.random_lightglue_state_dict(3, conf_bias=12.0)
    assert L.lightglue_forward(sd, k0, d0, k1, d1)["stop"] == 1
    sd = W.random_lightglue_state_dict(3, conf_bias=-12.0)
    assert L.lightglue_forward(sd, k0, d0, k1, d1)["stop"] == 9
    sd = W.random_lightglue_state_dict(4, match_bias=-4.6, conf_bias=2.3)
    out = L.lightglue_forward(sd, k0, d0, k1, d1, return_debug=True)
    assert out["debug"]["x_out0"].shape[0] < 128                 # points were pruned
    assert out["prune0"].max() > out["prune0"].min()
    off = L.lightglue_forward(sd, k0, d0, k1, d1, {"prune_min_kpts": 10 ** 6}, return_debug=True)
    assert off["debug"]["x_out0"].shape[0] == 128


def test_reference_matcher_filters_by_min_conf_and_handles_empty(W):
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    k0, d0, k1, d1 = lg_inputs.make_pair(128, seed=5)
    ij_all, sc_all, _ = L.reference_feature_matcher(sd, k0, k1, d0, d1, min_conf=0.0)
    ij_hi, sc_hi, _ = L.reference_feature_matcher(sd, k0, k1, d0, d1, min_conf=0.7)
    assert len(ij_hi) < len(ij_all) and np.all(sc_hi > 0.7)
    assert len(L.reference_feature_matcher(sd, k0[:0], k1, d0[:0], d1)[0]) == 0


def test_rotary_and_double_softmax_match_hf_port():
    """`transformers` ships an independent port of LightGlue; its rotary embedding and
    sigmoid_log_double_softmax must agree with the restatement (SURVEY 8(c) cross-check)."""
    hf = pytest.importorskip("transformers.models.lightglue.modeling_lightglue")
    torch.manual_seed(0)
    sim = torch.randn(1, 7, 5); z0 = torch.randn(1, 7, 1); z1 = torch.randn(1, 5, 1)
    ours = L.sigmoid_log_double_softmax(sim, z0, z1)
    theirs = hf.sigmoid_log_double_softmax(sim, z0, z1)
    torch.testing.assert_close(ours, theirs)
    x = torch.randn(1, 4, 6, 64)
    torch.testing.assert_close(L.rotate_half(x), hf.rotate_half(x))


def test_aliked_invariants(W):
    sd = W.random_aliked_state_dict(0)
    out = A.aliked_extract(sd, frames.structured_frame(0, h=120, w=200), 256, return_debug=True)
    kp, desc = out["keypoints"], out["descriptors"]
    assert len(kp) == 256 and desc.shape == (256, 128)
    np.testing.assert_allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)
    h, w = out["debug"]["score_map"].shape[-2:]
    iy, ix = np.divmod(out["indices"], w)
    assert ix.min() >= 2 and iy.min() >= 2 and ix.max() <= w - 3 and iy.max() <= h - 3   # border of radius 2
    assert np.all(np.diff(out["scores"]) <= 1e-3)        # score order (up to sub-pixel resampling)
    # resize plan: long side -> 1024, centred replicate padding to /32
    p = A.resize_plan(376, 1241)
    assert (p["new_h"], p["new_w"]) == (310, 1024) and (p["ky"], p["kx"]) == (3, 3)
    assert A.pad_amounts(310, 1024) == [0, 0, 5, 5]
    assert abs(A.flops_dense() / 1e9 - 6.56) < 0.01


def test_deform_conv_reduces_to_conv_at_zero_offset():
    x = torch.randn(1, 8, 12, 20); w = torch.randn(5, 8, 3, 3)
    out = A.deform_conv2d(x, torch.zeros(1, 18, 12, 20), w)
    torch.testing.assert_close(out, torch.nn.functional.conv2d(x, w, padding=1), atol=1e-5, rtol=1e-5)


def test_weight_packing_is_pure_reindexing(W):
    sd = W.random_lightglue_state_dict(0)
    blob = W.pack_lightglue(sd)
    assert blob.dtype == np.float32 and blob.size % 64 == 0
    perm = W._qkv_row_perm()
    assert sorted(perm.tolist()) == list(range(768))
    # [s][h][d] ordering: first 64 rows are q of head 0
    np.testing.assert_array_equal(perm[:3], [0, 3, 6])
    sda = W.random_aliked_state_dict(0)
    ab = W.pack_aliked(sda)
    assert ab.size % 64 == 0 and np.isfinite(ab).all()
