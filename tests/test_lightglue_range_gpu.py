"""Operand range of the split-precision LightGlue path (csrc/gemm_f16x3.hpp).

An fp32 value is carried as two fp16 planes, so a finite activation with |value| >= 65520 does
not fit.  Random-init test weights never get near that; trained checkpoints have heavier tails.
Contract tested here: for every input scale the result is EITHER fp32-grade (index arrays equal to
the oracle's on the same scaled input) OR an error is reported - never a silent inf / NaN; the
exact-fp32 path (precision 0) has no such limit and stays correct at every scale."""
import numpy as np
import pytest

import lg_inputs
from conftest import load_pkg
from oracle import lightglue_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale", [1e2, 1e3, 1e4, 1e5])
def test_scaled_descriptors_are_fp32_grade_or_reported(gpu_ctx, native, scale):
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    k0, d0, k1, d1 = lg_inputs.make_pair(256, 230, seed=12)
    d0s, d1s = (d0 * scale).astype(np.float32), (d1 * scale).astype(np.float32)   # token states scale with them
    ref = R.lightglue_forward(sd, k0, d0s, k1, d1s)
    keep = ref["scores"] > 0.1
    lg = LG(sd, max_kpts=256, ctx=gpu_ctx)
    # exact-fp32 path: always right
    lg.set_precision("f32")
    ij, sc, stop = lg.match(k0, d0s, k1, d1s, min_conf=0.1)
    np.testing.assert_array_equal(ij, ref["matches"][keep].numpy())
    assert stop == ref["stop"]
    # split path: right, or says so
    lg.set_precision("f16x3")
    try:
        ij, sc, stop = lg.match(k0, d0s, k1, d1s, min_conf=0.1)
    except native.NativeError as e:
        assert "fp16 range" in str(e)
        reported = True
    else:
        reported = False
        np.testing.assert_array_equal(ij, ref["matches"][keep].numpy())
        np.testing.assert_allclose(sc, ref["scores"][keep].numpy(), atol=1e-3)
        assert stop == ref["stop"]
    x_max = float(np.abs(R.lightglue_forward(sd, k0, d0s, k1, d1s, return_debug=True)["debug"]["x_in0"].numpy()).max())
    if x_max < 3e4:
        assert not reported, f"|x| max {x_max:.3g} fits fp16 but the call was rejected"
    if x_max > 1e5:
        assert reported, f"|x| max {x_max:.3g} cannot fit fp16 planes and nothing was reported"
    # the instance keeps working, and the flag does not stick
    ij2, _, _ = lg.match(k0, d0, k1, d1, min_conf=0.1)
    ref2 = R.lightglue_forward(sd, k0, d0, k1, d1)
    np.testing.assert_array_equal(ij2, ref2["matches"][ref2["scores"] > 0.1].numpy())
    assert lg.range_overflow() is False
    lg.close()


def test_dev_entry_raises_the_flag_instead(gpu_ctx):
    """The enqueue-only entry cannot fail synchronously: the flag is polled."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    k0, d0, k1, d1 = lg_inputs.make_pair(128, seed=3)
    lg = LG(sd, max_kpts=128, ctx=gpu_ctx)
    dev = [gpu_ctx.upload(a) for a in (k0, (d0 * 1e6).astype(np.float32), k1, (d1 * 1e6).astype(np.float32))]
    out = [gpu_ctx.malloc(128 * 8), gpu_ctx.malloc(128 * 4), gpu_ctx.malloc(16)]
    assert lg.range_overflow() is False
    lg.match_dev(dev[0], dev[1], 128, dev[2], dev[3], 128, *out)
    assert lg.range_overflow() is True
    assert lg.range_overflow() is False          # cleared by the read
    for p in dev + out:
        gpu_ctx.free(p)
    lg.close()


def test_weights_past_the_fp16_range_are_refused_at_creation(gpu_ctx, native):
    """Every weight matrix that feeds a split contraction is split once when the matcher is created (the transformer layers and,
    since the projections moved onto the split pipe, input_proj / final_proj): a weight that does not fit the planes is an error
    there, not a silent inf later."""
    W, LG = load_pkg("weights"), load_pkg("lightglue").LightGlueHIP
    sd = dict(W.random_lightglue_state_dict(1))
    sd["input_proj.weight"] = sd["input_proj.weight"] * np.float32(1e7)
    with pytest.raises(native.NativeError, match="weight"):
        LG(sd, max_kpts=128, ctx=gpu_ctx)
