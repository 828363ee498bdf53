"""Operand range of ALIKED's split-precision stages (csrc/gemm_f16x3.hpp planes in csrc/aliked_kernels.hip): block1.conv2,
block2, the offset and deformable convolutions and the descriptor-head GEMMs carry fp32 operands as fp16 (hi, lo) planes, so a
finite activation with |value| >= 65520 does not fit.  Contract (the matcher's, tests/test_lightglue_range_gpu.py): for every
scale the result is EITHER fp32-grade against the oracle OR reported - the host entry fails, the device entries leave
count -1 and raise the instance's sticky word; weights that do not fit are refused at creation.  Never a silent inf / NaN."""
import numpy as np
import pytest

import frames
from conftest import load_pkg
from oracle import aliked_ref as R

pytestmark = pytest.mark.gpu
IMG = np.ascontiguousarray(frames.structured_frame(3)[:200, :320])


def _scaled(W, scale):
    sd = W.random_aliked_state_dict(0)
    sd["block1.bn1.weight"] = (sd["block1.bn1.weight"] * scale).astype(np.float32)      # conv1's activations grow by `scale`
    sd["block1.bn1.bias"] = (sd["block1.bn1.bias"] * scale).astype(np.float32)
    return sd


@pytest.mark.parametrize("scale", [1e1, 1e2, 1e3, 1e5])
def test_scaled_activations_are_fp32_grade_or_reported(gpu_ctx, native, scale):
    W, AL = load_pkg("weights"), load_pkg("aliked").AlikedHIP
    sd = _scaled(W, scale)
    al = AL(sd, max_num_keypoints=1024, max_h=256, max_w=320, ctx=gpu_ctx)
    ref = R.aliked_extract(sd, IMG, 1024, return_debug=True)["debug"]
    peak = max(float(np.abs(ref[k][0].numpy()).max()) for k in ("x1", "x2", "x3", "x4"))
    try:
        xy, desc = al.extract(IMG, 1024)
    except native.NativeError as e:
        assert "fp16 range" in str(e)
        reported = True
    else:
        reported = False
        assert np.isfinite(desc).all() and np.isfinite(xy).all()
        np.testing.assert_allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)
        d = al.debug_read(2, (8,), np.int32)
        Hp, Wp = int(d[2]), int(d[3])
        for which, name, div, ch in ((3, "x1", 1, 16), (4, "x2", 2, 32), (5, "x3", 8, 64), (6, "x4", 32, 128)):
            if name in ("x3", "x4") and scale > 10:
                # the deformable stages sample at positions that are offset-conv OUTPUTS in pixels: with activations x 100
                # an offset is a sum of ~1e5-sized terms, fp32 rounding alone moves it by ~1e-2 px and the sampled map has
                # values in the thousands - the fp32 oracle is as ill-conditioned there as the split path; the dense
                # stages in front of them carry the comparison at these scales
                assert np.isfinite(al.debug_read(which, (ch, Hp // div, Wp // div))).all()
                continue
            want = ref[name][0].numpy()
            got = al.debug_read(which, (ch, Hp // div, Wp // div))
            # rtol on the value + atol relative to the stage's largest magnitude (an output is a cancelling sum of terms of
            # that size): 1e-4 for the plain convolutions, 1e-3 for the deformable stages
            rel = 1e-4 if name in ("x1", "x2") else 1e-3
            np.testing.assert_allclose(got, want, rtol=1e-3, atol=rel * max(1.0, float(np.abs(want).max())), err_msg=name)
    if peak < 2e4:
        assert not reported, f"largest stage activation {peak:.3g} fits the fp16 planes but the call was rejected"
    if scale >= 1e5:
        assert reported, f"conv1 activations of ~{scale:g} cannot fit the fp16 planes and nothing was reported"
    # the flag does not stick, and the instance keeps working on the device entry too
    assert al.range_overflow() is False
    al.close()


def test_device_entry_leaves_count_minus_one_and_the_sticky_word(gpu_ctx, native):
    W, AL, LG = load_pkg("weights"), load_pkg("aliked").AlikedHIP, load_pkg("lightglue").LightGlueHIP
    al = AL(_scaled(W, 1e5), max_num_keypoints=512, max_h=256, max_w=320, ctx=gpu_ctx)
    ctx = gpu_ctx
    img_d = ctx.upload(IMG)
    xy, desc, sc, cnt = ctx.malloc(512 * 8), ctx.malloc(512 * 512), ctx.malloc(512 * 4), ctx.malloc(16)
    al.extract_dev(img_d, IMG.shape[0], IMG.shape[1], 3, xy, desc, sc, cnt)
    n = np.zeros(4, np.int32); ctx.d2h(n, cnt)
    assert n[0] == -1
    assert al.range_overflow() is True and al.range_overflow() is False
    # a matcher fed the void frame treats it as empty (negative counts clamp to 0): no matches, no fault
    lg = LG(W.random_lightglue_state_dict(1), max_kpts=512, ctx=ctx)
    ij, msc, info = ctx.malloc(512 * 8), ctx.malloc(512 * 4), ctx.malloc(16)
    lg.match_dev(xy, desc, 512, xy, desc, 512, ij, msc, info, m_dev=cnt, n_dev=cnt)
    ctx.sync()
    out = np.zeros(4, np.int32); ctx.d2h(out, info)
    assert out[0] == 0
    lg.close(); al.close()
    for p in (img_d, xy, desc, sc, cnt, ij, msc, info):
        ctx.free(p)


def test_drop_in_extractor_raises_on_a_void_frame(gpu_ctx, native, monkeypatch):
    fu = load_pkg("slam.core.features_utils")
    W, AL = load_pkg("weights"), load_pkg("aliked").AlikedHIP
    from types import SimpleNamespace
    al = AL(_scaled(W, 1e5), max_num_keypoints=512, max_h=256, max_w=320, ctx=gpu_ctx)
    args = SimpleNamespace(use_lightglue=True)
    with pytest.raises(native.NativeError, match="fp16 range"):
        fu.feature_extractor(args, IMG, al)
    assert al.range_overflow() is False                       # reported once, cleared
    al.close()


def test_weights_that_do_not_fit_are_refused_at_creation(gpu_ctx, native):
    W, AL = load_pkg("weights"), load_pkg("aliked").AlikedHIP
    for key in ("block2.conv2.weight", "block3.conv1.regular_conv.weight", "block4.conv2.offset_conv.weight", "desc_head.agg_weights"):
        sd = W.random_aliked_state_dict(0)
        sd[key] = (sd[key] * 1e8).astype(np.float32)          # (the smallest of these weights are ~1e-2: far past 65520 then)
        with pytest.raises(native.NativeError, match="65520"):
            AL(sd, max_num_keypoints=256, max_h=128, max_w=160, ctx=gpu_ctx)
    # a BN scale folded into deformable-conv weights counts too
    sd = W.random_aliked_state_dict(0)
    sd["block3.bn1.weight"] = (sd["block3.bn1.weight"] * 1e8).astype(np.float32)
    with pytest.raises(native.NativeError, match="65520"):
        AL(sd, max_num_keypoints=256, max_h=128, max_w=160, ctx=gpu_ctx)
