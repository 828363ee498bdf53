"""GPU parity of `sslam_fmat_ransac_host` with the restated cv2.findFundamentalMat
(oracle/ransac_ref.py): same sample stream, same winning sample, same iteration count, identical
inlier masks (bit-exact index work; the two null-space methods agree to rounding, so a match whose
error sits within 1e-6 relative of the threshold is the only place they could differ)."""
import numpy as np
import pytest

import two_view
from conftest import load_pkg
from oracle import ransac_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    return load_pkg("epipolar")


def _check(E, p1, p2, thresh=1.0, conf=0.99):
    F, mask, info = E.find_fundamental_ransac(p1, p2, thresh, conf)
    Fr, mr, ir = R.find_fundamental_ransac(p1, p2, thresh, conf)
    assert (F is None) == (Fr is None)
    assert info["lmeds"] == ir["lmeds"]
    if F is None:
        assert mask is None and mr is None
        return None
    assert info["sample"] == ir["sample"] and info["iterations"] == ir["iterations"]
    np.testing.assert_array_equal(mask, mr)
    assert info["inliers"] == int(mr.sum())
    s = np.abs(Fr).max()
    np.testing.assert_allclose(F / s, Fr / s, atol=1e-7)
    return mask


# (600 matches at 65 % outliers: the loop runs through all three chunks of the launch sequence - budget in the hundreds)
@pytest.mark.parametrize("n,frac,seed", [(2048, 0.3, 0), (400, 0.5, 1), (100, 0.1, 2), (15, 0.2, 3), (900, 0.0, 4), (600, 0.65, 7)])
def test_ransac_matches_oracle(E, n, frac, seed):
    p1, p2, truth = two_view.make_matches(n, outlier_frac=frac, noise=0.3, seed=seed)
    mask = _check(E, p1, p2)
    assert mask is not None and mask.sum() >= 7
    if n >= 100:
        assert (mask & truth).sum() >= 0.75 * truth.sum()


def test_lmeds_branch_matches_oracle_at_fourteen_points(E):
    p1, p2, _ = two_view.make_matches(14, outlier_frac=0.15, noise=0.2, seed=2)
    _check(E, p1, p2)


@pytest.mark.parametrize("n,seed", [(8, 0), (11, 1), (13, 2)])
def test_lmeds_branch_below_fourteen_points_is_self_consistent(E, n, seed):
    """With 8..13 matches the median of the errors falls on one of the 7 sample points, which lie
    ON the model: every hypothesis has a median of rounding noise and OpenCV's argmin is arbitrary
    (so is ours - the two null-space methods round differently).  What is checked instead: the
    LMedS branch is taken, and the mask is exactly the LMedS rule applied to the returned F."""
    p1, p2, _ = two_view.make_matches(n, outlier_frac=0.15, noise=0.2, seed=seed)
    F, mask, info = E.find_fundamental_ransac(p1, p2, 1.0, 0.99)
    Fr, mr, ir = R.find_fundamental_ransac(p1, p2, 1.0, 0.99)
    assert info["lmeds"] and ir["lmeds"] and F is not None and Fr is not None
    assert info["iterations"] == ir["iterations"] == R.update_num_iters(0.99, 0.45, 7, 1000)
    e = R.compute_error(p1, p2, F)
    es = np.sort(e)
    med = float(es[n // 2]) if n % 2 else float((es[n // 2 - 1] + es[n // 2]) * np.float32(0.5))
    sigma = max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * np.sqrt(med), 0.001)
    np.testing.assert_array_equal(mask, e <= np.float32(sigma * sigma))
    assert mask.sum() >= 7 and mr.sum() >= 7


def test_thresholds_defaults_and_degenerate_inputs(E, native):
    p1, p2, _ = two_view.make_matches(300, outlier_frac=0.4, noise=0.5, seed=9)
    for thr, conf in [(0.5, 0.99), (3.0, 0.999), (-1.0, 2.0)]:           # last: cv2's own defaulting
        _check(E, p1, p2, thr, conf)
    z = np.zeros((40, 2), np.float32)
    assert _check(E, z, z) is None                                        # no admissible subset
    line = np.stack([np.arange(40, dtype=np.float32), 2 * np.arange(40, dtype=np.float32)], 1)
    _check(E, line, line + 1)                                             # collinear: every subset rejected
    with pytest.raises(native.NativeError, match="need >= 8"):
        E.find_fundamental_ransac(p1[:5], p2[:5])


def test_filter_matches_ransac_drop_in(E):
    fu = load_pkg("slam.core.features_utils")
    types = load_pkg("slam.core.types")
    p1, p2, truth = two_view.make_matches(500, outlier_frac=0.3, noise=0.3, seed=11)
    kp1 = [types.KeyPoint(float(x), float(y), 1.0) for x, y in p1]
    perm = np.random.default_rng(0).permutation(len(p2))
    kp2 = [None] * len(p2)
    for j, i in enumerate(perm):
        kp2[i] = types.KeyPoint(float(p2[j][0]), float(p2[j][1]), 1.0)
    matches = [types.DMatch(j, int(perm[j]), 0, 0.0) for j in range(len(p1))]
    kept = fu.filter_matches_ransac(kp1, kp2, matches, 1.0)
    _, mr, _ = R.find_fundamental_ransac(p1, p2, 1.0, 0.99)
    assert [m.queryIdx for m in kept] == list(np.flatnonzero(mr))
    assert all(m.trainIdx == perm[m.queryIdx] for m in kept)
    assert fu.filter_matches_ransac(kp1, kp2, matches[:5], 1.0) == matches[:5]      # < 8: unchanged (reference :189)


def _dev_filter(E, ctx, kp1, kp2, ij, n_used, thresh=1.0):
    """Run sslam_fmat_ransac_dev on device copies of (kp1, kp2, ij[, count]) and read its outputs back."""
    n_max = len(ij)
    d = [ctx.upload(np.ascontiguousarray(a)) for a in (kp1.astype(np.float32), kp2.astype(np.float32),
                                                        ij.astype(np.int32), np.array([n_used], np.int32))]
    out_ij, info, mask, F = ctx.malloc(n_max * 8), ctx.malloc(16), ctx.malloc(n_max), ctx.malloc(72)
    E.filter_matches_dev(ctx, n_max, d[3], d[0], d[1], d[2], out_ij, info, thresh=thresh, mask_out_dev=mask, F_out_dev=F)
    h_ij, h_info = np.empty((n_max, 2), np.int32), np.empty(4, np.int32)
    h_mask, h_F = np.empty(n_max, np.uint8), np.empty(9)
    ctx.d2h(h_ij, out_ij); ctx.d2h(h_info, info); ctx.d2h(h_mask, mask); ctx.d2h(h_F, F)
    for p in d + [out_ij, info, mask, F]:
        ctx.free(p)
    return h_ij[:h_info[0]], h_info, h_mask[:n_used].astype(bool), h_F.reshape(3, 3)


# (the last three: a loop that needs all three chunks; more matches than the head kernel's LDS image holds - the gather runs as
#  a launch of its own and the sampler reads global memory; a bound above that limit with a device count below it)
@pytest.mark.parametrize("n,frac,seed,spare", [(2048, 0.3, 0, 0), (400, 0.5, 1, 37), (15, 0.2, 3, 5), (14, 0.15, 2, 0),
                                               (600, 0.65, 7, 0), (5000, 0.3, 5, 0), (3000, 0.4, 6, 2000)])
def test_device_resident_filter_equals_the_host_entry(E, gpu_ctx, n, frac, seed, spare):
    """sslam_fmat_ransac_dev consumes the matcher's device outputs (keypoints, index pairs, a device
    count that may be below the buffer bound) and must leave exactly what the host entry returns for
    the gathered points: same mask, same F, same winning sample, the kept pairs in order."""
    p1, p2, _ = two_view.make_matches(n, outlier_frac=frac, noise=0.3, seed=seed)
    rng = np.random.default_rng(seed)
    # scatter the matched points into two keypoint arrays, as a matcher would index them
    n1, n2 = n + 50, n + 80
    q, t = rng.permutation(n1)[:n], rng.permutation(n2)[:n]
    kp1 = rng.uniform(0, 1000, (n1, 2)).astype(np.float32); kp1[q] = p1
    kp2 = rng.uniform(0, 1000, (n2, 2)).astype(np.float32); kp2[t] = p2
    ij = np.stack([q, t], 1).astype(np.int32)
    ij_buf = np.concatenate([ij, np.zeros((spare, 2), np.int32)])          # capacity beyond the device count
    kept, info, mask, F = _dev_filter(E, gpu_ctx, kp1, kp2, ij_buf, n)
    F_h, mask_h, info_h = E.find_fundamental_ransac(p1, p2, 1.0, 0.99, ctx=gpu_ctx)
    assert mask_h is not None
    np.testing.assert_array_equal(mask, mask_h)
    np.testing.assert_array_equal(kept, ij[mask_h])
    assert info[0] == mask_h.sum() and info[1] == info_h["iterations"] and info[3] == info_h["sample"]
    assert bool(info[2]) == info_h["lmeds"]
    np.testing.assert_array_equal(F, F_h)


def test_device_resident_filter_conventions(E, gpu_ctx):
    """features_utils.py:189-190: fewer than 8 matches come back unfiltered; :196-197: no model -> []."""
    rng = np.random.default_rng(0)
    kp = rng.uniform(0, 500, (40, 2)).astype(np.float32)
    ij = np.stack([np.arange(20), np.arange(20)], 1).astype(np.int32)
    for n_used in (0, 1, 7):
        kept, info, mask, _ = _dev_filter(E, gpu_ctx, kp, kp + 3, ij, n_used)
        assert info[0] == n_used and info[3] == -2 and mask.all()
        np.testing.assert_array_equal(kept, ij[:n_used])
    z = np.zeros((40, 2), np.float32)                                       # no admissible subset: cv2 returns mask None
    kept, info, _, F = _dev_filter(E, gpu_ctx, z, z, ij, 20)
    assert info[0] == 0 and info[3] == -1 and len(kept) == 0 and not F.any()


def test_device_resident_filter_optional_outputs_and_fixed_count(E, gpu_ctx, native):
    """n_dev = NULL means exactly n_max matches; mask / F / pair outputs are optional; bad arguments fail loudly."""
    p1, p2, _ = two_view.make_matches(200, outlier_frac=0.3, noise=0.3, seed=8)
    ij = np.stack([np.arange(200), np.arange(200)], 1).astype(np.int32)
    ctx = gpu_ctx
    d = [ctx.upload(np.ascontiguousarray(a)) for a in (p1.astype(np.float32), p2.astype(np.float32), ij)]
    info = ctx.malloc(16)
    E.filter_matches_dev(ctx, 200, None, d[0], d[1], d[2], None, info)          # count only
    h = np.empty(4, np.int32); ctx.d2h(h, info)
    _, mask_h, info_h = E.find_fundamental_ransac(p1, p2, 1.0, 0.99, ctx=ctx)
    assert h[0] == mask_h.sum() and h[1] == info_h["iterations"] and h[3] == info_h["sample"]
    with pytest.raises(native.NativeError, match="NULL"):
        E.filter_matches_dev(ctx, 200, None, d[0], d[1], None, None, info)
    with pytest.raises(native.NativeError, match="n_max"):
        E.filter_matches_dev(ctx, 0, None, d[0], d[1], d[2], None, info)
    for p_ in d + [info]:
        ctx.free(p_)
