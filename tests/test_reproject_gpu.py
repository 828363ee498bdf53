"""GPU parity of `sslam_reproject_match_host` / the drop-in `reproject_and_match_2d3d` with the
reference's own outputs (tests/golden/reproject_match.npz) and with the oracle on further scenes:
index arrays bit-exact."""
import numpy as np
import pytest

import reproject_scenes as RS
from conftest import ROOT, load_pkg
from oracle import reproject_ref as R

pytestmark = pytest.mark.gpu
G = np.load(ROOT / "tests" / "golden" / "reproject_match.npz")


@pytest.fixture(scope="module")
def P():
    return load_pkg("slam.core.pnp_utils")


@pytest.mark.parametrize("c", range(len(RS.CASES)))
def test_matches_reference_outputs(P, c):
    sc = RS.make_case(*RS.CASES[c])
    assert sc["digest"] == float(G[f"digest{c}"])
    m = P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"],
                                   radius_px=sc["radius"], max_l2=sc["max_l2"], use_cosine=sc["use_cosine"])
    np.testing.assert_array_equal(m.kp_indices, G[f"kp{c}"])
    np.testing.assert_array_equal(m.mp_ids, G[f"mp{c}"])
    np.testing.assert_array_equal(m.pts3d, G[f"pts3d{c}"])
    np.testing.assert_array_equal(m.pts2d, G[f"pts2d{c}"])


def test_c2_size_against_oracle_and_edge_cases(P, native):
    sc = RS.make_case(11, 5000, 2048, 12.0, 0.8, False)           # a C2-sized map / frame
    m = P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"])
    p3, p2, kp, mp = R.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"])
    assert len(kp) > 500
    np.testing.assert_array_equal(m.kp_indices, kp)
    np.testing.assert_array_equal(m.mp_ids, mp)
    # keypoint objects with .pt are accepted like arrays (pnp_utils.py:62-75)
    types = load_pkg("slam.core.types")
    kps = [types.KeyPoint(float(x), float(y), 1.0) for x, y in sc["kp"]]
    m2 = P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], kps, sc["des"], sc["W"], sc["H"])
    assert m2.kp_indices == m.kp_indices
    # empty inputs return the empty record, like the reference (:238-247)
    e = P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"][:0], sc["des"][:0], sc["W"], sc["H"])
    assert e.pts3d.shape == (0, 3) and e.kp_indices == []
    # a camera looking away sees nothing
    T = sc["Tcw"].copy(); T[:3, :3] = np.diag([1.0, 1.0, -1.0]) @ T[:3, :3]; T[2, 3] = -500.0
    e = P.reproject_and_match_2d3d(sc["wmap"], sc["K"], T, sc["kp"], sc["des"], sc["W"], sc["H"])
    assert e.kp_indices == []
    with pytest.raises(NotImplementedError):
        P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], np.zeros((len(sc["kp"]), 32), np.uint8),
                                   sc["W"], sc["H"])
    # a huge radius overflows the per-point candidate list: reported, not truncated silently
    with pytest.raises(native.NativeError, match="keypoints within"):
        P.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"], radius_px=400.0)
