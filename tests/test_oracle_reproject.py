"""oracle/reproject_ref.py against the outputs of the REFERENCE's own reproject_and_match_2d3d
(tests/golden/reproject_match.npz, scenes from tests/reproject_scenes.py)."""
import numpy as np
import pytest

import reproject_scenes as RS
from conftest import ROOT
from oracle import reproject_ref as R

G = np.load(ROOT / "tests" / "golden" / "reproject_match.npz")


@pytest.mark.parametrize("c", range(len(RS.CASES)))
def test_oracle_matches_reference_outputs(c):
    sc = RS.make_case(*RS.CASES[c])
    assert sc["digest"] == float(G[f"digest{c}"]), "scene generator drifted from the one that made the golden file"
    p3, p2, kp, mp = R.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"],
                                                sc["radius"], sc["max_l2"], sc["use_cosine"])
    np.testing.assert_array_equal(kp, G[f"kp{c}"])
    np.testing.assert_array_equal(mp, G[f"mp{c}"])
    np.testing.assert_array_equal(p3, G[f"pts3d{c}"])
    np.testing.assert_array_equal(p2, G[f"pts2d{c}"])
    assert len(set(kp)) == len(kp)                       # a keypoint is used once


def test_empty_inputs():
    sc = RS.make_case(*RS.CASES[4])
    for kp, des in ((sc["kp"][:0], sc["des"][:0]), (sc["kp"], None)):
        p3, p2, k, m = R.reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], kp, des, sc["W"], sc["H"])
        assert p3.shape == (0, 3) and p2.shape == (0, 2) and k == [] and m == []
