"""Generate tests/golden/reproject_match.npz from the REFERENCE's own
`slam.core.pnp_utils.reproject_and_match_2d3d` (pnp_utils.py:224-304).

Run in the build container only (needs /root/reference on disk):
    python tests/golden/make_reproject_golden.py
`pnp_utils` imports cv2 at module scope; an empty stub stands in (the float-descriptor path of the
function under test touches numpy and scipy.spatial.cKDTree only).  Map points are plain namespaces
with `.position` / `.observations`, which is all the function reads.  The scenes come from the
seeded generator tests/reproject_scenes.py; the .npz holds the reference's outputs plus a digest
of the generated inputs (data only), so the tests notice if the generator ever drifts.
"""
import sys
import types
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
from slam.core.pnp_utils import reproject_and_match_2d3d          # noqa: E402
import reproject_scenes as RS                                     # noqa: E402


def main(out="tests/golden/reproject_match.npz"):
    blob = {"n_cases": len(RS.CASES)}
    for c, args in enumerate(RS.CASES):
        sc = RS.make_case(*args)
        m = reproject_and_match_2d3d(sc["wmap"], sc["K"], sc["Tcw"], sc["kp"], sc["des"], sc["W"], sc["H"],
                                     radius_px=sc["radius"], max_l2=sc["max_l2"], use_cosine=sc["use_cosine"])
        print(f"case {c}: {len(m.kp_indices)} matches of {len(sc['wmap'].points)} points / {len(sc['kp'])} keypoints")
        blob[f"digest{c}"] = sc["digest"]
        blob[f"kp{c}"] = np.asarray(m.kp_indices, np.int64)
        blob[f"mp{c}"] = np.asarray(m.mp_ids, np.int64)
        blob[f"pts3d{c}"] = m.pts3d
        blob[f"pts2d{c}"] = m.pts2d
    np.savez_compressed(out, **blob)


if __name__ == "__main__":
    main()
