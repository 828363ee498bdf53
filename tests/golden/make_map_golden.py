"""Golden record of the REFERENCE's own map containers (slam/core/landmark_utils.py:47-161) run
through a scripted sequence of mutations, in the build container (cv2 stubbed: the containers use
numpy and scipy only).  The overlay's SoA-backed `Map` must end in the same state.

    python tests/golden/make_map_golden.py        # writes tests/golden/map_ops.npz
"""
import sys
import types
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference")
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
from slam.core.landmark_utils import Map                            # noqa: E402
import map_ops                                                      # noqa: E402


def main(out="tests/golden/map_ops.npz"):
    m = Map()
    map_ops.run(m)
    np.savez_compressed(out, **map_ops.state(m))
    print("wrote", out, len(m), "points")


if __name__ == "__main__":
    main()
