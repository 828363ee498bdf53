"""MI355X (gfx950) backend for the ALIKED + LightGlue + local-BA hot path of
KlrShaK/opencv-SimpleSLAM.

Layout
  csrc/            hand-written HIP kernels + the C-ABI (include/sslam_hip.h)
  _native.py       ctypes binding (no CPU fallback)
  slam/core/       host-side mirror of the reference's hot-path modules, same
                   names and signatures (features_utils, ba_utils, pose_utils)
  weights.py       upstream state-dict <-> flat device blob packing
  frame_shard.py   frame-sharded multi-GPU driver (RCCL all-gather collation)

The directory name contains a hyphen, so import it with
`importlib.import_module("opencv-simpleslam_amd")` or put this directory on
`sys.path` and `import slam.core.features_utils` (drop-in for the reference).
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
