"""bench.py's `dropin.cv2_classes` leg: the `value` and `slam_loop` loops of the drop-in leg in an interpreter where `cv2` is
importable (tests/cv2_stub.py with the C value classes of tests/cv2like/cv2like.c), so that the overlay takes the branch it
takes wherever the reference really runs.  One JSON object on the last line."""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
DUCK = "--duck-types" in sys.argv                  # the same two loops in the same kind of process WITHOUT cv2: the like-for-like partner
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
if not DUCK:
    import cv2_stub                                # noqa: E402
    cv2 = cv2_stub.install(native_classes=True)    # BEFORE the product binds cv2 at import
import bench                                       # noqa: E402
T = importlib.import_module("opencv-simpleslam_amd.slam.core.types")
if DUCK:
    assert not T.HAVE_CV2
else:
    assert T.HAVE_CV2 and T.KeyPoint is cv2.KeyPoint and T.DMatch is cv2.DMatch
out = bench.dropin_leg(int(argv[0]) if argv else 96, only_matched_loops=True)
out["classes"] = ("the overlay's duck types (no cv2 importable)" if DUCK else
                  f"{cv2.KeyPoint.__name__} / {cv2.DMatch.__name__} of tests/cv2like/cv2like.c (cv2 stand-in: the wheel is absent from the image)")
print(json.dumps(out))
