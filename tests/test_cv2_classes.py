"""The LightGlue drop-in with cv2's OWN KeyPoint / DMatch classes present - the only environment
slam/monocular/main_revamped.py runs in (reference slam/core/features_utils.py:2 imports cv2; :61-63 builds
`cv2.KeyPoint(x, y, 1)` per keypoint, :80-83 `cv2.DMatch(i, j, 0, 0.0)` per match, :185-200 reads them back).  The wheel is
absent from the image, so `cv2` is tests/cv2_stub.py with the value classes of tests/cv2like/cv2like.c (C structs behind
python objects, eager construction, `pt` a fresh tuple per read, writable fields, KeyPoint_convert in one C pass).  The
product binds cv2 at import, so every scenario runs in a child interpreter (SSLAM_TEST_CV2_CLASSES=1: tests/conftest.py
installs the stand-in first):

  * CPU: the bulk converters and the whole-list "still what I was built from" checks of slam/core/types.py, and the ring's
    host logic (tests/test_feature_ring_host_logic.py) once more with these classes;
  * GPU: tests/test_dropin_names_gpu.py once more - conventions, device-resident matcher, look-ahead, the filter behind the
    match, the keyframe pattern - in ONE child pytest process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ENV = dict(os.environ, SSLAM_TEST_CV2_CLASSES="1")

CHILD_TYPES = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import cv2_stub
cv2 = cv2_stub.install(native_classes=True)
T = importlib.import_module("opencv-simpleslam_amd.slam.core.types")
assert T.HAVE_CV2 and T.KeyPoint is cv2.KeyPoint and T.DMatch is cv2.DMatch and T.keypoint_shells is None
rng = np.random.default_rng(0)
xy = rng.uniform(0, 1200, (2048, 2)).astype(np.float32)

# keypoints: what the reference builds one by one (features_utils.py:62 cv2.KeyPoint(x, y, 1)), field for field
kps = T.keypoints_from_xy(xy)
ref = [cv2.KeyPoint(float(x), float(y), 1) for x, y in xy]
fields = lambda k: (k.pt, k.size, k.angle, k.response, k.octave, k.class_id)
assert isinstance(kps, list) and [fields(k) for k in kps] == [fields(k) for k in ref]
assert kps[0].response == 0.0                                # (KeyPoint_convert's own default is 1)
assert T.keypoints_from_xy(xy[:0]) == []
np.testing.assert_array_equal(T.xy_from_keypoints(kps), xy)
np.testing.assert_array_equal(T.xy_from_keypoints(tuple(kps[:7])), xy[:7])
class Other:                                                 # not a cv2.KeyPoint: the generic attribute pass
    def __init__(self, pt): self.pt = pt
mixed = kps[:3] + [Other((1.5, 2.5))]
np.testing.assert_array_equal(T.xy_from_keypoints(mixed), np.vstack([xy[:3], [[1.5, 2.5]]]).astype(np.float32))

# KeyPointList: EVERY element is read back, not a sample
L = T.KeyPointList(kps, xy)
assert L.pristine_xy() is xy
for i in (0, 1, 777, 2047):                                  # (777: never on the old 8-element grid)
    L = T.KeyPointList(T.keypoints_from_xy(xy), xy)
    x, y = L[i].pt
    L[i].pt = (x + 0.5, y)
    assert L.pristine_xy() is None, i
    assert L.pristine_xy() is None                           # (and it stays an ordinary list)
L = T.KeyPointList(T.keypoints_from_xy(xy), xy)
list.__setitem__(L, 5, Other((0.0, 0.0)))                    # behind the list's back: an element that is not cv2's class
assert L.pristine_xy() is None
L = T.KeyPointList(T.keypoints_from_xy(xy), xy); L.sort(key=lambda k: k.pt)
assert L.pristine_xy() is None
assert T.KeyPointList([], xy[:0]).pristine_xy() is not None

# matches: cv2.DMatch(i, j, 0, 0.0) (features_utils.py:82), prepared behind the running match and bound afterwards
ij = np.stack([np.sort(rng.choice(2048, 600, replace=False)), rng.integers(0, 2048, 600)], 1).astype(np.int32)
mf = lambda m: (m.queryIdx, m.trainIdx, m.imgIdx, m.distance)
ref = [cv2.DMatch(int(i), int(j), 0, 0.0) for i, j in ij]
assert [mf(m) for m in T.matches_from_ij(ij)] == [mf(m) for m in ref] and T.matches_from_ij(ij[:0]) == []
for prepared in (0, 100, 600, 900):
    shells, src = T.match_shells(prepared)
    assert len(shells) == prepared and all(type(m) is cv2.DMatch for m in shells)
    out = T.bind_matches(shells, src, ij)
    assert out is shells and [mf(m) for m in out] == [mf(m) for m in ref], prepared
assert T.bind_matches(T.match_shells(5)[0], None, ij[:0]) == []
assert len({id(m) for m in T.match_shells(50)[0]}) == 50     # distinct objects
M = T.MatchList(T.matches_from_ij(ij), ij)
assert M.pristine_ij() is ij
for i, field in ((0, "queryIdx"), (1, "trainIdx"), (333, "trainIdx"), (599, "queryIdx")):
    M = T.MatchList(T.matches_from_ij(ij), ij)
    setattr(M[i], field, getattr(M[i], field) + 1)
    assert M.pristine_ij() is None, (i, field)
M = T.MatchList(T.matches_from_ij(ij), ij); M[17].distance = 3.0       # not an index: the pairs are still the list's
assert M.pristine_ij() is ij
M = T.MatchList(T.matches_from_ij(ij), ij); M.reverse()
assert M.pristine_ij() is None
assert T.MatchList([], ij[:0]).pristine_ij() is not None
assert T.keypoint_edit_epoch() is None and T.dmatch_edit_epoch() is None
print("CV2 CLASSES OK")
'''


def test_bulk_converters_and_whole_list_checks_with_cv2_classes():
    out = subprocess.run([sys.executable, "-c", CHILD_TYPES % {"root": str(ROOT)}], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CV2 CLASSES OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


CHILD_NO_CONVERT = r'''
import importlib, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import cv2_stub
cv2 = cv2_stub.install(native_classes=False)                 # python classes, and NO KeyPoint_convert in the module
assert not hasattr(cv2, "KeyPoint_convert")
T = importlib.import_module("opencv-simpleslam_amd.slam.core.types")
assert T.HAVE_CV2 and T._kp_convert is None
xy = np.random.default_rng(1).uniform(0, 900, (300, 2)).astype(np.float32)
kps = T.keypoints_from_xy(xy)
assert [k.pt for k in kps] == [(float(x), float(y)) for x, y in xy] and kps[0].size == 1 and kps[0].response == 0.0
np.testing.assert_array_equal(T.xy_from_keypoints(kps), xy)
L = T.KeyPointList(kps, xy)
assert L.pristine_xy() is xy
L[123].pt = (1.0, 2.0)
assert L.pristine_xy() is None
ij = np.stack([np.arange(40), np.arange(40)[::-1]], 1).astype(np.int32)
out = T.bind_matches(T.match_shells(25)[0], None, ij)
assert [(m.queryIdx, m.trainIdx, m.imgIdx, m.distance) for m in out] == [(int(i), int(j), 0, 0.0) for i, j in ij]
M = T.MatchList(out, ij)
assert M.pristine_ij() is ij
M[39].trainIdx = 7
assert M.pristine_ij() is None
print("NO CONVERT OK")
'''


def test_a_cv2_without_keypoint_convert_takes_the_element_wise_paths():
    """A `cv2` module that lacks `KeyPoint_convert` (or whose classes are not the wheel's): the same results through the
    per-element constructors and the generic `.pt` pass."""
    out = subprocess.run([sys.executable, "-c", CHILD_NO_CONVERT % {"root": str(ROOT)}], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "NO CONVERT OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


def _child_pytest(args, timeout):
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *args], cwd=str(ROOT), env=ENV,
                         capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / "cv2_classes_child_fail.log").write_text(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    return out.stdout


def test_ring_host_logic_with_cv2_classes():
    """tests/test_feature_ring_host_logic.py (the ring on a stand-in for the native layer) with cv2's classes: the keyframe
    pattern, the memo, the look-ahead, the filter behind the match - the branch of feature_ring.py / types.py that runs
    wherever cv2 is installed."""
    out = _child_pytest(["tests/test_feature_ring_host_logic.py", "-m", "not gpu"], 600)
    assert " passed" in out and "skipped" not in out.splitlines()[-1], out[-500:]


@pytest.mark.gpu
def test_dropin_names_on_the_gpu_with_cv2_classes():
    """tests/test_dropin_names_gpu.py once more with `HAVE_CV2` true - every case, one child process."""
    out = _child_pytest(["tests/test_dropin_names_gpu.py", "tests/test_end_to_end_gpu.py", "-m", "gpu"], 1500)
    last = out.strip().splitlines()[-1]
    assert " passed" in last and "failed" not in last, out[-800:]
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "cv2_classes_child.log").write_text(out)
