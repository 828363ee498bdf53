"""KITTI-odometry loader on a synthetic directory in the benchmark's layout (no dataset in the image):
sequence selection, per-sequence calibration read from calib.txt (NOT the reference's sequence-05
constants, reference slam/core/dataloader.py:125-141), ground truth, frame decoding, and the ATE tool
on top of it."""
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import load_pkg

# KITTI odometry calib.txt of sequences 00-02 (public benchmark values)
CALIB_00 = """P0: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 0.000000000000e+00 0.000000000000e+00 7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P1: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 -3.861448000000e+02 0.000000000000e+00 7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P2: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 4.538225000000e+01 0.000000000000e+00 7.188560000000e+02 1.852157000000e+02 -1.130887000000e-01 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 3.779761000000e-03
Tr: 4.276802385584e-04 -9.999672484946e-01 -8.084491683471e-03 -1.198459927713e-02 -7.210626507497e-03 8.081198471645e-03 -9.999413164504e-01 -5.403984729748e-02 9.999738645903e-01 4.859485810390e-04 -7.206933692422e-03 -2.921968648686e-01
"""


def _make_tree(root, seq, n, h=24, w=40):
    from PIL import Image
    d = root / "kitti" / seq / "image_0"
    d.mkdir(parents=True)
    rng = np.random.default_rng(int(seq))
    imgs = []
    for i in range(n):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        Image.fromarray(a, mode="L").save(d / f"{i:06d}.png")
        imgs.append(a)
    (root / "kitti" / seq / "calib.txt").write_text(CALIB_00)
    (root / "kitti" / seq / "times.txt").write_text("\n".join(f"{0.1 * i:.6e}" for i in range(n)) + "\n")
    (root / "kitti" / "poses").mkdir(exist_ok=True)
    poses = []
    for i in range(n):
        T = np.eye(4)[:3]
        T[:, 3] = [0.1 * i, 0.0, 0.8 * i]
        poses.append(T.ravel())
    np.savetxt(root / "kitti" / "poses" / f"{seq}.txt", np.array(poses))
    return imgs


def test_kitti_layout(tmp_path):
    dl = load_pkg("slam.core.dataloader")
    imgs = _make_tree(tmp_path, "00", 5)
    _make_tree(tmp_path, "05", 3)
    args = SimpleNamespace(base_dir=str(tmp_path), dataset="kitti")
    seq = dl.load_sequence(args)                              # default: sequence 00 (BASELINE configs)
    assert len(seq) == 5 and seq[0].endswith("000000.png") and seq == sorted(seq)
    a, b = dl.load_frame_pair(args, seq, 1)
    assert a.dtype == np.uint8 and a.shape == (24, 40, 3)
    np.testing.assert_array_equal(a[:, :, 0], imgs[1]); np.testing.assert_array_equal(b[:, :, 2], imgs[2])
    cal = dl.load_calibration(args)
    assert cal["K_l"].shape == (3, 3) and cal["K_l"][0, 0] == 718.856 and cal["K_l"][0, 2] == 607.1928
    assert cal["K_l"][0, 0] != 707.0912                       # not the reference's sequence-05 constant
    assert cal["P_r"][0, 3] == -386.1448
    gt = dl.load_groundtruth(args)
    assert gt.shape == (5, 3, 4) and gt[3, 2, 3] == pytest.approx(2.4)
    assert dl.load_timestamps(args).shape == (5,)
    args5 = SimpleNamespace(base_dir=str(tmp_path), dataset="kitti", kitti_seq=5)
    assert len(dl.load_sequence(args5)) == 3
    args11 = SimpleNamespace(base_dir=str(tmp_path), dataset="kitti", kitti_seq="11")
    assert dl.load_groundtruth(args11) is None                # test sequences ship no poses
    with pytest.raises(RuntimeError):
        dl.load_sequence(args11)
    with pytest.raises(ValueError):
        dl.load_sequence(SimpleNamespace(base_dir=str(tmp_path), dataset="parking"))


def test_ate_on_loaded_groundtruth(tmp_path):
    """Ground truth -> centres -> ATE-RMSE of a scaled, rotated, noisy copy (monocular: Sim(3))."""
    dl = load_pkg("slam.core.dataloader"); T = load_pkg("slam.core.trajectory_eval")
    _make_tree(tmp_path, "00", 40)
    gt = dl.groundtruth_centres(dl.load_groundtruth(SimpleNamespace(base_dir=str(tmp_path), dataset="kitti")))
    rng = np.random.default_rng(0)
    ang = 0.3
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    est = (gt @ R.T) / 7.0 + [3.0, -1.0, 2.0] + rng.normal(0, 0.002, gt.shape)
    assert T.ate_rmse(gt, est) < 0.05                         # noise 2 mm x scale 7
    assert T.ate_rmse(gt, est + rng.normal(0, 0.1, gt.shape)) > 0.3
