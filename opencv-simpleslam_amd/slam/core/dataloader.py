"""KITTI-odometry sequence loader for the BASELINE configs (C1-C3: "KITTI-00 ...").

Mirrors the names of the reference's slam/core/dataloader.py (`load_sequence`,
`load_frame_pair`, `load_calibration`, `load_groundtruth`) for `args.dataset == 'kitti'`, with the
things the reference hard-codes made data-driven:

  * the sequence: the reference always opens '05' (dataloader.py:33, :225); here `args.kitti_seq`
    (default '00', the sequence BASELINE.json names)
  * the calibration: the reference returns the constants of sequences 04-12 (fx 707.0912,
    dataloader.py:125-141), which are WRONG for sequence 00 (fx 718.856); here P0 / P1 are read from
    the sequence's own calib.txt
  * ground truth: poses/<seq>.txt (12 numbers per line, 3x4 camera-to-world) and times.txt

Layout (KITTI odometry benchmark):  <base_dir>/kitti/<seq>/image_0/000000.png ...,
<base_dir>/kitti/<seq>/calib.txt, <base_dir>/kitti/<seq>/times.txt, <base_dir>/kitti/poses/<seq>.txt.

Images are read with cv2 when importable, otherwise with Pillow (8-bit grayscale PNG -> H x W x 3
uint8, what cv2.imread returns for these files).  Host-side I/O only: nothing here is on the hot path.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional

import numpy as np


def _seq(args) -> str:
    s = getattr(args, "kitti_seq", None) or getattr(args, "sequence", None) or "00"
    return f"{int(s):02d}" if str(s).isdigit() else str(s)


def _root(args) -> str:
    return os.path.join(args.base_dir, args.dataset)


def load_sequence(args) -> List[str]:
    """Sorted left-camera image paths of the sequence (reference dataloader.py:23-65)."""
    if args.dataset != "kitti":
        raise ValueError(f"Unknown dataset: {args.dataset} (this loader covers the KITTI odometry layout)")
    seq = sorted(glob.glob(os.path.join(_root(args), _seq(args), "image_0", "*.png")))
    if len(seq) < 2:
        raise RuntimeError("Dataset must contain at least two frames.")
    return seq


def load_stereo_paths(args) -> List[str]:
    if args.dataset != "kitti":
        return []
    return sorted(glob.glob(os.path.join(_root(args), _seq(args), "image_1", "*.png")))


def imread(path: str) -> np.ndarray:
    """uint8 H x W x 3 (BGR order; KITTI frames are grayscale, so the three planes are equal)."""
    try:                                     # pragma: no cover - cv2 absent in the build image
        import cv2
        img = cv2.imread(path)
        if img is None:
            raise IOError(path)
        return img
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(a[:, :, ::-1])


def load_frame_pair(args, seq, i):
    """BGR frames i and i+1 (reference dataloader.py:68-75)."""
    return imread(seq[i]), imread(seq[i + 1])


def load_calibration(args) -> Dict[str, np.ndarray]:
    """{'K_l','P_l','K_r','P_r'} from the sequence's calib.txt (P0 = left gray, P1 = right gray)."""
    if args.dataset != "kitti":
        raise ValueError(f"No calibration loader for {args.dataset}")
    path = os.path.join(_root(args), _seq(args), "calib.txt")
    P = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            key, vals = line.split(":", 1)
            v = np.array(vals.split(), np.float64)
            if v.size == 12:
                P[key.strip()] = v.reshape(3, 4)
    if "P0" not in P:
        raise RuntimeError(f"{path}: no P0 entry")
    P_l, P_r = P["P0"], P.get("P1")
    return {"K_l": P_l[:3, :3].copy(), "P_l": P_l, "K_r": None if P_r is None else P_r[:3, :3].copy(), "P_r": P_r}


def load_groundtruth(args) -> Optional[np.ndarray]:
    """[N,3,4] camera-to-world poses of the sequence, or None when the benchmark ships none
    (sequences 11-21) (reference dataloader.py:216-227)."""
    path = os.path.join(_root(args), "poses", _seq(args) + ".txt")
    if not os.path.exists(path):
        return None
    return np.loadtxt(path).reshape(-1, 3, 4)


def load_timestamps(args) -> Optional[np.ndarray]:
    path = os.path.join(_root(args), _seq(args), "times.txt")
    return np.loadtxt(path) if os.path.exists(path) else None


def groundtruth_centres(gt: np.ndarray) -> np.ndarray:
    """[N,3] camera centres of KITTI's camera-to-world rows (for trajectory_eval.ate_rmse)."""
    return np.asarray(gt, np.float64).reshape(-1, 3, 4)[:, :, 3].copy()
