"""ALIKED-n16 extractor instance on the HIP backend.

Stands in for the `lightglue.ALIKED` nn.Module the reference builds at
slam/core/features_utils.py:25; `extract()` covers `_bgr_to_tensor` +
`detector.extract` + `rbd` + descriptor re-normalisation (:92-100, :219-222).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native, weights as W


class AlikedHIP:
    def __init__(self, state_dict=None, max_num_keypoints: int = 4000, max_h: int = 1200, max_w: int = 2048,
                 ctx=None, max_frames: int = 1):
        """max_frames > 1: the instance also takes batches of that many frames (`extract_batch_dev`); one workspace
        block per frame."""
        self.ctx = ctx or _native.default_context()
        self.state_dict = state_dict if state_dict is not None else W.random_aliked_state_dict(0)
        blob = W.pack_aliked(self.state_dict)
        h = C.c_void_p()
        _native.check(_native.lib().sslam_aliked_create_batched(
            self.ctx.handle, _native.ptr(blob), blob.size, int(max_h), int(max_w), int(max_num_keypoints),
            int(max_frames), C.byref(h)), "sslam_aliked_create")
        self.handle = h
        self.max_frames = int(max_frames)
        self.max_num_keypoints = int(max_num_keypoints)
        self.max_h, self.max_w = int(max_h), int(max_w)

    def close(self):
        if getattr(self, "handle", None):
            _native.lib().sslam_aliked_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, img: np.ndarray, max_kpts: int | None = None, return_scores: bool = False):
        """uint8 HxWx3 BGR (or HxW gray, HxWx4 BGRA) -> (xy [N,2] float32 in input pixels,
        descriptors [N,128] float32 unit rows)."""
        img = np.ascontiguousarray(img)
        if img.dtype != np.uint8:
            raise TypeError("feature extraction expects a uint8 image (cv2.imread output)")
        if img.ndim == 2:
            H, Wd, Cn = img.shape[0], img.shape[1], 1
        elif img.ndim == 3:
            H, Wd, Cn = img.shape
        else:
            raise ValueError(f"unsupported image shape {img.shape}")
        n_lim = int(max_kpts or self.max_num_keypoints)
        xy = np.empty((n_lim, 2), np.float32)
        desc = np.empty((n_lim, 128), np.float32)
        sc = np.empty((n_lim,), np.float32)
        n = C.c_int(0)
        P = _native.ptr
        _native.check(_native.lib().sslam_aliked_extract_host(
            self.handle, P(img), H, Wd, Cn, n_lim, P(xy), P(desc), P(sc), C.byref(n)),
            "sslam_aliked_extract_host")
        k = n.value
        if return_scores:
            return xy[:k].copy(), desc[:k].copy(), sc[:k].copy()
        return xy[:k].copy(), desc[:k].copy()

    def extract_dev(self, img_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        P = _native.ptr
        _native.check(_native.lib().sslam_aliked_extract_dev(
            self.handle, P(img_dev), int(H), int(Wd), int(Cn), int(max_kpts or self.max_num_keypoints),
            P(xy_out), P(desc_out), P(score_out), P(n_out)), "sslam_aliked_extract_dev")

    def extract_batch_dev(self, imgs_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        """F frames of one size through one launch sequence: every argument but the sizes is a sequence of F device
        pointers (score_out may be None).  Same results as F `extract_dev` calls, bit for bit."""
        F = len(imgs_dev)
        arr = lambda seq: (C.c_void_p * F)(*[int(_native.ptr(p).value or 0) if p is not None else 0 for p in seq])
        a_img, a_xy, a_desc, a_n = arr(imgs_dev), arr(xy_out), arr(desc_out), arr(n_out)
        a_sc = arr(score_out) if score_out is not None else None
        _native.check(_native.lib().sslam_aliked_extract_batch_dev(
            self.handle, F, a_img, int(H), int(Wd), int(Cn), int(max_kpts or self.max_num_keypoints),
            a_xy, a_desc, a_sc, a_n), "sslam_aliked_extract_batch_dev")

    def use_graphs(self, enable: bool = True):
        """Replay `extract_dev` as a cached hipGraph per distinct argument tuple (same results)."""
        _native.check(_native.lib().sslam_aliked_use_graphs(self.handle, int(bool(enable))))

    def range_overflow(self) -> bool:
        """True if, since the last call, a finite activation left the fp16 range of the split-precision stages in some frame
        (that frame's count read -1, its features are void).  Synchronises; clears the flag."""
        f = C.c_int(0)
        _native.check(_native.lib().sslam_aliked_range_overflow(self.handle, C.byref(f)), "sslam_aliked_range_overflow")
        return bool(f.value)

    def debug_read(self, which: int, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        _native.check(_native.lib().sslam_aliked_debug_read(self.handle, which, _native.ptr(out), out.nbytes))
        return out
