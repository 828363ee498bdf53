"""LightGlue(features='aliked') matcher instance on the HIP backend.

Stands in for the `lightglue.LightGlue` nn.Module the reference builds at
slam/core/features_utils.py:26; `match()` is `matcher({...})` + `rbd` + the
`scores > min_conf` filter of features_utils.py:157-169.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native, weights as W


class LightGlueHIP:
    default_conf = dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1,
                        prune_min_kpts=-1)

    def __init__(self, state_dict=None, max_kpts: int = 4096, ctx=None, max_pairs: int = 1, **conf):
        self.ctx = ctx or _native.default_context()
        self.state_dict = state_dict if state_dict is not None else W.random_lightglue_state_dict(0)
        blob = W.pack_lightglue(self.state_dict)
        h = C.c_void_p()
        _native.check(_native.lib().sslam_lightglue_create_batched(
            self.ctx.handle, _native.ptr(blob), blob.size, int(max_kpts), int(max_pairs), C.byref(h)),
            "sslam_lightglue_create_batched")
        self.handle = h
        self.max_pairs = int(max_pairs)
        kc = C.c_int()
        _native.check(_native.lib().sslam_lightglue_capacity(h, C.byref(kc)))
        self.capacity = int(kc.value)
        self.max_kpts = int(max_kpts)
        self.conf = dict(self.default_conf)
        self.precision = 2               # 'f16x3p1', what sslam_lightglue_create* starts in (set_precision)
        self.epoch = 0                   # bumped by every call that changes what a match returns (memoised results go stale)
        self.set_conf(**conf)

    def set_conf(self, **conf):
        self.epoch += 1
        self.conf.update(conf)
        c = self.conf
        _native.check(_native.lib().sslam_lightglue_set_conf(
            self.handle, float(c["depth_confidence"]), float(c["width_confidence"]),
            float(c["filter_threshold"]), int(c["prune_min_kpts"])), "sslam_lightglue_set_conf")

    def parameters(self):          # the reference probes `next(matcher.parameters()).device`
        return iter(())

    def close(self):
        if getattr(self, "handle", None):
            _native.lib().sslam_lightglue_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match(self, xy0, desc0, xy1, desc1, min_conf: float = 0.7, size0=None, size1=None):
        """Host arrays in, host arrays out: (matches [K,2] int32, scores [K], stop_layer).  size0 / size1: optional
        (W, H) of the images - the 'image_size' entry of upstream feature dicts: keypoints are then normalised by
        it instead of by their bounding box (the reference's legacy pair entry, features_utils.py:233-247)."""
        xy0 = np.ascontiguousarray(xy0, np.float32).reshape(-1, 2)
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
        M, N = len(xy0), len(xy1)
        if M == 0 or N == 0:
            return np.zeros((0, 2), np.int32), np.zeros((0,), np.float32), 0
        desc0 = np.ascontiguousarray(desc0, np.float32).reshape(M, -1)
        desc1 = np.ascontiguousarray(desc1, np.float32).reshape(N, -1)
        if desc0.shape[1] != 128 or desc1.shape[1] != 128:
            raise ValueError("LightGlue(features='aliked') expects 128-d descriptors")
        kmax = max(1, min(M, N))
        ij = np.empty((kmax, 2), np.int32)
        sc = np.empty((kmax,), np.float32)
        k, stop = C.c_int(0), C.c_int(0)
        P = _native.ptr
        s0 = None if size0 is None else np.ascontiguousarray(size0, np.float32).reshape(2)
        s1 = None if size1 is None else np.ascontiguousarray(size1, np.float32).reshape(2)
        _native.check(_native.lib().sslam_lightglue_match_host_sized(
            self.handle, P(xy0), P(desc0), M, P(s0), P(xy1), P(desc1), N, P(s1), float(min_conf), P(ij), P(sc),
            C.byref(k), C.byref(stop)), "sslam_lightglue_match_host")
        return ij[:k.value].copy(), sc[:k.value].copy(), int(stop.value)

    def match_dev(self, xy0, desc0, M, xy1, desc1, N, ij_out, score_out, info_out, min_conf=0.7,
                  m_dev=None, n_dev=None):
        """Device pointers (ints or torch tensors); enqueues only, no sync.  m_dev / n_dev:
        optional device int32 counts (<= M, N) produced by AlikedHIP.extract_dev."""
        P = _native.ptr
        _native.check(_native.lib().sslam_lightglue_match_dev(
            self.handle, P(xy0), P(desc0), int(M), P(xy1), P(desc1), int(N), P(m_dev), P(n_dev),
            float(min_conf), P(ij_out), P(score_out), P(info_out)), "sslam_lightglue_match_dev")

    def match_batch_dev(self, pairs, ij_out, score_out, info_out, out_stride, min_conf=0.7):
        """One enqueue for a list of pairs.  pairs: sequence of (xy0, desc0, M, xy1, desc1, N[, m_dev,
        n_dev]) with device pointers (ints or torch tensors).  Outputs: device buffers holding
        [n_pairs][out_stride][2] int32, [n_pairs][out_stride] float32, [n_pairs][4] int32."""
        n = len(pairs)
        if not 1 <= n <= self.max_pairs:
            raise ValueError(f"{n} pairs, instance capacity is {self.max_pairs}")
        vp = C.c_void_p * n

        def col(i):
            vals = []
            for pr in pairs:
                v = pr[i] if len(pr) > i else None
                vals.append(None if v is None else _native.ptr(v))
            return vp(*vals)
        ia = C.c_int32 * n
        _native.check(_native.lib().sslam_lightglue_match_batch_dev(
            self.handle, n, col(0), col(1), col(6), ia(*[int(pr[2]) for pr in pairs]),
            col(3), col(4), col(7), ia(*[int(pr[5]) for pr in pairs]), float(min_conf),
            _native.ptr(ij_out), _native.ptr(score_out), _native.ptr(info_out), int(out_stride)),
            "sslam_lightglue_match_batch_dev")

    def use_graphs(self, enable: bool = True):
        """Replay `match_dev` / `match_batch_dev` as a cached hipGraph per distinct argument tuple."""
        _native.check(_native.lib().sslam_lightglue_use_graphs(self.handle, int(bool(enable))))

    def range_overflow(self) -> bool:
        """True if, since the last call, a finite activation left the fp16 range of the split-precision
        path (results of those calls are not fp32-grade).  Synchronises; clears the flag."""
        f = C.c_int(0)
        _native.check(_native.lib().sslam_lightglue_range_overflow(self.handle, C.byref(f)))
        return bool(f.value)

    def debug_key_split(self, ks: int):
        """Test hook: the key split of the attention launches.  0 = by batch size (none for batched launches, 2 or 4 key
        ranges for one pair, merged by the fused FFN's tiles) on the hand-scheduled assembly kernel; -5 = the same with the
        merge as a launch of its own (the r04 form); -4 = that policy on the r02 4-wave kernel (merge launch); 1 / 2 / 4 = that many ranges (4-wave kernel); 101 / 102 / 104 = that many (assembly kernel); no split at
        any size: -1 the 4-wave kernel, -3 the assembly kernel - for A/B and bit-identity checks."""
        self.epoch += 1
        _native.check(_native.lib().sslam_lightglue_debug_key_split(self.handle, int(ks)))

    def debug_split_form(self, mask: int):
        """Precision-study hook (profiles/r04_split_study.md): drop cross terms of the split products; 0 = product."""
        self.epoch += 1
        _native.check(_native.lib().sslam_lightglue_debug_split_form(self.handle, int(mask)))

    def debug_big_gemm(self, mode: int):
        """Test hook: -1 linears by batch size, 0 always the 64-row ring kernels (single-pair form), 1 always the
        batched form (128 x 128 projections + the whole FFN as one kernel, its tile by token count), 2 / 3 the batched
        form with 64- / 32-token FFN tiles forced (bit-identical results), 5 the batched form with the token heads as a
        launch of their own (by default the cross block's fused FFN evaluates them on the state it writes)."""
        self.epoch += 1
        _native.check(_native.lib().sslam_lightglue_debug_big_gemm(self.handle, int(mode)))

    def debug_read(self, which: int, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        _native.check(_native.lib().sslam_lightglue_debug_read(self.handle, which, _native.ptr(out), out.nbytes))
        return out

    def set_precision(self, mode: str | int):
        """'f32' / 0: exact-fp32 matrix-core path; 'f16x3' / 1: fp16 hi/lo split path, three MFMAs per product everywhere;
        'f16x3p1' / 2 (DEFAULT since r05): the split path with the softmax weights as ONE fp16 plane in P.V (-12 % attention
        time; the same match indices as 'f16x3' over 131 199 oracle matches and score error 1.06e-4 against its 4.4e-5,
        profiles/r05_flip_soak.md; token states 2.4e-5 from exact instead of 4e-6)."""
        self.epoch += 1
        m = {"f32": 0, "f16x3": 1, "f16x3p1": 2}.get(mode, mode)
        _native.check(_native.lib().sslam_lightglue_set_precision(self.handle, int(m)))
        self.precision = int(m)

    def debug_layers(self, layers: int, self_only: bool = False):
        """Test hook: stop the next matches after `layers` layers (after the self block of the last
        one when `self_only`), so `debug_read(0, ...)` returns that intermediate token state."""
        self.epoch += 1
        _native.check(_native.lib().sslam_lightglue_debug_layers(self.handle, int(layers), int(bool(self_only))))

    def profile(self, enable: bool):
        _native.check(_native.lib().sslam_lightglue_profile(self.handle, int(bool(enable))))

    def profile_read(self):
        """(summed ms, launches) of the HIP-event-bracketed attention launches since the last read."""
        ms, n = C.c_float(), C.c_int()
        _native.check(_native.lib().sslam_lightglue_profile_read(self.handle, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)
