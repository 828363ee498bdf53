"""A CPU stand-in for the native layer UNDER the drop-in names, for tests of the host logic only (feature_ring.py:
residency, memo, look-ahead scheduling; features_utils.py: argument handling) where there is no GPU.

"Device memory" is a process-wide heap of numpy buffers found by address; streams execute at enqueue time, so every
`sync()` is a no-op and ordering bugs are NOT what this catches - the -m gpu tests run the same scenarios on the real
library.  The fake extractor / matcher / filter are deterministic functions of their inputs, the same function on the
"device" and on the "host" entry, so the tests can demand that every path returns the same thing.  Nothing here is the
product's arithmetic and nothing in the product imports it."""
import numpy as np

BASE = 0x10000


class _Heap:
    """The "device memory" of a process: every FakeContext of the process allocates from it (as every stream of a GPU sees
    the same memory), allocations are separate numpy buffers found by address."""

    def __init__(self):
        self.blocks = []                 # sorted (start address, uint8 buffer)
        self.top = BASE

    def malloc(self, nbytes):
        n = (int(nbytes) + 255) // 256 * 256
        p = self.top
        self.blocks.append((p, np.zeros(n, np.uint8)))
        self.top += n + 256              # (a guard gap: running off the end of a block is an error, not a neighbour's data)
        return p

    def view(self, dptr, nbytes):
        import bisect
        dptr = int(dptr)
        i = bisect.bisect_right(self.blocks, dptr, key=lambda b: b[0]) - 1
        assert i >= 0, "fake device: address below every allocation"
        start, buf = self.blocks[i]
        o = dptr - start
        assert 0 <= o and o + nbytes <= len(buf), "fake device: access outside an allocation"
        return buf[o:o + nbytes]


_HEAP = _Heap()


class FakeContext:
    def __init__(self, device=0, stream=None):
        self.heap = _HEAP
        self.device = int(device)
        self.syncs = 0
        self.stream = 0

    def view(self, dptr, nbytes):
        return self.heap.view(dptr, nbytes)

    def malloc(self, nbytes):
        return self.heap.malloc(nbytes)

    def free(self, dptr): pass
    def sync(self): self.syncs += 1
    def event(self): return 1
    def timing_event(self): return 1
    def record(self, ev): pass
    def wait(self, ev): pass

    def host_alloc(self, nbytes):
        return np.zeros(int(nbytes), np.uint8)

    def h2d(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        self.view(dptr, a.nbytes)[:] = a.view(np.uint8).reshape(-1)

    h2d_async = h2d

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.malloc(max(a.nbytes, 1))
        if a.nbytes:
            self.h2d(p, a)
        return p

    def d2h_async(self, arr, dptr, nbytes=None):
        n = arr.nbytes if nbytes is None else int(nbytes)
        arr.view(np.uint8).reshape(-1)[:n] = self.view(dptr, n)

    d2h = d2h_async

    def d2d_async(self, dst, src, nbytes):
        self.view(dst, nbytes)[:] = self.view(src, nbytes).copy()

    def memset_async(self, dst, value, nbytes):
        self.view(dst, nbytes)[:] = value

    def f32(self, dptr, n): return self.view(dptr, 4 * n).view(np.float32)
    def i32(self, dptr, n): return self.view(dptr, 4 * n).view(np.int32)


class FakeAliked:
    """Serves the frames of a synthetic chain (tests/lg_inputs.py::make_chain) so that frames match: in extraction order, or -
    `by_image=True` - the chain frame whose index the image carries in its first two bytes (the frame-sharded pipeline
    extracts frames in any order on any rank).  The image only has to be a valid uint8 array in allocated memory."""

    def __init__(self, ctx, chain, max_num_keypoints=256, by_image=False, max_frames=1):
        self.ctx, self.max_num_keypoints, self.chain = ctx, int(max_num_keypoints), chain
        self.by_image, self.max_frames = by_image, int(max_frames)
        self.calls = 0

    def use_graphs(self, enable=True): pass
    def close(self): pass
    def range_overflow(self): return False

    def extract_dev(self, img_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        img = self.ctx.view(img_dev, H * Wd * Cn)      # (the image must have been uploaded into allocated memory)
        idx = int(img[0]) + 256 * int(img[1]) if self.by_image else self.calls
        xy, d = self.chain[idx % len(self.chain)]
        self.calls += 1
        n = len(xy)
        assert n <= int(max_kpts or self.max_num_keypoints)
        self.ctx.f32(xy_out, 2 * n)[:] = xy.reshape(-1)
        self.ctx.f32(desc_out, 128 * n)[:] = d.reshape(-1)
        self.ctx.i32(n_out, 1)[:] = n

    def extract_batch_dev(self, imgs_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        assert 1 <= len(imgs_dev) <= self.max_frames
        for f in range(len(imgs_dev)):
            self.extract_dev(imgs_dev[f], H, Wd, Cn, xy_out[f], desc_out[f], None if score_out is None else score_out[f],
                             n_out[f], max_kpts=max_kpts)


def fake_match(xy0, d0, xy1, d1, min_conf):
    """Mutual nearest neighbours by cosine with a score = the cosine; + a weak dependence on the keypoints, so that
    edited keypoints change the answer.  Ascending query index like LightGlue."""
    s = d0 @ d1.T - 1e-3 * np.abs(xy0[:, None, 0] - xy1[None, :, 0]) / 300.0
    j = s.argmax(1)
    i_back = s.argmax(0)
    q = np.arange(len(d0))
    ok = (i_back[j] == q) & (s[q, j] > 0.5 * min_conf)
    ij = np.column_stack([q[ok], j[ok]]).astype(np.int32)
    return ij, s[q, j][ok].astype(np.float32)


def fake_inliers(p1, p2, thresh):
    """-> (mask, none): 'inlier' = displacement within 40 x thresh px of the median displacement; no model below 7."""
    d = p2 - p1
    mask = (np.abs(d - np.median(d, axis=0)) < 40.0 * thresh).all(1)
    return mask.astype(np.uint8), int(mask.sum()) < 7


class FakeLightGlue:
    def __init__(self, ctx, max_kpts=256, max_pairs=1):
        self.ctx, self.max_kpts, self.max_pairs = ctx, int(max_kpts), int(max_pairs)
        self.epoch = 0
        self.dev_calls, self.dev_pairs, self.host_calls = 0, 0, 0

    def close(self): pass
    def use_graphs(self, enable=True): pass
    def range_overflow(self): return False
    def parameters(self): return iter(())

    def match(self, xy0, desc0, xy1, desc1, min_conf=0.7, size0=None, size1=None):
        self.host_calls += 1
        ij, sc = fake_match(np.asarray(xy0, np.float32), np.asarray(desc0, np.float32), np.asarray(xy1, np.float32),
                            np.asarray(desc1, np.float32), min_conf)
        return ij, sc, 9

    def _one(self, xy0, desc0, M, xy1, desc1, N, m_dev, n_dev, min_conf, ij_out, score_out, info_out):
        c = self.ctx
        m = min(int(M), int(c.i32(m_dev, 1)[0])) if m_dev else int(M)
        n = min(int(N), int(c.i32(n_dev, 1)[0])) if n_dev else int(N)
        ij, sc = fake_match(c.f32(xy0, 2 * m).reshape(m, 2), c.f32(desc0, 128 * m).reshape(m, 128),
                            c.f32(xy1, 2 * n).reshape(n, 2), c.f32(desc1, 128 * n).reshape(n, 128), min_conf)
        k = len(ij)
        c.i32(ij_out, 2 * k)[:] = ij.reshape(-1)
        c.f32(score_out, k)[:] = sc
        c.i32(info_out, 4)[:] = (k, 9, m, n)
        self.dev_pairs += 1

    def match_dev(self, xy0, desc0, M, xy1, desc1, N, ij_out, score_out, info_out, min_conf=0.7, m_dev=None, n_dev=None):
        self.dev_calls += 1
        self._one(xy0, desc0, M, xy1, desc1, N, m_dev, n_dev, min_conf, ij_out, score_out, info_out)

    def match_batch_dev(self, pairs, ij_out, score_out, info_out, out_stride, min_conf=0.7):
        assert 1 <= len(pairs) <= self.max_pairs
        self.dev_calls += 1
        for p, pr in enumerate(pairs):
            self._one(pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6] if len(pr) > 6 else None,
                      pr[7] if len(pr) > 7 else None, min_conf, ij_out + 8 * out_stride * p, score_out + 4 * out_stride * p,
                      info_out + 16 * p)


class FakeEpipolar:
    """Stands in for opencv-simpleslam_amd/epipolar.py (both entries, one rule)."""

    def __init__(self):
        self.host_calls, self.dev_calls = 0, 0

    def find_fundamental_ransac(self, pts1, pts2, thresh=1.0, confidence=0.99, max_iters=1000, ctx=None):
        self.host_calls += 1
        mask, none = fake_inliers(np.float32(pts1), np.float32(pts2), thresh)
        if none:
            return None, None, {}
        return np.eye(3), mask.astype(bool), {}

    def filter_matches_dev(self, ctx, n_max, n_dev, xy1_dev, xy2_dev, ij_dev, ij_out_dev, info_out_dev, thresh=1.0,
                           confidence=0.99, max_iters=1000, mask_out_dev=None, F_out_dev=None):
        self.dev_calls += 1
        k = max(0, min(int(n_max), int(ctx.i32(n_dev, 1)[0])))
        info = ctx.i32(info_out_dev, 4)
        if k < 8:
            info[:] = (k, 0, 0, -2)
            ctx.view(mask_out_dev, k)[:] = 1
            return
        ij = ctx.i32(ij_dev, 2 * k).reshape(k, 2)
        hi0, hi1 = int(ij[:, 0].max()) + 1, int(ij[:, 1].max()) + 1
        p1 = ctx.f32(xy1_dev, 2 * hi0).reshape(hi0, 2)[ij[:, 0]]
        p2 = ctx.f32(xy2_dev, 2 * hi1).reshape(hi1, 2)[ij[:, 1]]
        mask, none = fake_inliers(p1, p2, thresh)
        ctx.view(mask_out_dev, k)[:] = mask            # (the raw mask even when there is no model: the caller must read info[3])
        info[:] = (0 if none else int(mask.sum()), 1, 0, -1 if none else 0)
