"""Seeded synthetic matcher inputs: keypoints uniform in
[2, W-3] x [2, H-3] of a 1241x376 frame, descriptors = row-normalised normal
draws; image 1 = shuffled, jittered, partly replaced copy of image 0 so that a
non-trivial set of true correspondences exists."""
import numpy as np

W_IMG, H_IMG = 1241, 376


def make_pair(m, n=None, seed=0, noise=0.05, drop=0.2):
    n = m if n is None else n
    rng = np.random.default_rng(seed)
    k0 = np.column_stack([rng.uniform(2, W_IMG - 3, m), rng.uniform(2, H_IMG - 3, m)]).astype(np.float32)
    d0 = rng.standard_normal((m, 128)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    src = rng.permutation(m)
    src = np.resize(src, n)
    k1 = (k0[src] + np.float32([3.0, 1.0]) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    d1 = d0[src] + noise * rng.standard_normal((n, 128)).astype(np.float32)
    nd = int(drop * n)
    d1[:nd] = rng.standard_normal((nd, 128))
    k1[:nd] = np.column_stack([rng.uniform(2, W_IMG - 3, nd), rng.uniform(2, H_IMG - 3, nd)])
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    return k0, d0, k1.astype(np.float32), d1.astype(np.float32)


def make_chain(n_frames, m, seed=0, noise=0.05, drop=0.2, period=8):
    """A frame sequence for the reference's frame-loop call pattern (prev -> cur on every frame, keyframe -> cur a few
    frames apart): every frame is a shuffled, jittered, partly replaced view of ONE base set of keypoints, translated by
    (3, 1) px per step of a `period`-frame cycle - so any two frames of the chain share true correspondences and are related
    by a pure translation (one F-matrix model fits them all).  -> [(xy [m,2] f32, desc [m,128] f32 unit rows)] * n_frames"""
    rng = np.random.default_rng(seed)
    base_xy = np.column_stack([rng.uniform(2, W_IMG - 3 - 3 * period, m), rng.uniform(2, H_IMG - 3 - period, m)])
    base_d = rng.standard_normal((m, 128))
    base_d /= np.linalg.norm(base_d, axis=1, keepdims=True)
    out = []
    nd = int(drop * m)
    for f in range(n_frames):
        r = np.random.default_rng(1000 * (seed + 1) + f)
        src = r.permutation(m)
        xy = base_xy[src] + np.float64([3.0, 1.0]) * (f % period) + r.normal(0, 0.5, (m, 2))
        d = base_d[src] + noise * r.standard_normal((m, 128))
        d[:nd] = r.standard_normal((nd, 128))
        xy[:nd] = np.column_stack([r.uniform(2, W_IMG - 3, nd), r.uniform(2, H_IMG - 3, nd)])
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        out.append((np.ascontiguousarray(xy, np.float32), np.ascontiguousarray(d, np.float32)))
    return out


class PlantedExtractor:
    """Random-init ALIKED descriptors are all alike (pairwise cosine 0.9995), so nothing matches on extracted frames.  To give
    the calls after `feature_extractor` real work, this wraps the detector's `extract_dev`: the extraction runs as always and,
    right behind it ON THE SAME STREAM, the frame's device record {count, keypoints, descriptors} is overwritten with the next
    set of a synthetic chain.  Everything downstream - the record read back into the caller's arrays, the look-ahead match the
    extraction enqueues, the device-resident keyframe matches - then sees one consistent frame, with no white-box access to
    the product's state.  `frames`: [(xy, desc)]; the sets are served round-robin from DEVICE-resident copies made here, once
    (r05: one device-to-device copy, ~10 us; as an upload from the host it put a copy-engine hand-over of ~18 us on either side
    of a 1 MB transfer between every extraction and the match behind it.  A copy on a stream of its own, concurrent with the
    extraction, was tried and measured the same frame time: not kept)."""

    def __init__(self, detector, frames):
        self.det, self.ctx = detector, detector.ctx
        self.real = detector.extract_dev
        self.i = 0
        self.sets = []
        for xy, desc in frames:
            n = len(xy)
            blk = np.empty(16 + n * 8 + n * 512, np.uint8)
            blk[:16].view(np.int32)[:] = (n, 0, 0, 0)
            blk[16:16 + n * 8].view(np.float32)[:] = xy.reshape(-1)
            blk[16 + n * 8:].view(np.float32)[:] = desc.reshape(-1)
            dev = self.ctx.malloc(blk.nbytes)
            self.ctx.h2d(dev, blk)
            self.sets.append((n, dev, blk.nbytes))
        self.ctx.sync()
        detector.extract_dev = self

    def __call__(self, img_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        self.real(img_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=max_kpts)
        n, dev, nbytes = self.sets[self.i % len(self.sets)]
        self.i += 1
        if xy_out == n_out + 16 and desc_out == xy_out + n * 8:        # a full record [count | xy | desc] in one block: one copy
            self.ctx.d2d_async(n_out, dev, nbytes)
            return
        self.ctx.d2d_async(n_out, dev, 16)
        self.ctx.d2d_async(xy_out, dev + 16, n * 8)
        self.ctx.d2d_async(desc_out, dev + 16 + n * 8, n * 512)

    def restore(self):
        del self.det.extract_dev
        self.ctx.sync()
        for _, dev, _ in self.sets:
            self.ctx.free(dev)
        self.sets = []


class PlantedBatchExtractor:
    """`PlantedExtractor` for the BATCHED extractor entry the frame pipeline drives (`extract_batch_dev`, frame_shard.py): the F
    extractions run as always and, right behind them on the same stream, every frame's pipeline record [xy K x 2 | desc K x 128 |
    count] is overwritten with the next set of a synthetic MATCHED chain - one device-to-device copy per frame when the chain's
    sets fill the record (n == max_kpts: xy, descriptors and count are contiguous), three otherwise.  What the batched matches,
    the read-back and the collation see is then one consistent stream of frames that really match (~600 matches per pair with
    the `match_gain` weights), at the cost of F small copies per call inside the timed region."""

    def __init__(self, detector, frames, max_kpts):
        self.det, self.ctx = detector, detector.ctx
        self.real = detector.extract_batch_dev
        self.K = K = int(max_kpts)
        self.i = 0
        self.sets = []
        for xy, desc in frames:
            n = len(xy)
            rec = np.zeros(K * 130 + 4, np.float32)
            rec[:2 * n] = xy.reshape(-1)
            rec[2 * K:2 * K + 128 * n] = desc.reshape(-1)
            rec[K * 130:K * 130 + 1].view(np.int32)[0] = n
            dev = self.ctx.malloc(rec.nbytes)
            self.ctx.h2d(dev, rec)
            self.sets.append((n, dev))
        self.ctx.sync()
        detector.extract_batch_dev = self

    def __call__(self, imgs_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=None):
        self.real(imgs_dev, H, Wd, Cn, xy_out, desc_out, score_out, n_out, max_kpts=max_kpts)
        K = self.K
        for xy_p, de_p, n_p in zip(xy_out, desc_out, n_out):
            n, dev = self.sets[self.i % len(self.sets)]
            self.i += 1
            if de_p == xy_p + K * 8 and n_p == xy_p + K * 520:          # one pipeline record: one copy
                self.ctx.d2d_async(xy_p, dev, K * 520 + 16)
            else:
                self.ctx.d2d_async(xy_p, dev, n * 8)
                self.ctx.d2d_async(de_p, dev + K * 8, n * 512)
                self.ctx.d2d_async(n_p, dev + K * 520, 16)

    def restore(self):
        del self.det.extract_batch_dev
        self.ctx.sync()
        for _, dev in self.sets:
            self.ctx.free(dev)
        self.sets = []
