"""Synthetic frames of SURVEY.md section 8(d): uint8[376,1241,3] white noise from
default_rng(1234 + frame_idx), and the structured variant (9x9 box low-pass of
the same noise, stretched to 0..255, shifted 3 px per frame)."""
import numpy as np

H_KITTI, W_KITTI = 376, 1241


def noise_frame(idx, h=H_KITTI, w=W_KITTI, c=3):
    rng = np.random.default_rng(1234 + idx)
    shape = (h, w, c) if c > 1 else (h, w)
    return rng.integers(0, 256, shape, dtype=np.uint8)


def structured_frame(idx, h=H_KITTI, w=W_KITTI, c=3, k=9):
    f = noise_frame(0, h, w, c).astype(np.float32)
    if f.ndim == 2:
        f = f[:, :, None]
    pad = np.pad(f, ((k // 2, k // 2), (k // 2, k // 2), (0, 0)), mode="edge")
    cs = np.cumsum(np.cumsum(pad, 0), 1)
    cs = np.pad(cs, ((1, 0), (1, 0), (0, 0)))
    f = (cs[k:, k:] - cs[:-k, k:] - cs[k:, :-k] + cs[:-k, :-k]) / (k * k)
    f = (f - f.min()) / (f.max() - f.min()) * 255.0
    out = np.roll(f, 3 * idx, axis=1).astype(np.uint8)
    return out if c > 1 else out[:, :, 0]


def disc_pair(shift=(5, 3)):
    """The reference's own tests/test_lightglue_vs_manual.py:16-27 input: two 200x200 BGR
    images with four white discs of radius 5, the second shifted by (+5, +3)."""
    def draw(dx, dy):
        img = np.zeros((200, 200, 3), np.uint8)
        yy, xx = np.mgrid[0:200, 0:200]
        for x, y in [(50, 50), (150, 50), (50, 150), (150, 150)]:
            img[(xx - x - dx) ** 2 + (yy - y - dy) ** 2 <= 25] = 255
        return img
    return draw(0, 0), draw(*shift)
