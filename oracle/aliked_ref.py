"""CPU oracle: ALIKED-n16 extraction, torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  The reference calls `lightglue.ALIKED(max_num_keypoints=...)
.extract(img)` (slam/core/features_utils.py:25, :92-100); the `lightglue`
package (requirements.txt:1, cvg/LightGlue HEAD) and its dependencies
torchvision (`deform_conv2d`) and kornia (`resize`) are absent from
/root/reference and from this image, and no `aliked-n16.pth` is on disk.  This
module restates the published algorithm of cvg/LightGlue `lightglue/aliked.py`
+ `lightglue/utils.py` (Extractor.extract, ImagePreprocessor), torchvision's
`deform_conv2d` and kornia's antialiased `resize`, anchored on the reference's
call site:

  features_utils.py:219-222  BGR uint8 -> RGB float32 / 255, CHW
  ALIKED(model 'aliked-n16', detection_threshold 0.2, nms_radius 2,
         preprocess resize 1024 long side), max_num_keypoints = max_features
  features_utils.py:95-100   rbd, keypoints -> (x, y), descriptors re-normalised
                             rows / (||row|| + 1e-8)

Restated details (each a place a checkpoint-equipped upstream run must confirm):
  * resize: kornia `resize(side='long', antialias=True)`: gaussian blur with
    sigma = max((factor-1)/2, 0.001), kernel = max(int(4 sigma), 3) made odd,
    reflect border, then bilinear interpolate (align_corners=False)
  * padding to /32: `InputPadder` centred, replicate mode; maps un-padded again
    before DKD / SDDH
  * DKD threshold mode (top_k = -1): 5x5 NMS with two recovery rounds, border
    of `radius` zeroed, score > 0.2 (mean fallback when none), raster order,
    if more than n_limit: top n_limit by score (this oracle: stable sort =
    ties broken by raster order), 5x5 soft-argmax refinement T = 0.1, score by
    bilinear grid_sample(align_corners=True)
  * SDDH: 3x3 patch (get_patches corner rule) -> 16 offsets -> bilinear samples
    -> 1x1 conv + SELU -> einsum with agg_weights -> L2 normalise
  * keypoints back to input pixels: (kp + 0.5) / scale - 0.5

State-dict keys are upstream's.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

CFG = dict(c1=16, c2=32, c3=64, c4=128, dim=128, K=3, M=16,
           detection_threshold=0.2, nms_radius=2, resize=1024)
BN_EPS = 1e-5


# --------------------------------------------------------------------------- #
#  pre-processing (features_utils.py:219-222, lightglue.utils.ImagePreprocessor)
# --------------------------------------------------------------------------- #
def bgr_to_tensor(image: np.ndarray) -> torch.Tensor:
    """features_utils.py:219-222.  HxW gray input (what IMREAD_UNCHANGED gives for
    KITTI, main_revamped.py:112) is replicated to 3 channels, as upstream
    ALIKED does for 1-channel tensors."""
    if image.ndim == 2:
        image = np.repeat(image[:, :, None], 3, axis=2)
    if image.shape[2] == 4:
        image = image[:, :, :3]
    rgb = image[:, :, ::-1].astype(np.float32) / 255.0
    return torch.from_numpy(np.ascontiguousarray(rgb)).permute(2, 0, 1).unsqueeze(0)


def gaussian_kernel1d(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def resize_plan(h: int, w: int, resize: int = 1024):
    """kornia side='long' size rule + antialias blur parameters."""
    ar = w / h
    if ar > 1:
        new_h, new_w = int(resize / ar), resize
    else:
        new_h, new_w = resize, int(resize * ar)
    fy, fx = h / new_h, w / new_w
    blur = max(fy, fx) > 1
    sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
    ky, kx = int(max(2.0 * 2 * sy, 3)), int(max(2.0 * 2 * sx, 3))
    ky += (ky % 2 == 0)
    kx += (kx % 2 == 0)
    return dict(new_h=new_h, new_w=new_w, blur=blur, sigma_y=sy, sigma_x=sx, ky=ky, kx=kx)


def preprocess(img: torch.Tensor, resize: int = 1024):
    """ImagePreprocessor(resize=1024, side='long', antialias=True): returns
    (resized [1,3,h,w], scales [2] = (w'/w, h'/h))."""
    h, w = img.shape[-2:]
    p = resize_plan(h, w, resize)
    x = img
    if p["blur"]:
        gx = gaussian_kernel1d(p["kx"], p["sigma_x"])
        gy = gaussian_kernel1d(p["ky"], p["sigma_y"])
        c = x.shape[1]
        xp = F.pad(x, (p["kx"] // 2, p["kx"] // 2, 0, 0), mode="reflect")
        x = F.conv2d(xp, gx.view(1, 1, 1, -1).repeat(c, 1, 1, 1), groups=c)
        xp = F.pad(x, (0, 0, p["ky"] // 2, p["ky"] // 2), mode="reflect")
        x = F.conv2d(xp, gy.view(1, 1, -1, 1).repeat(c, 1, 1, 1), groups=c)
    x = F.interpolate(x, size=(p["new_h"], p["new_w"]), mode="bilinear", align_corners=False)
    scales = torch.tensor([x.shape[-1] / w, x.shape[-2] / h], dtype=torch.float32)
    return x, scales


def pad_amounts(h: int, w: int, div: int = 32):
    pad_h = (((h // div) + 1) * div - h) % div
    pad_w = (((w // div) + 1) * div - w) % div
    return [pad_w // 2, pad_w - pad_w // 2, pad_h // 2, pad_h - pad_h // 2]   # l, r, t, b


# --------------------------------------------------------------------------- #
#  network
# --------------------------------------------------------------------------- #
def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.0, BN_EPS)


def deform_conv2d(x, offset, weight, padding=1):
    """torchvision.ops.deform_conv2d (stride 1, dilation 1, one offset group, no mask).
    offset channel 2k = dy, 2k+1 = dx of kernel tap k (row-major)."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    assert B == 1
    ys = torch.arange(H, dtype=torch.float32).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, W)
    cols = []
    xf = x[0].reshape(C, H * W)
    for k in range(kh * kw):
        ki, kj = k // kw, k % kw
        py = ys - padding + ki + offset[0, 2 * k]
        px = xs - padding + kj + offset[0, 2 * k + 1]
        inside = ~((py <= -1) | (py >= H) | (px <= -1) | (px >= W))
        y0, x0 = torch.floor(py), torch.floor(px)
        ly, lx = py - y0, px - x0
        hy, hx = 1 - ly, 1 - lx
        y0, x0 = y0.long(), x0.long()
        y1, x1 = y0 + 1, x0 + 1

        def tap(yy, xx, ok):
            ok = ok & inside
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(-1)
            return xf[:, idx] * ok.reshape(1, -1).float()

        v1 = tap(y0, x0, (y0 >= 0) & (x0 >= 0))
        v2 = tap(y0, x1, (y0 >= 0) & (x1 <= W - 1))
        v3 = tap(y1, x0, (y1 <= H - 1) & (x0 >= 0))
        v4 = tap(y1, x1, (y1 <= H - 1) & (x1 <= W - 1))
        w1, w2, w3, w4 = (hy * hx).reshape(1, -1), (hy * lx).reshape(1, -1), \
                         (ly * hx).reshape(1, -1), (ly * lx).reshape(1, -1)
        cols.append(w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4)      # [C, H*W]
    col = torch.stack(cols, 1).reshape(C * kh * kw, H * W)       # (c, k) order = weight layout
    out = weight.reshape(Co, C * kh * kw) @ col
    return out.reshape(1, Co, H, W)


def _conv(sd, p, x, dcn):
    if not dcn:
        return F.conv2d(x, sd[p + ".weight"], None, padding=1)
    h, w = x.shape[2:]
    max_offset = max(h, w) / 4.0
    off = F.conv2d(x, sd[p + ".offset_conv.weight"], sd[p + ".offset_conv.bias"], padding=1)
    off = off.clamp(-max_offset, max_offset)
    return deform_conv2d(x, off, sd[p + ".regular_conv.weight"], padding=1)


def conv_block(sd, p, x):
    x = F.selu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, False)))
    return F.selu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", x, False)))


def res_block(sd, p, x, dcn):
    out = F.selu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, dcn)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, dcn))
    identity = F.conv2d(x, sd[p + ".downsample.weight"], sd.get(p + ".downsample.bias"))
    return F.selu(out + identity)


def extract_dense_map(sd, image, return_debug=False):
    h, w = image.shape[-2:]
    pl, pr, pt, pb = pad_amounts(h, w)
    x = F.pad(image, (pl, pr, pt, pb), mode="replicate")
    x1 = conv_block(sd, "block1", x)
    x2 = res_block(sd, "block2", F.avg_pool2d(x1, 2, 2), False)
    x3 = res_block(sd, "block3", F.avg_pool2d(x2, 4, 4), True)
    x4 = res_block(sd, "block4", F.avg_pool2d(x3, 4, 4), True)
    g1 = F.selu(F.conv2d(x1, sd["conv1.weight"]))
    g2 = F.selu(F.conv2d(x2, sd["conv2.weight"]))
    g3 = F.selu(F.conv2d(x3, sd["conv3.weight"]))
    g4 = F.selu(F.conv2d(x4, sd["conv4.weight"]))
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=True)  # noqa: E731
    x1234 = torch.cat([g1, up(g2, 2), up(g3, 8), up(g4, 32)], dim=1)
    s = F.selu(F.conv2d(x1234, sd["score_head.0.weight"]))
    s = F.selu(F.conv2d(s, sd["score_head.2.weight"], padding=1))
    s = F.selu(F.conv2d(s, sd["score_head.4.weight"], padding=1))
    score_map = torch.sigmoid(F.conv2d(s, sd["score_head.6.weight"], padding=1))
    feature_map = F.normalize(x1234, p=2, dim=1)
    H, W = feature_map.shape[-2:]
    feature_map = feature_map[..., pt:H - pb, pl:W - pr]
    score_map = score_map[..., pt:H - pb, pl:W - pr]
    if return_debug:
        return feature_map, score_map, dict(x1=x1, x2=x2, x3=x3, x4=x4, g1=g1, g2=g2, g3=g3, g4=g4)
    return feature_map, score_map


# --------------------------------------------------------------------------- #
#  DKD
# --------------------------------------------------------------------------- #
def simple_nms(scores, r):
    zeros = torch.zeros_like(scores)
    mp = lambda t: F.max_pool2d(t, kernel_size=2 * r + 1, stride=1, padding=r)   # noqa: E731
    max_mask = scores == mp(scores)
    for _ in range(2):
        supp_mask = mp(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == mp(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def dkd(score_map, n_limit, radius=2, thr=0.2, temperature=0.1):
    b, c, h, w = score_map.shape
    nms = simple_nms(score_map, radius)
    nms[:, :, :radius, :] = 0
    nms[:, :, :, :radius] = 0
    nms[:, :, -radius:, :] = 0
    nms[:, :, :, -radius:] = 0
    mask = nms > thr
    if mask.sum() == 0:
        mask = nms > score_map.reshape(b, -1).mean(dim=1).reshape(b, 1, 1, 1)
    mask = mask.reshape(-1)
    sview = score_map.reshape(-1)
    indices = mask.nonzero()[:, 0]
    if len(indices) > n_limit:
        sort_idx = sview[indices].sort(descending=True, stable=True)[1]
        indices = indices[sort_idx[:n_limit]]
    k = 2 * radius + 1
    x = torch.linspace(-radius, radius, k)
    hw_grid = torch.stack(torch.meshgrid([x, x], indexing="ij")).view(2, -1).t()[:, [1, 0]]
    patches = F.unfold(score_map, k, padding=radius)[0].t()          # (H*W) x k^2
    patch = patches[indices]
    xy_nms = torch.stack([indices % w, torch.div(indices, w, rounding_mode="trunc")], dim=1)
    max_v = patch.max(dim=1).values[:, None]
    x_exp = ((patch - max_v) / temperature).exp()
    xy_res = x_exp @ hw_grid / x_exp.sum(dim=1)[:, None]
    wh = torch.tensor([w - 1, h - 1], dtype=torch.float32)
    kp = (xy_nms + xy_res) / wh * 2 - 1
    kscore = F.grid_sample(score_map, kp.view(1, 1, -1, 2), mode="bilinear", align_corners=True)[0, 0, 0, :]
    return kp, kscore, indices


# --------------------------------------------------------------------------- #
#  SDDH
# --------------------------------------------------------------------------- #
def get_patches(tensor, required_corners, ps):
    c, h, w = tensor.shape
    corner = (required_corners - ps / 2 + 1).long()
    corner[:, 0] = corner[:, 0].clamp(min=0, max=w - 1 - ps)
    corner[:, 1] = corner[:, 1].clamp(min=0, max=h - 1 - ps)
    offset = torch.arange(0, ps)
    xg, yg = torch.meshgrid(offset, offset, indexing="ij")
    patches = torch.stack((xg, yg)).permute(2, 1, 0).unsqueeze(2)
    patches = patches.to(corner) + corner[None, None]
    pts = patches.reshape(-1, 2)
    sampled = tensor.permute(1, 2, 0)[tuple(pts.T)[::-1]]
    sampled = sampled.reshape(ps, ps, -1, c)
    return sampled.permute(2, 3, 0, 1)


def sddh(sd, fmap, kp, K=3, M=16):
    x = fmap[0]
    c, h, w = x.shape
    wh = torch.tensor([[w - 1, h - 1]], dtype=torch.float32)
    max_offset = max(h, w) / 4.0
    n = len(kp)
    kwh = (kp / 2 + 0.5) * wh
    patch = get_patches(x, kwh.long(), K)
    off = F.conv2d(patch, sd["desc_head.offset_conv.0.weight"], sd["desc_head.offset_conv.0.bias"])
    off = F.conv2d(F.selu(off), sd["desc_head.offset_conv.2.weight"], sd["desc_head.offset_conv.2.bias"])
    off = off.clamp(-max_offset, max_offset)
    off = off[:, :, 0, 0].view(n, 2, M).permute(0, 2, 1)
    pos = kwh.unsqueeze(1) + off
    pos = 2.0 * pos / wh[None] - 1
    pos = pos.reshape(1, n * M, 1, 2)
    feats = F.grid_sample(x.unsqueeze(0), pos, mode="bilinear", align_corners=True)
    feats = feats.reshape(c, n, M, 1).permute(1, 0, 2, 3)
    feats = F.selu(F.conv2d(feats, sd["desc_head.sf_conv.weight"])).squeeze(-1)
    descs = torch.einsum("ncp,pcd->nd", feats, sd["desc_head.agg_weights"])
    return F.normalize(descs, p=2.0, dim=1), off


# --------------------------------------------------------------------------- #
#  extract (what features_utils.py:85-101 returns on the LightGlue path)
# --------------------------------------------------------------------------- #
@torch.no_grad()
def aliked_extract(sd, image_u8: np.ndarray, max_kpts: int = 2048, return_debug=False):
    sd = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
    t0 = bgr_to_tensor(image_u8)
    img, scales = preprocess(t0, CFG["resize"])
    if return_debug:
        fmap, smap, dbg = extract_dense_map(sd, img, True)
    else:
        fmap, smap = extract_dense_map(sd, img)
        dbg = {}
    kp, kscore, idx = dkd(smap, max_kpts, CFG["nms_radius"], CFG["detection_threshold"])
    desc, off = sddh(sd, fmap, kp, CFG["K"], CFG["M"])
    _, _, h, w = img.shape
    wh = torch.tensor([w - 1, h - 1], dtype=torch.float32)
    kpts = wh * (kp + 1) / 2.0
    kpts = (kpts + 0.5) / scales[None] - 0.5
    des = desc.numpy().astype(np.float32)
    des = des / (np.linalg.norm(des, axis=1, keepdims=True) + 1e-8).astype(np.float32)   # features_utils.py:100
    out = dict(keypoints=kpts.numpy(), descriptors=des, scores=kscore.numpy(), indices=idx.numpy())
    if return_debug:
        dbg.update(img=img, score_map=smap, feature_map=fmap, kp_norm=kp, offsets=off, scales=scales)
        out["debug"] = dbg
    return out


def flops_dense(h=320, w=1024):
    """Dense-conv FLOPs per frame at the padded network size (6.56 GF)."""
    px = h * w
    f = 2 * 9 * (3 * 16 + 16 * 16) * px
    f += (2 * 9 * (16 * 32 + 32 * 32) + 2 * 16 * 32) * px / 4
    f += (2 * 9 * (32 * 64 + 64 * 64 + 32 * 18 + 64 * 18) + 2 * 32 * 64) * px / 64
    f += (2 * 9 * (64 * 128 + 128 * 128 + 64 * 18 + 128 * 18) + 2 * 64 * 128) * px / 1024
    f += 2 * 32 * (16 * px + 32 * px / 4 + 64 * px / 64 + 128 * px / 1024)
    f += (2 * 128 * 8 + 2 * 9 * (8 * 4 + 4 * 4 + 4)) * px
    return f
