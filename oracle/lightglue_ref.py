"""CPU oracle: LightGlue(features='aliked') forward, torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  The reference calls `lightglue.LightGlue(features='aliked')`
(slam/core/features_utils.py:26, :157-162); the `lightglue` package
(requirements.txt:1 `lightglue==0.0` = git+https://github.com/cvg/LightGlue.git,
no commit pin) is absent from /root/reference and from this image, and no
checkpoint is on disk.  This module restates the published algorithm of
cvg/LightGlue `lightglue/lightglue.py` as it runs on the torch-CPU device the
reference falls back to, anchored on the reference's call site:

  * input: {'keypoints','descriptors'} only, no 'image_size'
    (features_utils.py:158-161) -> keypoints normalised by their own bounding
    box (`normalize_keypoints(kpts, None)`)
  * conf: input_dim 128, descriptor_dim 256, 9 layers, 4 heads,
    depth_confidence 0.95, width_confidence 0.99, filter_threshold 0.1
  * CPU execution path: self-attention through
    F.scaled_dot_product_attention, cross-attention through the einsum branch
    (sqrt(scale) on both operands, softmax over each side), early stopping ON,
    point pruning threshold `pruning_keypoint_thresholds['cpu'] = -1`, i.e.
    pruning is evaluated after every layer whenever width_confidence > 0
    (SURVEY.md App. A.2 reads -1 as "never"; upstream's test is
    `desc.shape[-2] > pruning_th`, which -1 always satisfies - kept behind the
    `prune_min_kpts` switch so either reading can be selected)
  * output: matches [K,2] ascending in index 0, scores [K], stop layer,
    prune counters; then the reference keeps `scores > min_conf`
    (features_utils.py:164-169).

State-dict keys are upstream's so a real
`aliked_lightglue.pth` loads unchanged.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_CONF = dict(
    input_dim=128, descriptor_dim=256, n_layers=9, num_heads=4,
    depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1,
    prune_min_kpts=-1,        # 'cpu' entry of upstream's pruning_keypoint_thresholds
)


def normalize_keypoints(kpts: torch.Tensor, size=None) -> torch.Tensor:
    """`normalize_keypoints(kpts, size)`: size = 1 + max - min when None (the split API of the reference,
    features_utils.py:157-162, passes no 'image_size'); the (W, H) of the image when the features carry it - the
    legacy pair entry features_utils.py:233-247 hands `extractor.extract` dicts, which do."""
    if size is None:
        size = 1 + kpts.max(-2).values - kpts.min(-2).values
    else:
        size = torch.as_tensor(size, dtype=kpts.dtype).reshape(1, 2)
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def posenc(sd, kpts):
    """LearnableFourierPositionalEncoding(2, 64, 64): [2, B, 1, N, 64]."""
    projected = F.linear(kpts, sd["posenc.Wr.weight"])
    emb = torch.stack([torch.cos(projected), torch.sin(projected)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_rotary(freqs, t):
    return (t * freqs[0]) + (rotate_half(t) * freqs[1])


def _ffn(sd, p, x):
    h = F.linear(x, sd[p + ".ffn.0.weight"], sd[p + ".ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd[p + ".ffn.1.weight"], sd[p + ".ffn.1.bias"], 1e-5)
    h = F.gelu(h)
    return F.linear(h, sd[p + ".ffn.3.weight"], sd[p + ".ffn.3.bias"])


PROBE = None      # a dict while a caller wants statistics of the cross-attention logits (never set by the tests)


def self_block(sd, i, x, enc, heads):
    p = f"transformers.{i}.self_attn"
    qkv = F.linear(x, sd[p + ".Wqkv.weight"], sd[p + ".Wqkv.bias"])
    qkv = qkv.unflatten(-1, (heads, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = apply_rotary(enc, q)
    k = apply_rotary(enc, k)
    if q.shape[-2] == 0:
        ctx = q.new_zeros((*q.shape[:-1], v.shape[-1]))
    else:
        ctx = F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())
    msg = F.linear(ctx.transpose(1, 2).flatten(start_dim=-2),
                   sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
    return x + _ffn(sd, p, torch.cat([x, msg], -1))


def cross_block(sd, i, x0, x1, heads):
    p = f"transformers.{i}.cross_attn"
    dim_head = x0.shape[-1] // heads
    scale = dim_head ** -0.5
    qk0 = F.linear(x0, sd[p + ".to_qk.weight"], sd[p + ".to_qk.bias"])
    qk1 = F.linear(x1, sd[p + ".to_qk.weight"], sd[p + ".to_qk.bias"])
    v0 = F.linear(x0, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    v1 = F.linear(x1, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    qk0, qk1, v0, v1 = (t.unflatten(-1, (heads, -1)).transpose(1, 2) for t in (qk0, qk1, v0, v1))
    qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
    sim = torch.einsum("bhid, bhjd -> bhij", qk0, qk1)
    attn01 = F.softmax(sim, dim=-1)
    attn10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    if PROBE is not None and sim.numel():               # (scripts/flip_soak.py: how peaked the attention of these weights is)
        PROBE["max_abs_logit"] = max(PROBE.get("max_abs_logit", 0.0), float(sim.abs().max()))
        PROBE.setdefault("row_max_weight", []).append(float(attn01.max(-1).values.mean()))
    m0 = torch.einsum("bhij, bhjd -> bhid", attn01, v1)
    m1 = torch.einsum("bhji, bhjd -> bhid", attn10.transpose(-2, -1), v0)
    m0, m1 = (t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1))
    m0 = F.linear(m0, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])
    m1 = F.linear(m1, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])
    x0 = x0 + _ffn(sd, p, torch.cat([x0, m0], -1))
    x1 = x1 + _ffn(sd, p, torch.cat([x1, m1], -1))
    return x0, x1


def confidence_threshold(layer_index: int, n_layers: int) -> float:
    thr = 0.8 + 0.1 * np.exp(-4.0 * layer_index / n_layers)
    return float(np.clip(thr, 0, 1))


def token_confidence(sd, i, x):
    return torch.sigmoid(F.linear(x, sd[f"token_confidence.{i}.token.0.weight"],
                                  sd[f"token_confidence.{i}.token.0.bias"])).squeeze(-1)


def matchability(sd, i, x):
    return F.linear(x, sd[f"log_assignment.{i}.matchability.weight"],
                    sd[f"log_assignment.{i}.matchability.bias"])


def sigmoid_log_double_softmax(sim, z0, z1):
    b, m, n = sim.shape
    certainties = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    scores0 = F.log_softmax(sim, 2)
    scores1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    scores = sim.new_full((b, m + 1, n + 1), 0)
    scores[:, :m, :n] = scores0 + scores1 + certainties
    scores[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def log_assignment(sd, i, x0, x1):
    p = f"log_assignment.{i}"
    md0 = F.linear(x0, sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"])
    md1 = F.linear(x1, sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"])
    d = md0.shape[-1]
    md0, md1 = md0 / d ** 0.25, md1 / d ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    return sigmoid_log_double_softmax(sim, matchability(sd, i, x0), matchability(sd, i, x1)), sim


def filter_matches(scores, th):
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    m0, m1 = max0.indices, max1.indices
    indices0 = torch.arange(m0.shape[1])[None]
    indices1 = torch.arange(m1.shape[1])[None]
    mutual0 = indices0 == m1.gather(1, m0)
    mutual1 = indices1 == m0.gather(1, m1)
    max0_exp = max0.values.exp()
    zero = max0_exp.new_tensor(0)
    mscores0 = torch.where(mutual0, max0_exp, zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, m1), zero)
    valid0 = mutual0 & (mscores0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    m0 = torch.where(valid0, m0, -1)
    m1 = torch.where(valid1, m1, -1)
    return m0, m1, mscores0, mscores1


@torch.no_grad()
def lightglue_forward(sd, kpts0, desc0, kpts1, desc1, conf=None, return_debug=False, size0=None, size1=None):
    """kpts [M,2]/[N,2] pixel coords, desc [M,128]/[N,128]; returns dict with
    matches [K,2] int64, scores [K], stop (1-based layer count), prune0/prune1."""
    c = dict(DEFAULT_CONF)
    c.update(conf or {})
    sd = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
    kpts0 = torch.as_tensor(kpts0, dtype=torch.float32)[None]
    kpts1 = torch.as_tensor(kpts1, dtype=torch.float32)[None]
    x0 = torch.tensor(np.asarray(desc0), dtype=torch.float32)[None].contiguous()      # (a copy: the drop-in hands out read-only descriptor arrays)
    x1 = torch.tensor(np.asarray(desc1), dtype=torch.float32)[None].contiguous()
    m, n = kpts0.shape[1], kpts1.shape[1]
    L, H = c["n_layers"], c["num_heads"]
    dbg = {}

    k0 = normalize_keypoints(kpts0, size0).clone()
    k1 = normalize_keypoints(kpts1, size1).clone()
    x0 = F.linear(x0, sd["input_proj.weight"], sd["input_proj.bias"])
    x1 = F.linear(x1, sd["input_proj.weight"], sd["input_proj.bias"])
    enc0, enc1 = posenc(sd, k0), posenc(sd, k1)
    if return_debug:
        dbg["kn0"], dbg["kn1"] = k0[0].clone(), k1[0].clone()
        dbg["x_in0"], dbg["x_in1"] = x0[0].clone(), x1[0].clone()
        dbg["layers"] = []

    do_early_stop = c["depth_confidence"] > 0
    do_prune = c["width_confidence"] > 0
    pruning_th = c["prune_min_kpts"]
    ind0 = torch.arange(0, m)[None]
    ind1 = torch.arange(0, n)[None]
    prune0 = torch.ones_like(ind0)
    prune1 = torch.ones_like(ind1)
    thr = [confidence_threshold(i, L) for i in range(L)]

    tok0 = tok1 = None
    i = 0
    for i in range(L):
        if x0.shape[1] == 0 or x1.shape[1] == 0:
            break
        x0 = self_block(sd, i, x0, enc0, H)
        x1 = self_block(sd, i, x1, enc1, H)
        if return_debug:
            dbg["layers"].append({"self0": x0[0].clone(), "self1": x1[0].clone()})
        x0, x1 = cross_block(sd, i, x0, x1, H)
        if return_debug:
            dbg["layers"][-1].update({"cross0": x0[0].clone(), "cross1": x1[0].clone()})
        if i == L - 1:
            continue
        if do_early_stop:
            tok0, tok1 = token_confidence(sd, i, x0), token_confidence(sd, i, x1)
            confs = torch.cat([tok0[..., :m], tok1[..., :n]], -1)
            ratio = 1.0 - (confs < thr[i]).float().sum() / (m + n)
            if ratio > c["depth_confidence"]:
                break
        if do_prune and x0.shape[-2] > pruning_th:
            s0 = torch.sigmoid(matchability(sd, i, x0)).squeeze(-1)
            keep = s0 > (1 - c["width_confidence"])
            if tok0 is not None:
                keep |= tok0 <= thr[i]
            keep0 = torch.where(keep)[1]
            ind0 = ind0.index_select(1, keep0)
            x0 = x0.index_select(1, keep0)
            enc0 = enc0.index_select(-2, keep0)
            prune0[:, ind0] += 1
        if do_prune and x1.shape[-2] > pruning_th:
            s1 = torch.sigmoid(matchability(sd, i, x1)).squeeze(-1)
            keep = s1 > (1 - c["width_confidence"])
            if tok1 is not None:
                keep |= tok1 <= thr[i]
            keep1 = torch.where(keep)[1]
            ind1 = ind1.index_select(1, keep1)
            x1 = x1.index_select(1, keep1)
            enc1 = enc1.index_select(-2, keep1)
            prune1[:, ind1] += 1

    if x0.shape[1] == 0 or x1.shape[1] == 0:
        return {"matches": torch.empty(0, 2, dtype=torch.long), "scores": torch.empty(0),
                "stop": i + 1, "prune0": prune0[0], "prune1": prune1[0], "debug": dbg}

    scores, sim = log_assignment(sd, i, x0, x1)
    m0, m1, ms0, ms1 = filter_matches(scores, c["filter_threshold"])
    valid = m0[0] > -1
    mi0 = torch.where(valid)[0]
    mi1 = m0[0][valid]
    if do_prune:
        mi0 = ind0[0, mi0]
        mi1 = ind1[0, mi1]
    out = {"matches": torch.stack([mi0, mi1], -1), "scores": ms0[0][valid], "stop": i + 1,
           "prune0": prune0[0], "prune1": prune1[0]}
    if return_debug:
        dbg.update({"sim": sim[0], "log_scores": scores[0], "ind0": ind0[0], "ind1": ind1[0],
                    "x_out0": x0[0], "x_out1": x1[0]})
        out["debug"] = dbg
    return out


def reference_feature_matcher(sd, kp0_xy, kp1_xy, des0, des1, min_conf=0.7, conf=None):
    """What slam/core/features_utils.py:109-171 returns on the LightGlue path:
    list of (queryIdx, trainIdx) with score > min_conf, ascending queryIdx."""
    if len(kp0_xy) == 0 or len(kp1_xy) == 0:
        return np.zeros((0, 2), np.int64), np.zeros((0,), np.float32), 0
    out = lightglue_forward(sd, kp0_xy, des0, kp1_xy, des1, conf)
    keep = out["scores"] > float(min_conf)
    return out["matches"][keep].numpy(), out["scores"][keep].numpy(), out["stop"]


def flops(n: int, layers: int, d: int = 256, d_in: int = 128) -> float:
    """SURVEY.md section 8(d) algorithmic FLOPs per pair at M = N = n."""
    per_layer = (2 * (6 * n * d * d + 4 * n * n * d + 2 * n * d * d + 8 * n * d * d + 4 * n * d * d)
                 + 2 * (4 * n * d * d + 4 * n * n * d + 2 * n * d * d + 8 * n * d * d + 4 * n * d * d))
    return 4 * n * d_in * d + layers * per_layer + (4 * n * d * d + 2 * n * n * d)
