"""Weights: upstream state-dict layout <-> the flat fp32 blobs the C-ABI takes.

PyTorch is used here for weight LOADING only (`torch.load` of an upstream
`.pth`); everything else is numpy.  The reference downloads
`aliked-n16.pth` / `aliked_lightglue.pth` through torch.hub at construction
(slam/core/features_utils.py:25-26); there is no network here, so
`random_*_state_dict` provides seeded random-init weights of the same
architecture and key names for tests and synthetic benchmarks.

Blob layout: tensors in the fixed order of `LIGHTGLUE_ORDER` / `ALIKED_ORDER`,
each padded to a multiple of 64 floats (256 B) - csrc walks the same order.
"""
from __future__ import annotations

import numpy as np

PAD = 64
LG_LAYERS = 9
LG_DIM = 256
LG_HEADS = 4
LG_IN = 128


def _pad_cat(arrs):
    out = []
    for a in arrs:
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        n = (a.size + PAD - 1) // PAD * PAD
        b = np.zeros(n, np.float32)
        b[:a.size] = a
        out.append(b)
    return np.concatenate(out)


def to_numpy_state_dict(sd):
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().float().numpy()
        out[k] = np.asarray(v, np.float32)
    return out


def load_state_dict(path):
    """Read an upstream checkpoint (.pth) with torch; returns numpy arrays."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if "state_dict" in sd:
        sd = sd["state_dict"]
    # upstream renames old checkpoints: self_attn.i -> transformers.i.self_attn etc.
    ren = {}
    for k, v in sd.items():
        for i in range(LG_LAYERS):
            k = k.replace(f"self_attn.{i}.", f"transformers.{i}.self_attn.") if k.startswith("self_attn.") else k
            k = k.replace(f"cross_attn.{i}.", f"transformers.{i}.cross_attn.") if k.startswith("cross_attn.") else k
        ren[k.replace("matcher.", "") if k.startswith("matcher.") else k] = v
    return to_numpy_state_dict(ren)


# --------------------------------------------------------------------------- #
#  LightGlue(features='aliked')
# --------------------------------------------------------------------------- #
def random_lightglue_state_dict(seed=0, match_gain=1.0, conf_bias=0.0, match_bias=0.0, conf_gain=1.0, ffn_gain=1.0,
                                final_identity=0.0):
    """Seeded random init with upstream key names/shapes.
    nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm (1, 0).
    `match_gain` scales log_assignment.final_proj (sharper assignments),
    `conf_bias` / `match_bias` shift the token-confidence / matchability
    logits (to drive early stopping and pruning in tests); `conf_gain` scales the
    token-confidence weights (spreads the confidences, so that SOME points are
    confident - and prunable - layers before 95 % of them are and the pair stops).
    `ffn_gain` scales the output Linear of every FFN (<< 1: the token state stays close to the input
    projection of the descriptor) and `final_identity` = g > 0 replaces final_proj by g.I: together a
    random init that behaves like a mutual-nearest-neighbour matcher on the descriptors - every kernel
    still runs on every layer, and frames that overlap produce hundreds of matches, which an untrained
    transformer otherwise never does (benchmarks of the callers downstream of the matcher need them)."""
    rng = np.random.default_rng(seed)
    D = LG_DIM

    def lin(out_f, in_f, gain=1.0):
        b = gain / np.sqrt(in_f)
        return (rng.uniform(-b, b, (out_f, in_f)).astype(np.float32),
                rng.uniform(-b, b, (out_f,)).astype(np.float32))

    sd = {}
    sd["input_proj.weight"], sd["input_proj.bias"] = lin(D, LG_IN)
    sd["posenc.Wr.weight"] = rng.normal(0, 1.0, (D // LG_HEADS // 2, 2)).astype(np.float32)
    for i in range(LG_LAYERS):
        for blk, names in (("self_attn", (("Wqkv", 3 * D, D), ("out_proj", D, D))),
                           ("cross_attn", (("to_qk", D, D), ("to_v", D, D), ("to_out", D, D)))):
            p = f"transformers.{i}.{blk}"
            for nm, o, ii in names:
                sd[f"{p}.{nm}.weight"], sd[f"{p}.{nm}.bias"] = lin(o, ii)
            sd[f"{p}.ffn.0.weight"], sd[f"{p}.ffn.0.bias"] = lin(2 * D, 2 * D)
            sd[f"{p}.ffn.1.weight"] = (1.0 + 0.1 * rng.standard_normal(2 * D)).astype(np.float32)
            sd[f"{p}.ffn.1.bias"] = (0.1 * rng.standard_normal(2 * D)).astype(np.float32)
            sd[f"{p}.ffn.3.weight"], sd[f"{p}.ffn.3.bias"] = lin(D, 2 * D, gain=ffn_gain)
    for i in range(LG_LAYERS):
        p = f"log_assignment.{i}"
        w, b = lin(D, D, gain=match_gain)
        if final_identity > 0:
            w, b = np.float32(final_identity) * np.eye(D, dtype=np.float32), np.zeros(D, np.float32)
        sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"] = w, b
        w, b = lin(1, D)
        sd[p + ".matchability.weight"], sd[p + ".matchability.bias"] = w, b + np.float32(match_bias)
    for i in range(LG_LAYERS - 1):
        w, b = lin(1, D)
        sd[f"token_confidence.{i}.token.0.weight"] = w * np.float32(conf_gain)
        sd[f"token_confidence.{i}.token.0.bias"] = b + np.float32(conf_bias)
    return sd


def _qkv_row_perm():
    """Upstream Wqkv output column c = h*192 + d*3 + s  (unflatten(-1,(H,-1,3))).
    csrc wants [s][h][d] so q, k, v of one head are contiguous 64-wide slabs."""
    H, Dh = LG_HEADS, LG_DIM // LG_HEADS
    perm = np.empty(3 * LG_DIM, np.int64)
    for s in range(3):
        for h in range(H):
            for d in range(Dh):
                perm[s * LG_DIM + h * Dh + d] = h * Dh * 3 + d * 3 + s
    return perm


def _fold_out_proj(w1, b1, wo, bo):
    """ffn.0 applied to cat[x, out_proj(ctx)] == ffn.0' applied to cat[x, ctx] with
    W' = [W1x | W1m @ Wo],  b' = b1 + W1m @ bo  (W1 = [W1x | W1m] over the concat axis).
    The product is formed in float64 and rounded once; this removes one GEMM per block."""
    d = wo.shape[0]
    w1x, w1m = w1[:, :d].astype(np.float64), w1[:, d:].astype(np.float64)
    wf = np.concatenate([w1x, w1m @ wo.astype(np.float64)], axis=1)
    bf = b1.astype(np.float64) + w1m @ bo.astype(np.float64)
    return wf.astype(np.float32), bf.astype(np.float32)


def lightglue_order(sd):
    """(name, array) list in blob order.  Re-indexing / concatenation of upstream tensors, plus
    one algebraic fold per attention block: the output projection (out_proj / to_out) is
    multiplied into the message half of the FFN's first Linear (`_fold_out_proj`)."""
    perm = _qkv_row_perm()
    out = [("input_proj.weight", sd["input_proj.weight"]), ("input_proj.bias", sd["input_proj.bias"]),
           ("posenc.Wr.weight", sd["posenc.Wr.weight"])]
    for i in range(LG_LAYERS):
        p = f"transformers.{i}.self_attn"
        out += [(p + ".Wqkv.weight[perm]", sd[p + ".Wqkv.weight"][perm]),
                (p + ".Wqkv.bias[perm]", sd[p + ".Wqkv.bias"][perm])]
        wf, bf = _fold_out_proj(sd[p + ".ffn.0.weight"], sd[p + ".ffn.0.bias"],
                                sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
        out += [(p + ".ffn.0.weight*out_proj", wf), (p + ".ffn.0.bias*out_proj", bf)]
        out += [(p + f".ffn.{j}.{w}", sd[p + f".ffn.{j}.{w}"]) for j in (1, 3) for w in ("weight", "bias")]
        p = f"transformers.{i}.cross_attn"
        out += [(p + ".to_qk|to_v.weight", np.concatenate([sd[p + ".to_qk.weight"], sd[p + ".to_v.weight"]], 0)),
                (p + ".to_qk|to_v.bias", np.concatenate([sd[p + ".to_qk.bias"], sd[p + ".to_v.bias"]], 0))]
        wf, bf = _fold_out_proj(sd[p + ".ffn.0.weight"], sd[p + ".ffn.0.bias"],
                                sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])
        out += [(p + ".ffn.0.weight*to_out", wf), (p + ".ffn.0.bias*to_out", bf)]
        out += [(p + f".ffn.{j}.{w}", sd[p + f".ffn.{j}.{w}"]) for j in (1, 3) for w in ("weight", "bias")]
    for i in range(LG_LAYERS):
        p = f"log_assignment.{i}"
        out += [(p + ".final_proj.weight", sd[p + ".final_proj.weight"]),
                (p + ".final_proj.bias", sd[p + ".final_proj.bias"]),
                (p + ".matchability.weight", sd[p + ".matchability.weight"]),
                (p + ".matchability.bias", sd[p + ".matchability.bias"])]
    for i in range(LG_LAYERS - 1):
        p = f"token_confidence.{i}.token.0"
        out += [(p + ".weight", sd[p + ".weight"]), (p + ".bias", sd[p + ".bias"])]
    return out


def pack_lightglue(sd) -> np.ndarray:
    sd = to_numpy_state_dict(sd)
    return _pad_cat([a for _, a in lightglue_order(sd)])


# --------------------------------------------------------------------------- #
#  ALIKED-n16
# --------------------------------------------------------------------------- #
AL = dict(c1=16, c2=32, c3=64, c4=128, dim=128, K=3, M=16)


def random_aliked_state_dict(seed=0, score_gain=0.1, desc_centered=False):
    """Seeded random init with upstream key names/shapes:
    kaiming-uniform convs, BatchNorm with non-trivial running statistics.
    `score_gain` scales the last score-head conv (spread of the score map).
    `desc_centered`: the aggregation weights of the descriptor head are drawn from U(-0.5, 0.5)
    instead of upstream's U(0, 1) init.  With all-positive weights an untrained head sums its
    (SELU, positive-mean) features into almost the same direction for every keypoint - any two
    descriptors have cosine 0.9995 - so nothing downstream can match; zero-mean weights cancel the
    common component and overlapping frames give matchable descriptors (drop-in benchmarks)."""
    rng = np.random.default_rng(seed)

    def conv(co, ci, k, bias=False, gain=1.0):
        b = abs(gain) * np.sqrt(3.0 / (ci * k * k))
        w = (np.sign(gain) * rng.uniform(-b, b, (co, ci, k, k))).astype(np.float32)
        return (w, rng.uniform(-b, b, (co,)).astype(np.float32)) if bias else w

    def bn(p, c, sd):
        sd[p + ".weight"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[p + ".bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[p + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        sd[p + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    sd = {}
    c1, c2, c3, c4, dim = AL["c1"], AL["c2"], AL["c3"], AL["c4"], AL["dim"]
    sd["block1.conv1.weight"] = conv(c1, 3, 3, gain=1.7)
    bn("block1.bn1", c1, sd)
    sd["block1.conv2.weight"] = conv(c1, c1, 3, gain=1.7)
    bn("block1.bn2", c1, sd)
    for name, ci, co, dcn in (("block2", c1, c2, False), ("block3", c2, c3, True), ("block4", c3, c4, True)):
        for cv, cin in (("conv1", ci), ("conv2", co)):
            if dcn:
                w, b = conv(18, cin, 3, bias=True, gain=0.5)
                sd[f"{name}.{cv}.offset_conv.weight"], sd[f"{name}.{cv}.offset_conv.bias"] = w, b
                sd[f"{name}.{cv}.regular_conv.weight"] = conv(co, cin, 3, gain=1.7)
            else:
                sd[f"{name}.{cv}.weight"] = conv(co, cin, 3, gain=1.7)
        bn(f"{name}.bn1", co, sd)
        bn(f"{name}.bn2", co, sd)
        sd[f"{name}.downsample.weight"], sd[f"{name}.downsample.bias"] = conv(co, ci, 1, bias=True)
    for i, ci in enumerate((c1, c2, c3, c4), 1):
        sd[f"conv{i}.weight"] = conv(dim // 4, ci, 1, gain=1.7)
    sd["score_head.0.weight"] = conv(8, dim, 1, gain=1.7)
    sd["score_head.2.weight"] = conv(4, 8, 3, gain=1.7)
    sd["score_head.4.weight"] = conv(4, 4, 3, gain=1.7)
    sd["score_head.6.weight"] = conv(1, 4, 3, gain=1.7 * score_gain)
    w, b = conv(2 * AL["M"], dim, 3, bias=True, gain=2.0)
    sd["desc_head.offset_conv.0.weight"], sd["desc_head.offset_conv.0.bias"] = w, b
    w, b = conv(2 * AL["M"], 2 * AL["M"], 1, bias=True, gain=4.0)
    sd["desc_head.offset_conv.2.weight"], sd["desc_head.offset_conv.2.bias"] = w, b
    sd["desc_head.sf_conv.weight"] = conv(dim, dim, 1, gain=1.7)
    sd["desc_head.agg_weights"] = rng.uniform(0, 1, (AL["M"], dim, dim)).astype(np.float32)
    if desc_centered:
        sd["desc_head.agg_weights"] = sd["desc_head.agg_weights"] - np.float32(0.5)
    return sd


BN_EPS = 1e-5


def _bn_affine(sd, p):
    """Inference BatchNorm as torch evaluates it: y = x * alpha + beta with
    alpha = weight / sqrt(running_var + eps), beta = bias - running_mean * alpha (fp32)."""
    inv = np.float32(1.0) / np.sqrt(sd[p + ".running_var"].astype(np.float32) + np.float32(BN_EPS))
    alpha = (sd[p + ".weight"] * inv).astype(np.float32)
    beta = (sd[p + ".bias"] - sd[p + ".running_mean"] * alpha).astype(np.float32)
    return alpha, beta


def _cito(w):
    """conv weight [co][ci][kh][kw] -> [ci][tap][co] (wave-uniform scalar loads per (ci, tap))."""
    co, ci, kh, kw = w.shape
    return np.ascontiguousarray(w.reshape(co, ci, kh * kw).transpose(1, 2, 0))


def aliked_order(sd):
    out = []

    def conv(p, bnp):
        a, b = _bn_affine(sd, bnp)
        out.extend([(p + ".weight", _cito(sd[p + ".weight"])), (bnp + ".alpha", a), (bnp + ".beta", b)])

    def dcn(p, bnp):
        a, b = _bn_affine(sd, bnp)
        out.extend([(p + ".offset_conv.weight", _cito(sd[p + ".offset_conv.weight"])),
                    (p + ".offset_conv.bias", sd[p + ".offset_conv.bias"]),
                    (p + ".regular_conv.weight", _cito(sd[p + ".regular_conv.weight"])),
                    (bnp + ".alpha", a), (bnp + ".beta", b)])

    def down(p):
        w = sd[p + ".downsample.weight"]
        bias = sd.get(p + ".downsample.bias")
        if bias is None:
            bias = np.zeros(w.shape[0], np.float32)
        out.extend([(p + ".downsample.weight", np.ascontiguousarray(w[:, :, 0, 0].T)), (p + ".downsample.bias", bias)])

    conv("block1.conv1", "block1.bn1"); conv("block1.conv2", "block1.bn2")
    conv("block2.conv1", "block2.bn1"); conv("block2.conv2", "block2.bn2"); down("block2")
    dcn("block3.conv1", "block3.bn1"); dcn("block3.conv2", "block3.bn2"); down("block3")
    dcn("block4.conv1", "block4.bn1"); dcn("block4.conv2", "block4.bn2"); down("block4")
    for i in range(1, 5):
        out.append((f"conv{i}.weight", np.ascontiguousarray(sd[f"conv{i}.weight"][:, :, 0, 0].T)))
    out.append(("score_head.0.weight", np.ascontiguousarray(sd["score_head.0.weight"][:, :, 0, 0].T)))
    for k in (2, 4, 6):
        out.append((f"score_head.{k}.weight", _cito(sd[f"score_head.{k}.weight"])))
    out.append(("desc_head.offset_conv.0.weight", sd["desc_head.offset_conv.0.weight"].reshape(32, -1)))
    out.append(("desc_head.offset_conv.0.bias", sd["desc_head.offset_conv.0.bias"]))
    out.append(("desc_head.offset_conv.2.weight", sd["desc_head.offset_conv.2.weight"][:, :, 0, 0]))
    out.append(("desc_head.offset_conv.2.bias", sd["desc_head.offset_conv.2.bias"]))
    out.append(("desc_head.sf_conv.weight", sd["desc_head.sf_conv.weight"][:, :, 0, 0]))
    agg = sd["desc_head.agg_weights"]                       # [p][c][d] -> [d][p*128 + c]
    out.append(("desc_head.agg_weights^T", np.ascontiguousarray(agg.reshape(-1, agg.shape[2]).T)))
    return out


def pack_aliked(sd) -> np.ndarray:
    sd = to_numpy_state_dict(sd)
    return _pad_cat([a for _, a in aliked_order(sd)])
