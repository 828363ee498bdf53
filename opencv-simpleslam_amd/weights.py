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
def random_lightglue_state_dict(seed=0, match_gain=1.0, conf_bias=0.0, match_bias=0.0):
    """Seeded random init with upstream key names/shapes.
    nn.Linear-style U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm (1, 0).
    `match_gain` scales log_assignment.final_proj (sharper assignments),
    `conf_bias` / `match_bias` shift the token-confidence / matchability
    logits (to drive early stopping and pruning in tests)."""
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
            sd[f"{p}.ffn.3.weight"], sd[f"{p}.ffn.3.bias"] = lin(D, 2 * D)
    for i in range(LG_LAYERS):
        p = f"log_assignment.{i}"
        w, b = lin(D, D, gain=match_gain)
        sd[p + ".final_proj.weight"], sd[p + ".final_proj.bias"] = w, b
        w, b = lin(1, D)
        sd[p + ".matchability.weight"], sd[p + ".matchability.bias"] = w, b + np.float32(match_bias)
    for i in range(LG_LAYERS - 1):
        w, b = lin(1, D)
        sd[f"token_confidence.{i}.token.0.weight"] = w
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


def lightglue_order(sd):
    """(name, array) list in blob order.  Pure re-indexing / concatenation of
    upstream tensors - no arithmetic, so numerics are untouched."""
    perm = _qkv_row_perm()
    out = [("input_proj.weight", sd["input_proj.weight"]), ("input_proj.bias", sd["input_proj.bias"]),
           ("posenc.Wr.weight", sd["posenc.Wr.weight"])]
    for i in range(LG_LAYERS):
        p = f"transformers.{i}.self_attn"
        out += [(p + ".Wqkv.weight[perm]", sd[p + ".Wqkv.weight"][perm]),
                (p + ".Wqkv.bias[perm]", sd[p + ".Wqkv.bias"][perm]),
                (p + ".out_proj.weight", sd[p + ".out_proj.weight"]),
                (p + ".out_proj.bias", sd[p + ".out_proj.bias"])]
        out += [(p + f".ffn.{j}.{w}", sd[p + f".ffn.{j}.{w}"]) for j in (0, 1, 3) for w in ("weight", "bias")]
        p = f"transformers.{i}.cross_attn"
        out += [(p + ".to_qk|to_v.weight", np.concatenate([sd[p + ".to_qk.weight"], sd[p + ".to_v.weight"]], 0)),
                (p + ".to_qk|to_v.bias", np.concatenate([sd[p + ".to_qk.bias"], sd[p + ".to_v.bias"]], 0)),
                (p + ".to_out.weight", sd[p + ".to_out.weight"]),
                (p + ".to_out.bias", sd[p + ".to_out.bias"])]
        out += [(p + f".ffn.{j}.{w}", sd[p + f".ffn.{j}.{w}"]) for j in (0, 1, 3) for w in ("weight", "bias")]
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
