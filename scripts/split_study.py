"""SURVEY 7.2 / VERDICT r03 item 2: what do cheaper operand forms of the split-precision products cost in matches?

The product evaluates an fp32 product a.b as hi.hi + (hi.lo + lo.hi) 2^-11 on the f16 matrix pipe (3 MFMA).  This script
drops cross terms one product family at a time (sslam_lightglue_debug_split_form: the dropped term's low plane is read
from an all-zero plane, bit-identical to not issuing the MFMA; P's low plane is dropped in the 4-wave kernel) and
reports, against the torch-CPU oracle on the same inputs:
  * flipped matches (symmetric difference of the (i, j) sets) per 1e5 oracle matches, and pairs with any flip,
  * max |score difference|, and max |dx| of the final token state against an fp64 evaluation of the oracle.
Inputs: the parity suite's cases, the six-seed 1024 x 960 set, the C2 size, and the same seeds with the q / k
projections scaled x2 / x4 (logits x4 / x16: peaked attention, which random-init weights do not produce by themselves).

    python scripts/split_study.py [--quick] > profiles/r04_split_study.md
"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lg_inputs
from oracle import lightglue_ref as R

W = importlib.import_module("opencv-simpleslam_amd.weights")
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP

FORMS = [
    (0x00, "product: 3 terms everywhere"),
    (0x01, "logits: K one plane (kh.qh + kh.ql)"),
    (0x02, "logits: Q one plane (kh.qh + kl.qh)"),
    (0x04, "context: P one plane (vh.ph + vl.ph, row sum over rounded P)"),
    (0x08, "context: V one plane (vh.ph + vh.pl)"),
    (0x05, "attention 2 + 2: K one plane + P one plane"),
    (0x10, "projections: activation low plane dropped (weight low kept)"),
    (0x40, "projections: weight low plane dropped"),
    (0x20, "FFN: activation low plane dropped (weight low kept)"),
    (0x80, "FFN: weight low plane dropped"),
    (0x30, "all W.x: activation low plane dropped"),
    (0x35, "everything 2 terms (K, P, activations one plane)"),
]


def sharpen(sd, gamma):
    """q / k projections x gamma (logits x gamma^2): peaked attention rows."""
    sd = dict(sd)
    for i in range(9):
        w, b = sd[f"transformers.{i}.self_attn.Wqkv.weight"].copy(), sd[f"transformers.{i}.self_attn.Wqkv.bias"].copy()
        rows = np.arange(w.shape[0]) % 3 != 2          # upstream layout: column c = h 192 + d 3 + s, s = q, k, v
        w[rows] *= gamma; b[rows] *= gamma
        sd[f"transformers.{i}.self_attn.Wqkv.weight"], sd[f"transformers.{i}.self_attn.Wqkv.bias"] = w, b
        sd[f"transformers.{i}.cross_attn.to_qk.weight"] = sd[f"transformers.{i}.cross_attn.to_qk.weight"] * gamma
        sd[f"transformers.{i}.cross_attn.to_qk.bias"] = sd[f"transformers.{i}.cross_attn.to_qk.bias"] * gamma
    return sd


def fwd64(sd, k0, d0, k1, d1, conf):
    import inspect, types
    old = torch.get_default_dtype(); torch.set_default_dtype(torch.float64)
    try:
        src = inspect.getsource(R).replace("torch.float32", "torch.float64")
        mod = types.ModuleType("lg64"); exec(compile(src, "lg64", "exec"), mod.__dict__)
        return mod.lightglue_forward(sd, k0.astype(np.float64), d0.astype(np.float64), k1.astype(np.float64),
                                     d1.astype(np.float64), conf, return_debug=True)
    finally:
        torch.set_default_dtype(old)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    NOSTOP = {"depth_confidence": -1, "width_confidence": -1}
    groups = []          # (name, [(sd, conf, (k0, d0, k1, d1), min_conf, lg kwargs, want fp64)])

    def case(seed_w, m, n, seed_in, gamma=None, conf=None, gain=4.0, bias=3.0, want64=False):
        sd = W.random_lightglue_state_dict(seed_w, match_gain=gain, match_bias=bias)
        if gamma:
            sd = sharpen(sd, gamma)
        return (sd, conf, lg_inputs.make_pair(m, n, seed=seed_in), 0.0, want64)

    groups.append(("parity suite sizes (weights seed 1; 512x512, 300x417, 129x128, 64x33)",
                   [case(1, m, n, m + n) for m, n in ((512, 512), (300, 417), (129, 128), (64, 33))]))
    groups.append(("six seeds, 1024 x 960 (tests/test_lightglue_gpu.py::test_index_parity_over_seeds)",
                   [case(s, 1024, 960, s) for s in (range(101, 103) if args.quick else range(101, 107))]))
    groups.append(("token state, 512 x 512, no early stop (test_token_state_per_layer_is_fp32_grade's case)",
                   [case(5, 512, 512, 21, conf=NOSTOP, want64=True)]))
    if not args.quick:
        groups.append(("C2 size, 2048 x 2048 (weights seeds 1, 2)", [case(1, 2048, 2048, 7), case(2, 2048, 2048, 8)]))
    groups.append(("peaked attention: q / k projections x2 (logits x4), seeds 101-103, 1024 x 960",
                   [case(s, 1024, 960, s, gamma=2.0, want64=(s == 101)) for s in (101, 102, 103)]))
    groups.append(("peaked attention: q / k projections x4 (logits x16), seeds 101-103, 1024 x 960",
                   [case(s, 1024, 960, s, gamma=4.0, want64=(s == 101)) for s in (101, 102, 103)]))

    print("# r04 split study: cheaper operand forms of the split-precision products\n")
    print("Generated by `scripts/split_study.py` on the GPU box (HIP path = ring linears + attention kernels with the named "
          "cross terms dropped; reference = `oracle/lightglue_ref.py` on torch-CPU fp32, token state also against its fp64 evaluation).")
    print("`flips` = |HIP matches (i, j) symmetric-difference oracle matches| at min_conf 0 (everything above LightGlue's 0.1 filter); "
          "`per 1e5` = flips per 1e5 oracle matches; `stop` = pairs whose early-stop layer differs from the oracle's.\n")
    t_all = time.time()
    for gname, cases in groups:
        refs = []
        for sd, conf, (k0, d0, k1, d1), mc, want64 in cases:
            ref = R.lightglue_forward(sd, k0, d0, k1, d1, conf, return_debug=True)
            r64 = fwd64(sd, k0, d0, k1, d1, conf) if want64 else None
            refs.append((ref, r64))
        n_or = sum(len(r["matches"]) for r, _ in refs)
        print(f"## {gname}\n")
        print(f"{len(cases)} pairs, {n_or} oracle matches. fp32 oracle vs its fp64 evaluation: "
              + ", ".join(f"max|dx| {np.abs(torch.cat([r['debug']['x_out0'], r['debug']['x_out1']]).numpy() - torch.cat([q['debug']['x_out0'], q['debug']['x_out1']]).numpy()).max():.2e}"
                          for r, q in refs if q is not None and r['debug']['x_out0'].shape == q['debug']['x_out0'].shape) + "\n")
        print("| mask | form | flips | per 1e5 | pairs with a flip | stop | max abs score diff | max abs dx vs fp64 | max abs dx vs fp32 oracle |")
        print("|---|---|---|---|---|---|---|---|---|")
        lgs = {}
        for mask, fname in FORMS:
            flips = bad_pairs = stops = 0
            dsc = dx64 = dx32 = 0.0
            for (sd, conf, (k0, d0, k1, d1), mc, want64), (ref, r64) in zip(cases, refs):
                cap = max(len(k0), len(k1))
                key = (id(sd), cap)
                if key not in lgs:
                    kw = {} if conf is None else dict(depth_confidence=-1.0, width_confidence=-1.0)
                    lgs[key] = LG(sd, max_kpts=cap, **kw)
                    lgs[key].debug_big_gemm(0)            # ring linears: every operand plane (hidden included) lives in HBM
                lg = lgs[key]
                lg.debug_split_form(mask)
                ij, sc, stop = lg.match(k0, d0, k1, d1, min_conf=mc)
                rij = ref["matches"].numpy(); rsc = ref["scores"].numpy()
                a = {(int(i), int(j)) for i, j in ij}; b = {(int(i), int(j)) for i, j in rij}
                f = len(a ^ b)
                flips += f; bad_pairs += f > 0; stops += int(stop != ref["stop"])
                common = {p: s for p, s in zip(map(tuple, rij.tolist()), rsc)}
                d = [abs(float(s) - float(common[tuple(p)])) for p, s in zip(ij.tolist(), sc) if tuple(p) in common]
                dsc = max(dsc, max(d) if d else 0.0)
                Kc = lg.capacity
                x = lg.debug_read(0, (2, Kc, 256))
                n0, n1 = ref["debug"]["x_out0"].shape[0], ref["debug"]["x_out1"].shape[0]
                if stop == ref["stop"] and n0 == len(k0):
                    xg = np.concatenate([x[0, :n0], x[1, :n1]])
                    xr = torch.cat([ref["debug"]["x_out0"], ref["debug"]["x_out1"]]).numpy()
                    dx32 = max(dx32, float(np.abs(xg - xr).max()))
                    if r64 is not None and r64["debug"]["x_out0"].shape[0] == n0:
                        x64 = torch.cat([r64["debug"]["x_out0"], r64["debug"]["x_out1"]]).numpy()
                        dx64 = max(dx64, float(np.abs(xg - x64).max()))
            print(f"| 0x{mask:02x} | {fname} | {flips} | {1e5 * flips / max(n_or, 1):.1f} | {bad_pairs} / {len(cases)} | {stops} | "
                  f"{dsc:.2e} | {dx64:.2e} | {dx32:.2e} |", flush=True)
        for lg in lgs.values():
            lg.close()
        print()
    print(f"(run time {time.time() - t_all:.0f} s)")


if __name__ == "__main__":
    main()
