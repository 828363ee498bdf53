"""WHAT are the wrong 1 / ||F|| values of al_aggregate_kernel's unstable code shapes?

Run on a library built with a failing variant (scripts/diag_agg_rnorm.sh: -DAL_AGG_FAST_SELU=2).  The concurrency stress of
scripts/stress_aliked_repeat.py (3 extractor instances x 2 frames x 4 un-synchronised calls per repeat); for every repeat in
which frame 0's `rnorm` map of an instance differs from the first run, for each run of differing pixels:

  * the wrong and the correct values, n2 = 1 / rnorm^2 of both and their difference;
  * n2 rebuilt on the host (float64) from the stage buffers the kernel read - g1 (channel-last) and rows 8..12 of the three
    `pre*` maps (S, H, V, D1, D2: the quadratic form of agg_level) - which validates the replica against the CORRECT value;
  * hypotheses, each evaluated on all pixels of the run and reported when it reproduces the WRONG n2 to 1e-5 relative:
      store / slot   wrong == the value the OTHER frame slot of the call has at this pixel; == this frame's correct value
                     at a shifted pixel (+-1, +-16, +-64, +-Wp)
      gather         ONE of the 30 gathers of the quadratic form (3 levels x {S00, S01, S10, S11, H00, H10, V00, V01, D1, D2})
                     returned a neighbouring element (+-1, +- one map row), the element of the other frame slot at the same
                     offset, or zero
      term           one whole level's term, or the g1 term, taken from the other frame slot
    and, when nothing fits, the residual delta against each level's term size (is it "a few per cent of one term"?).

usage: diag_agg_rnorm.py [repeats=120]"""
import hashlib
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
import frames                                                    # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 3              # extractor instances (streams)
AGGR = sys.argv[3] if len(sys.argv) > 3 else "none"            # what ELSE runs beside them: none | copy (a device-to-device copy stream) | lightglue
F = 2
pkg = importlib.import_module("opencv-simpleslam_amd")
W = importlib.import_module("opencv-simpleslam_amd.weights")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
nat = pkg._native
K, H, Wd = 384, 376, 1241
sd = W.random_aliked_state_dict(0)
ctxs = [nat.Context(0) for _ in range(NI)]
dets = [AL(sd, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=c, max_frames=F) for c in ctxs]
imgs = [frames.structured_frame(i, h=H, w=Wd) for i in range(F * NI)]
d_img = [[ctxs[j].upload(imgs[j * F + f]) for f in range(F)] for j in range(NI)]
out = [[dict(xy=ctxs[j].malloc(K * 8), desc=ctxs[j].malloc(K * 512), sc=ctxs[j].malloc(K * 4), n=ctxs[j].malloc(64)) for f in range(F)]
       for j in range(NI)]


def enqueue(j):
    o = out[j]
    dets[j].extract_batch_dev(d_img[j], H, Wd, 3, [x["xy"] for x in o], [x["desc"] for x in o], [x["sc"] for x in o], [x["n"] for x in o])


def stage(det):
    d = det.debug_read(2, (8,), np.int32)
    Hp, Wp = int(d[2]), int(d[3])
    return dict(Hp=Hp, Wp=Wp, g1=det.debug_read(10, (Hp, Wp, 32)), rnorm=det.debug_read(11, (Hp, Wp)),
                pre2=det.debug_read(15, (13, Hp // 2, Wp // 2)), pre3=det.debug_read(16, (13, Hp // 8, Wp // 8)),
                pre4=det.debug_read(17, (13, Hp // 32, Wp // 32)))


# the stage buffers of every image as FRAME 0 of a quiet single-frame instance (what the other frame slot of a call holds)
quiet = AL(sd, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=nat.Context(0), max_frames=1)
by_img = []
for im in imgs:
    quiet.extract(im, K)
    by_img.append(stage(quiet))
quiet.close()

f32 = np.float32


def taps(y, xs, Hp, Wp, S):
    """up_tap of aliked_kernels.hip in float32, for row y and the columns xs -> offsets (4) and weights."""
    ih, iw = Hp // S, Wp // S
    sy, sx = f32(ih - 1) / f32(Hp - 1), f32(iw - 1) / f32(Wp - 1)
    fy, fx = f32(sy * f32(y)), (sx * xs.astype(f32)).astype(f32)
    y0, x0 = int(fy), fx.astype(np.int64)
    y1, x1 = y0 + (y0 < ih - 1), x0 + (x0 < iw - 1)
    ly, lx = f32(fy - f32(y0)), (fx - x0.astype(f32)).astype(f32)
    hy, hx = f32(1) - ly, f32(1) - lx
    return dict(o00=y0 * iw + x0, o01=y0 * iw + x1, o10=y1 * iw + x0, o11=y1 * iw + x1, hx=hx.astype(np.float64), lx=lx.astype(np.float64),
                hy=float(hy), ly=float(ly), iw=iw, n=ih * iw)


GATHERS = ("S00", "S01", "S10", "S11", "H00", "H10", "V00", "V01", "D1", "D2")


def level_gathers(pre, t):
    """The ten gathers of agg_level's quadratic form -> dict name -> (map row 8..12, offsets, coefficient in n2)."""
    w00, w01, w10, w11 = t["hy"] * t["hx"], t["hy"] * t["lx"], t["ly"] * t["hx"], t["ly"] * t["lx"]
    return {"S00": (8, t["o00"], w00 * w00), "S01": (8, t["o01"], w01 * w01), "S10": (8, t["o10"], w10 * w10), "S11": (8, t["o11"], w11 * w11),
            "H00": (9, t["o00"], 2 * w00 * w01), "H10": (9, t["o10"], 2 * w10 * w11), "V00": (10, t["o00"], 2 * w00 * w10),
            "V01": (10, t["o01"], 2 * w01 * w11), "D1": (11, t["o00"], 2 * w00 * w11), "D2": (12, t["o00"], 2 * w01 * w10)}


def n2_terms(st, y, xs):
    """-> (g1 term, [level terms], [(taps, gathers)]) in float64 for row y, columns xs."""
    g1 = (st["g1"][y, xs].astype(np.float64) ** 2).sum(-1)
    lv, info = [], []
    for S, key in ((2, "pre2"), (8, "pre3"), (32, "pre4")):
        t = taps(y, xs, st["Hp"], st["Wp"], S)
        g = level_gathers(st[key], t)
        flat = st[key].reshape(13, -1).astype(np.float64)
        lv.append(sum(c * flat[row][off] for row, off, c in g.values()))
        info.append((t, g, flat))
    return g1, lv, info


def analyse(j, ref_st, cur):
    Hp, Wp = ref_st["Hp"], ref_st["Wp"]
    other = by_img[j * F + 1]                                  # what frame slot 1 of this instance's call holds
    ys, xs_all = np.nonzero(cur != ref_st["rnorm"])
    for y in sorted(set(ys.tolist())):
        cols = np.sort(xs_all[ys == y])
        runs = np.split(cols, np.flatnonzero(np.diff(cols) > 1) + 1)
        for xs in runs:
            lo, hi = int(xs[0]), int(xs[-1])
            span = np.arange(lo & ~15, (hi | 15) + 1)          # the 16-lane groups the run touches
            bad = np.isin(span, xs)
            wrong, good = cur[y, span].astype(np.float64), ref_st["rnorm"][y, span].astype(np.float64)
            n2w, n2g = 1.0 / wrong ** 2, 1.0 / good ** 2
            print(f"  row {y} cols {lo}..{hi} ({len(xs)} px; 16-lane groups {lo >> 4}..{hi >> 4}; byte offset of the first in rnorm: "
                  f"{(y * Wp + lo) * 4} = {((y * Wp + lo) * 4) % 128} mod 128)")
            print("    wrong  : " + " ".join(f"{v:.6f}" for v in wrong[bad]))
            print("    correct: " + " ".join(f"{v:.6f}" for v in good[bad]))
            print("    rel err: " + " ".join(f"{(a / b - 1) * 100:+.3f}%" for a, b in zip(wrong[bad], good[bad])))
            g1, lv, info = n2_terms(ref_st, y, span)
            host = g1 + sum(lv)
            print(f"    host replica of n2 vs the correct value: max rel {np.abs(host / n2g - 1).max():.2e}; terms (mean) g1 {g1.mean():.4g} "
                  f"L2 {lv[0].mean():.4g} L3 {lv[1].mean():.4g} L4 {lv[2].mean():.4g}; delta n2 (wrong - correct) mean {np.mean((n2w - n2g)[bad]):+.4g} "
                  f"= {np.mean((n2w / n2g - 1)[bad]) * 100:+.3f} % of n2")
            hits = []

            def test(name, cand):
                ok = np.abs(cand[bad] / n2w[bad] - 1) < 1e-5
                if ok.sum() >= max(1, int(0.8 * bad.sum())):
                    hits.append(f"{name} ({int(ok.sum())}/{int(bad.sum())} px)")
            # store / slot hypotheses
            test("rnorm of the OTHER frame slot at this pixel", 1.0 / other["rnorm"][y, span].astype(np.float64) ** 2)
            for d in (-1, 1, -16, 16, -64, 64, -Wp, Wp):
                idx = y * Wp + span + d
                if idx.min() >= 0 and idx.max() < Hp * Wp:
                    test(f"this frame's correct rnorm at pixel {d:+d}", 1.0 / ref_st["rnorm"].reshape(-1)[idx].astype(np.float64) ** 2)
            # term hypotheses
            og1, olv, oinfo = n2_terms(other, y, span)
            test("g1 term of the other frame slot", host - g1 + og1)
            for L in range(3):
                test(f"level {('/2', '/8', '/32')[L]} term of the other frame slot", host - lv[L] + olv[L])
                test(f"level {('/2', '/8', '/32')[L]} term missing", host - lv[L])
            test("every term of the other frame slot", og1 + sum(olv))
            # gather hypotheses
            for L in range(3):
                t, g, flat = info[L]
                oflat = oinfo[L][2]
                for name in GATHERS:
                    row, off, c = g[name]
                    base = c * flat[row][off]
                    for dname, o2 in (("-1", off - 1), ("+1", off + 1), ("-row", off - t["iw"]), ("+row", off + t["iw"])):
                        o2c = np.clip(o2, 0, t["n"] - 1)
                        test(f"gather {name} of level {('/2', '/8', '/32')[L]} returned element {dname}", host - base + c * flat[row][o2c])
                    test(f"gather {name} of level {('/2', '/8', '/32')[L]} returned the other frame slot's element", host - base + c * oflat[row][off])
                    test(f"gather {name} of level {('/2', '/8', '/32')[L]} returned 0", host - base)
                    for r2 in range(8, 13):
                        if r2 != row:
                            test(f"gather {name} of level {('/2', '/8', '/32')[L]} returned map row {r2} (of S H V D1 D2 = 8..12) at its offset",
                                 host - base + c * flat[r2][off])
            print("    hypotheses reproducing the wrong n2: " + ("; ".join(hits) if hits else "NONE of the tested ones"))
            if not hits:
                d = (n2w - n2g)[bad]
                print("    delta n2 per pixel: " + " ".join(f"{v:+.4g}" for v in d))
                for L in range(3):
                    t, g, flat = info[L]
                    sizes = {name: float(np.mean(np.abs(g[name][2] * flat[g[name][0]][g[name][1]])[bad])) for name in GATHERS}
                    print(f"    level {('/2', '/8', '/32')[L]} gather term sizes: " + " ".join(f"{k} {v:.3g}" for k, v in sizes.items()))


from aggressor_util import make_aggressor                     # noqa: E402
aggr_ctx, aggressor = make_aggressor(AGGR, nat, W, ROOT)
print(f"{NI} extractor instance(s) x {F} frames x 4 un-synchronised calls per repeat; beside them: {AGGR}", flush=True)
for j in range(NI):
    enqueue(j)
for c in ctxs:
    c.sync()
ref_st = [stage(dets[j]) for j in range(NI)]
for j in range(NI):                                            # the quiet single-frame run must agree with the first stressed run
    q = by_img[j * F]
    same = {k: bool(np.array_equal(q[k], ref_st[j][k])) for k in ("g1", "rnorm", "pre2", "pre3", "pre4")}
    print(f"instance {j}: first run vs the quiet single-frame run of the same image: {same}", flush=True)
ref_hash = [hashlib.sha1(ref_st[j]["rnorm"].tobytes()).hexdigest() for j in range(NI)]
events = 0
for r in range(reps):
    for k in range(4):
        if k % 2 == 0:
            aggressor()
        for j in range(NI):
            enqueue(j)
    for c in ctxs:
        c.sync()
    aggr_ctx.sync()
    for j in range(NI):
        d = dets[j].debug_read(11, (ref_st[j]["Hp"], ref_st[j]["Wp"]))
        if hashlib.sha1(d.tobytes()).hexdigest() != ref_hash[j]:
            events += 1
            cur = stage(dets[j])
            others = [k for k in ("g1", "pre2", "pre3", "pre4") if not np.array_equal(cur[k], ref_st[j][k])]
            print(f"repeat {r}: instance {j}: rnorm differs in {int((d != ref_st[j]['rnorm']).sum())} pixels; other stage buffers differing: {others or 'none'}", flush=True)
            analyse(j, ref_st[j], d)
try:                                                           # AL_AGG_LOAD_NOP=3 builds: the packed-versus-scalar record of agg_level
    w = dets[0].debug_read(99, (64,), np.uint32)
    names = ("c0*H00", "c1*H10", "c2*V00", "c3*V01", "c4*D1", "c5*D2")
    print(f"in-kernel check (the six cross products of the quadratic form, compiler's packed form against v_mul_f32 on the SAME registers): "
          f"{int(w[0])} lanes disagreed; products that ever differed: {[n for i, n in enumerate(names) if w[1] >> i & 1]}; "
          f"16-lane groups of the wave: {[q for q in range(4) if w[2] >> q & 1]}")
    for k in range(min(int(w[0]), 8)):
        o = w[8 + 6 * k: 14 + 6 * k]
        f = o[1:5].view(np.float32)
        which = [n for i, n in enumerate(names) if int(o[0]) & 63 >> i & 1]
        print(f"    lane {int(o[0]) >> 8 & 63} (level map of {int(o[0]) >> 16} px, block {int(o[5]) & 255},{int(o[5]) >> 8 & 0xffff}, frame {int(o[5]) >> 24}): {which}: "
              f"packed {f[0]!r} scalar {f[1]!r} = coefficient {f[2]!r} x gathered {f[3]!r} (exact product {np.float32(f[2]) * np.float32(f[3])!r})")
except Exception as e:                                         # noqa: BLE001
    print(f"(no in-kernel record in this build: {type(e).__name__})")
print(f"{reps} repeats: {events} events", flush=True)
