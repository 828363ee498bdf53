"""What runs BESIDE the extractor streams in the concurrency experiments (scripts/diag_agg_rnorm.py, scripts/stress_aliked_repeat.py):
`make_aggressor(spec, nat, W, ROOT)` -> (context, enqueue function).  spec: none | copy | lightglue[:opt,opt] | synthetic:<kind>[:launches:blocks:iters]
(profiles/r06_aggregate_rnorm_diagnosis.md)."""
import importlib


def make_aggressor(AGGR, nat, W, ROOT):
    aggr_ctx = nat.Context(0)
    aggressor = lambda: None
    if AGGR == "copy":                                             # HBM traffic and nothing else: 256 MB device-to-device copies
        NB = 256 << 20
        a_src, a_dst = aggr_ctx.malloc(NB), aggr_ctx.malloc(NB)
        aggressor = lambda: [aggr_ctx.d2d_async(a_dst, a_src, NB) for _ in range(6)]
    elif AGGR.startswith("lightglue"):                             # MFMA / transcendental / LDS-DMA kernels of another model
        import lg_inputs
        LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
        lg = LG(W.random_lightglue_state_dict(0), max_kpts=1024, ctx=aggr_ctx)
        # "lightglue:<opt>[,<opt>]": f32 (exact-fp32 kernels: no fp16 planes, no LDS-DMA rings, no assembly attention), ring (the 64-row
        # ring GEMMs instead of the 128 x 128 projections + fused FFN), noasm (the 4-wave HIP attention kernel), layers<N>
        for opt in (AGGR.split(":")[1].split(",") if ":" in AGGR else []):
            if opt == "f32":
                lg.set_precision("f32")
            elif opt == "ring":
                lg.debug_big_gemm(0)
            elif opt == "big":
                lg.debug_big_gemm(1)
            elif opt == "noasm":
                lg.debug_key_split(-1)
            elif opt.startswith("layers"):
                lg.debug_layers(int(opt[6:]))
        k0, d0, k1, d1 = lg_inputs.make_pair(1024, 1024, seed=1)
        dev = [aggr_ctx.upload(a) for a in (k0, d0, k1, d1)]
        o_ij, o_sc, o_info = aggr_ctx.malloc(1024 * 8), aggr_ctx.malloc(1024 * 4), aggr_ctx.malloc(16)
        aggressor = lambda: [lg.match_dev(dev[0], dev[1], 1024, dev[2], dev[3], 1024, o_ij, o_sc, o_info, min_conf=0.1) for _ in range(3)]
    elif AGGR.startswith("synthetic:"):                            # scripts/ubench/aggressors.hip: ONE property each
        import ctypes
        KIND = {"trans": 1, "mfma": 2, "pk": 3, "valu": 4, "lds": 5, "gather": 6, "store": 7, "scalar": 8, "ldsdma": 9,
                "mixlo": 10, "mixhi": 11, "sdwa": 12, "cvtpk": 13, "perm": 14, "permswap": 15, "bitop3": 16, "mov64": 17, "max3": 18, "pkmul_hi10": 19, "pkmul_01": 20, "pkfma_hi101": 21, "mfma16": 22, "cvtf16": 23, "fmamk": 24, "bfi": 25, "cmpabs": 26, "lshladd64": 27, "exp": 28, "pkmul": 30, "cvt_f16": 31, "shl64": 32, "insn_all": 40,
                "mfma_16x16x32_f16": 50, "mfma_32x32x16_bf16": 51, "mfma_16x16x32_bf16": 52, "mfma_32x32x8_f16": 53, "mfma_16x16x16_f16": 54, "mfma_32x32x64_f8f6f4": 55, "mfma_32x32x2_f32": 56, "mfma_32x32x32_i8": 57, "mfma_32x32x16_fp8": 58, "mfma_16x16x128_f8f6f4": 59}[AGGR.split(":")[1]]
        so = ROOT / "scripts" / "ubench" / "libaggr.so"
        if not so.exists():
            import subprocess
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", str(so),
                            str(ROOT / "scripts" / "ubench" / "aggressors.hip")], check=True)
        A = ctypes.CDLL(str(so))
        A.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        # (kinds >= 10: one instruction each in the form lg_attention_p_kernel uses it, four per loop iteration)
        ITERS = {1: 6000, 2: 3000, 3: 12000, 4: 24000, 5: 6000, 6: 3000, 7: 3000, 8: 6000, 9: 3000, 22: 2000, 40: 1000}.get(KIND, 2000 if KIND >= 50 else 12000)
        # "synthetic:<kind>": 3 long launches of 1024 workgroups per two extractor calls; "synthetic:<kind>:<launches>:<blocks>:<iters>":
        # many short ones (kernel BOUNDARIES of another queue - dispatch-time cache invalidates, wave launches - beside the extractor)
        spec = AGGR.split(":")
        LAUNCHES, BLOCKS, ITERS = (int(spec[2]), int(spec[3]), int(spec[4])) if len(spec) == 5 else (3, 1024, ITERS)

        def aggressor():
            for _ in range(LAUNCHES):
                assert A.aggr_launch(KIND, ctypes.c_void_p(int(aggr_ctx.stream)), BLOCKS, ITERS) == 0
    elif AGGR.startswith("aliked"):                                # "aliked[:<instances>]": extractor streams (batched ALIKED calls of two frames)
        import numpy as np                                             # noqa: F401
        import frames
        AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
        n_inst = int(AGGR.split(":")[1]) if ":" in AGGR else 2
        K, H, Wd = 384, 376, 1241
        sd = W.random_aliked_state_dict(0)
        actx = [aggr_ctx] + [nat.Context(0) for _ in range(n_inst - 1)]
        dets = [AL(sd, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=c, max_frames=2) for c in actx]
        imgs = [[c.upload(frames.structured_frame(2 * j + f, h=H, w=Wd)) for f in range(2)] for j, c in enumerate(actx)]
        outs = [[dict(xy=c.malloc(K * 8), desc=c.malloc(K * 512), sc=c.malloc(K * 4), n=c.malloc(64)) for _ in range(2)] for c in actx]

        def aggressor():
            for _ in range(2):
                for j, d in enumerate(dets):
                    o = outs[j]
                    d.extract_batch_dev(imgs[j], H, Wd, 3, [x["xy"] for x in o], [x["desc"] for x in o], [x["sc"] for x in o], [x["n"] for x in o])
        sync_all = lambda: [c.sync() for c in actx]
        aggr_ctx.sync_all = sync_all
    return aggr_ctx, aggressor
