#!/usr/bin/env python3
"""bench.py - frames/s of the ALIKED + LightGlue hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Metric (BASELINE.json): frames/sec ALIKED+LightGlue @1241x376.  Workload =
config C2/C4: synthetic 1241x376x3 uint8 frames (white
noise, default_rng(1234 + frame)), ALIKED-n16 -> 2048 keypoints per frame,
LightGlue(features='aliked') on every (t-1, t) pair, min_conf 0.7, random-init
weights of the upstream architecture (no network for checkpoints).

A "step" = one round of the frame-sharded pipeline: every rank extracts its
B frames and matches each against its predecessor (B extracts + B matches per
rank).  Inputs are resident in HBM before the timed region.  One process per
GPU; for N > 1 launch with torch.distributed.run (RCCL over xGMI collates the
features of each round into the shared map, frame_shard.py).

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     - the dominant kernel (LightGlue attention, fp32 MFMA), timed
                 live with HIP events around every launch in the timed region
  cpu_baseline - the torch-CPU oracle (kind "port") on a bounded sample
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

H_IMG, W_IMG, C_IMG = 376, 1241, 3
MAX_KPTS = 2048
MIN_CONF = 0.7
FRAMES_PER_RANK = int(os.environ.get("SSLAM_BENCH_FRAMES", 24))     # frames per GPU per step
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak (spec, no sparsity)
F32_MFMA_PEAK_TFLOPS = 157.3           # same table: v_mfma_f32_32x32x2_f32, the exact-fp32 matrix-core rate
ALIKED_GFLOP_PER_FRAME = 8.9           # 6.56 dense conv + 2.34 SDDH at 2048 keypoints


def _pmc_traffic():
    """Per-launch HBM-side bytes of the attention kernel at the bench size, from profiles/ (None when absent)."""
    f = ROOT / "profiles" / "r01_attention_traffic.json"
    try:
        d = json.loads(f.read_text())
        return int(d["fetch_bytes_per_launch"]) + int(d["write_bytes_per_launch"])
    except Exception:
        return None


def lightglue_gflop(n, layers):
    """SURVEY 8(d) F(N, L): algorithmic FLOPs of one pair with M = N = n keypoints, L layers executed."""
    D, d_in = 256, 128
    blk_self = 6 * n * D * D + 4 * n * n * D + 2 * n * D * D + 8 * n * D * D + 4 * n * D * D
    blk_cross = 4 * n * D * D + 4 * n * n * D + 2 * n * D * D + 8 * n * D * D + 4 * n * D * D
    return (4 * n * d_in * D + layers * (2 * blk_self + 2 * blk_cross) + 4 * n * D * D + 2 * n * n * D) / 1e9


def noise_frame(idx):
    return np.random.default_rng(1234 + idx).integers(0, 256, (H_IMG, W_IMG, C_IMG), dtype=np.uint8)


_STRUCT_BASE = None


def structured_frame(idx):
    """SURVEY 8(d) second input: the noise low-pass filtered with a 9x9 box (stretched back to
    0..255) and translated 3 px per frame, so consecutive frames really overlap."""
    global _STRUCT_BASE
    if _STRUCT_BASE is None:
        from scipy.ndimage import uniform_filter
        base = np.random.default_rng(1234).integers(0, 256, (H_IMG, W_IMG + 512, C_IMG)).astype(np.float32)
        low = uniform_filter(base, size=(9, 9, 1), mode="reflect")
        lo, hi = low.min(), low.max()
        _STRUCT_BASE = np.clip((low - lo) * (255.0 / (hi - lo)), 0, 255).astype(np.uint8)
    off = (3 * idx) % 512
    return np.ascontiguousarray(_STRUCT_BASE[:, off:off + W_IMG])


def attention_flops(n0, n1):
    """Algorithmic FLOPs of one attention launch (both images, 4 heads x 64): SURVEY 8(d) counts
    4 N^2 D per image per block (QK^T + AV)."""
    return 2.0 * 256 * (n0 * n1 * 2) * 2


def cpu_baseline(max_frames=8, budget_s=20.0, threads=None):
    """Reference path restated on torch-CPU (oracle/), timed on this host's cores: a bounded
    sample of the same workload (stops after `budget_s` seconds or `max_frames` frames).
    torch intra-op threading stops scaling (and collapses) well before a 256-thread host is
    full on these small operators, so the thread count is capped at 16 and reported."""
    import torch
    from oracle import aliked_ref, lightglue_ref
    W = importlib.import_module("opencv-simpleslam_amd.weights")
    torch.set_num_threads(threads or min(os.cpu_count() or 1, 16))
    sd_a, sd_l = W.random_aliked_state_dict(0), W.random_lightglue_state_dict(0)
    prev = aliked_ref.aliked_extract(sd_a, noise_frame(0), MAX_KPTS)         # warm-up + first frame
    t0 = time.perf_counter()
    n_frames = 0
    for i in range(1, max_frames + 1):
        cur = aliked_ref.aliked_extract(sd_a, noise_frame(i), MAX_KPTS)
        lightglue_ref.reference_feature_matcher(sd_l, prev["keypoints"], cur["keypoints"],
                                                prev["descriptors"], cur["descriptors"], MIN_CONF)
        prev = cur
        n_frames += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n_frames / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n_frames} frames 1241x376 (extract + match t-1->t, 2048 kpts, 9 layers), "
                      f"torch-CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # the ROCm default, pinned: 8 streams on 4 hardware queues is the measured optimum
    # (scripts/sweep_queues.sh: 5 queues -30 %, 3 queues -11 %); must be set before HIP initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one process per GPU; SSLAM_DIST_BACKEND=gloo + fewer GPUs than ranks is a test-only mode that
    # exercises the N > 1 code path on a single-GPU box (ranks share device local_rank % n_gpus)
    backend = os.environ.get("SSLAM_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = importlib.import_module("opencv-simpleslam_amd")
    W = importlib.import_module("opencv-simpleslam_amd.weights")
    AlikedHIP = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
    LightGlueHIP = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
    fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")

    # extractor / matcher instances, one HIP stream each.  Streams are multiplexed onto
    # GPU_MAX_HW_QUEUES (= 4, pinned in main()) hardware queues: 1 + 7 streams on one GPU (566
    # frames/s vs 552 for 2 + 6; a ninth stream drops it to 490); with N > 1 one stream fewer,
    # RCCL brings its own
    N_EXT = int(os.environ.get("SSLAM_BENCH_NE", 1))
    N_MAT = int(os.environ.get("SSLAM_BENCH_NM", 7 if world == 1 else 6))
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        streams_e = [torch.cuda.Stream() for _ in range(N_EXT)]      # default priority: prioritised
        streams_m = [torch.cuda.Stream() for _ in range(N_MAT)]      # streams cost 40 % (measured)
        ctx_e = [pkg._native.Context(device_index, stream=st.cuda_stream) for st in streams_e]
        ctx_m = [pkg._native.Context(device_index, stream=st.cuda_stream) for st in streams_m]
        sd_a, sd_l = W.random_aliked_state_dict(0), W.random_lightglue_state_dict(0)
        dets = [AlikedHIP(sd_a, max_num_keypoints=MAX_KPTS, max_h=H_IMG, max_w=W_IMG, ctx=c) for c in ctx_e]
        mats = [LightGlueHIP(sd_l, max_kpts=MAX_KPTS, ctx=c) for c in ctx_m]
        plan = fs.ShardPlan(world, rank, FRAMES_PER_RANK)
        pipe = fs.FrameStreamPipeline(dets, mats, plan, MAX_KPTS, MIN_CONF, streams_e=streams_e,
                                      streams_m=streams_m)

        # synthetic stream, resident in HBM: a pool of rounds that the timed loop cycles through
        n_pool = 4
        pool = [torch.from_numpy(np.stack([noise_frame(f) for f in plan.frames(r)])).cuda(non_blocking=False)
                for r in range(n_pool)]

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for i in range(args.warmup):
            pipe.round(pool[i % n_pool], H_IMG, W_IMG, C_IMG)
        barrier()
        for mat in mats:
            mat.profile(True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            pipe.round(pool[(args.warmup + i) % n_pool], H_IMG, W_IMG, C_IMG)
        barrier()
        dt = time.perf_counter() - t0
        attn_ms, attn_n = 0.0, 0
        for mat in mats:
            mat.profile(False)
            ms_, n_ = mat.profile_read()
            attn_ms += ms_; attn_n += n_
        info = pipe.info.cpu().numpy()
        # Kernel-level figure for the roofline: the same launches replayed on ONE stream with the
        # other streams idle.  In the timed region up to 8 streams share the chip, so the HIP-event
        # bracket of a launch there also contains the time its blocks wait for CUs held by other
        # streams' kernels; the single-stream bracket is the kernel's own duration (it is what
        # rocprofv3 --kernel-trace reports for the kernel in either mode).
        torch.cuda.synchronize()
        m0 = mats[0]
        m0.profile(True)
        with torch.cuda.stream(streams_m[0]):
            for rep in range(4):
                m0.match_dev(pipe.xy[0], pipe.desc[0], MAX_KPTS, pipe.xy[1], pipe.desc[1], MAX_KPTS,
                             pipe.ij[1], pipe.msc[1], pipe.info[1], min_conf=MIN_CONF,
                             m_dev=pipe.count[0], n_dev=pipe.count[1])
        torch.cuda.synchronize()
        m0.profile(False)
        iso_ms, iso_n = m0.profile_read()

        # second input of SURVEY 8(d): the structured (low-pass, translating) stream, same pipeline
        spool = [torch.from_numpy(np.stack([structured_frame(f) for f in plan.frames(r)])).cuda()
                 for r in range(2)]
        for i in range(2):
            pipe.round(spool[i % 2], H_IMG, W_IMG, C_IMG)
        barrier()
        s_steps = max(2, args.steps // 2)
        ts0 = time.perf_counter()
        for i in range(s_steps):
            pipe.round(spool[i % 2], H_IMG, W_IMG, C_IMG)
        barrier()
        s_dt = time.perf_counter() - ts0
        s_info = pipe.info.cpu().numpy()

    t = torch.tensor([dt, s_dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, s_dt_max = float(t[0].item()), float(t[1].item())

    if rank == 0:
        frames_total = args.steps * plan.frames_per_round()
        n0, n1, stop = int(info[-1, 2]), int(info[-1, 3]), int(info[-1, 1])
        ach = attention_flops(n0, n1) / (iso_ms / max(iso_n, 1) * 1e-3) / 1e12 if iso_n else None
        out = {
            "metric": "frames/sec ALIKED+LightGlue @1241x376",
            "value": round(frames_total / dt_max, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (contractions: f16 hi/lo split operands, 3 MFMA per product, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "C2/C4: synthetic 1241x376x3 uint8 frame stream, ALIKED-n16 extract + "
                                   "LightGlue match (t-1,t), 2048 kpts/frame, min_conf 0.7, random-init weights",
                       "frames_per_step_per_gpu": FRAMES_PER_RANK, "max_kpts": MAX_KPTS,
                       "lightglue_layers_executed": stop, "kpts_matched": [n0, n1],
                       "parallelism": f"frame-shard x{world}; per GPU {N_EXT} extractor + {N_MAT} matcher streams"},
            # achieved = ALGORITHMIC flops (8 n0 n1 256 per launch) / HIP-event launch duration on one
            # stream; the kernel issues 3 v_mfma_f32_32x32x16_f16 per algorithmic product (executed = 3x)
            "roofline": {"bound": "mfma", "kernel": "lg_attention_p_kernel (v_mfma_f32_32x32x16_f16 x3 per product)",
                         "achieved": round(ach, 2) if ach else None, "peak": F16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / F16_MFMA_PEAK_TFLOPS, 4) if ach else None,
                         # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
                         # scripts/pmc_traffic.sh); not re-measured here: PMC collection needs rocprofv3
                         "traffic": _pmc_traffic(),
                         "executed_mfma_frac": round(3 * ach / F16_MFMA_PEAK_TFLOPS, 4) if ach else None,
                         "launches_timed": iso_n,
                         "avg_launch_us": round(iso_ms / max(iso_n, 1) * 1e3, 2),
                         "timed_region_launches": attn_n,
                         "timed_region_avg_bracket_us": round(attn_ms / max(attn_n, 1) * 1e3, 2),
                         # second denominator: the results are fp32-grade, and the exact-fp32 matrix-core
                         # rate (157.3 TFLOP/s) is the ceiling of any path that feeds fp32 operands to MFMA
                         "frac_of_f32_mfma_peak": round(ach / F32_MFMA_PEAK_TFLOPS, 4) if ach else None,
                         "pipeline_algorithmic_tflops": round(
                             frames_total / dt_max * (lightglue_gflop(min(n0, n1), stop) + ALIKED_GFLOP_PER_FRAME)
                             / 1e3 / world, 2),
                         "pipeline_frac_of_f32_mfma_peak": round(
                             frames_total / dt_max * (lightglue_gflop(min(n0, n1), stop) + ALIKED_GFLOP_PER_FRAME)
                             / 1e3 / world / F32_MFMA_PEAK_TFLOPS, 4)},
        }
        out["structured_input"] = {
            "value": round(s_steps * plan.frames_per_round() / s_dt_max, 2), "unit": "frames/s", "steps": s_steps,
            "what": "same pipeline on the 9x9-box low-pass noise translating 3 px/frame (SURVEY 8(d))",
            "matches_last_pair": int(s_info[-1, 0]), "lightglue_layers_executed": int(s_info[-1, 1]),
            "kpts_matched": [int(s_info[-1, 2]), int(s_info[-1, 3])]}
        if not args.no_cpu_baseline and world == 1:        # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
