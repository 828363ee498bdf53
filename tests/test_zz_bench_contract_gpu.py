"""bench.py's JSON contract, run as a subprocess the way the driver runs it.  Collected LAST (file name) so that no
parity file sits behind it under `-x`, and it asserts the CONTRACT of the line only (keys, shapes, signs): relations
between two sub-second timings are noise on a fresh box and do not belong in a parity suite."""
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(extra_env, cmd):
    import json, os, subprocess, sys
    env = dict(os.environ, SSLAM_BENCH_FRAMES="6", SSLAM_BENCH_NE="1", SSLAM_BENCH_NM="2", SSLAM_BENCH_PAIRS="4", **extra_env)
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]               # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_contract_single_rank():
    import sys
    d = _run_bench({}, [sys.executable, "bench.py", "--steps", "2", "--warmup", "4", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "structured_input",
                "exact_f32", "ba", "reproject", "step_ms", "timed_region_s", "early_stop", "dropin", "pcie", "c5", "kpts4000", "f16x3", "build",
                "gpu_busy_s"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["unit"] == "frames/s"
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["in_pipeline_frac"] > 0 and d["exact_f32"]["value"] > 0        # present and positive; which is larger is bench.py's business
    assert d["ba"]["device_lm_ms"] > 0 and 0 < d["ba"]["residual_kernel"]["frac"] < 1 and d["reproject"]["wall_ms"] > 0
    assert 0 < d["step_ms"]["p10"] <= d["step_ms"]["p50"] <= d["step_ms"]["p90"] <= d["step_ms"]["max"]
    assert d["dropin"]["value"] > 0 and d["dropin"]["feature_matcher_ms"] > 0, d["dropin"]
    # the drop-in leg does the work of the reference's loop: real matches, so RANSAC / DMatch / read-back are in the number
    assert d["dropin"]["matches_median"] > 100 and d["dropin"]["filter_matches_ransac_ms"] > 0, d["dropin"]
    # ... and the reference's real call pattern: keyframe -> cur (+ its duplicate) beyond the cooldown, answered from the device
    sl = d["dropin"]["slam_loop"]
    assert sl["value"] > 0 and sl["keyframe_frames"] >= 3 and sl["answered_from"]["memo"] >= 3 and sl["answered_from"]["reupload"] == 0, sl
    dc = d["dropin"]["slam_loop_depth_control"]
    assert dc.get("value", 0) > 0 and dc["lightglue_layers_last_pair"] < 9 and dc["matches_median"] > 100, dc      # early stop fired, matches kept
    v4 = d["dropin"]["value_kpts4000"]                   # the reference's default max_features
    assert v4.get("value", 0) > 0 and v4["keypoints"] == 4000 and v4["matches_median"] > 100, v4
    assert "resident in HBM" in d["config"]["workload"] and d["gpu_busy_s"] > 0
    assert d["pcie"]["value"] > 0 and d["pcie"]["h2d_bytes_per_round"] == 6 * 1241 * 376 * 3, d["pcie"]
    for leg in ("c5", "kpts4000"):
        assert d[leg].get("value", 0) > 0 and 0 < d[leg]["aliked_hbm"]["frac"] < 1, d[leg]
    assert d["f16x3"]["value"] > 0 and 0 < d["f16x3"]["attention"]["frac"] < 1 and "f16x3p1" in d["dtype"]
    assert len(d["build"]["csrc_digest"]) == 12
    assert d["kpts4000"]["max_kpts"] == 4000 and 0 < d["kpts4000"]["attention"]["frac"] < 1
    es = d["early_stop"]
    assert es["value"] > 0 and es["lightglue_layers_histogram"] and set(es["lightglue_layers_histogram"]) != {"9"}, es
    assert es["points_pruned"] and es["kpts_after_pruning_min_max"][0] < 2048, es          # the width control fired under load
    # r06: the batched pipeline on frames that match (emit / compaction / the per-pair outputs carry hundreds of matches) ...
    pm = d["planted_matches"]
    assert pm.get("value", 0) > 0 and pm["matches_per_pair"] > 100 and pm["lightglue_layers_executed"] == 9, pm
    # ... and the drop-in loops with cv2's own KeyPoint / DMatch classes beside their like-for-like partner (present and sane;
    # how close the two rates are is bench.py's business, not a test's)
    cv = d["dropin"]["cv2_classes"]
    assert cv.get("value", 0) > 0 and cv["matches_median"] > 100 and cv["slam_loop"]["value"] > 0, cv
    assert cv["duck_types_same_conditions"]["value"] > 0 and cv["answered_from"]["ahead"] > 0, cv


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run starts its ranks as a child process
    (same command shape as --gpus 1); gloo + one shared GPU on this box."""
    import sys
    d = _run_bench({"SSLAM_DIST_BACKEND": "gloo", "MASTER_PORT": "29617"},
                   [sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "4", "--no-cpu-baseline",
                    "--no-extras"])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["frames_per_step_per_gpu"] == 6


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 code path (frame sharding, all-gather collation, boundary pair, max-over-ranks
    timing) on a 1-GPU box: two ranks on the same device, gloo instead of RCCL."""
    import sys
    d = _run_bench({"SSLAM_DIST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"},
                   [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2",
                    "--steps", "2", "--warmup", "4", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["frames_per_step_per_gpu"] == 6


def test_bench_direct_rccl_with_one_rank():
    """bench.py's N > 1 branches (SSLAM_BENCH_FORCE_DIST=1: gloo rendezvous, collective barrier around the timed region,
    max-reduce of the times, the pipeline's collation path) on the production exchange - RCCL driven directly, no torch in
    the data path - with one rank on this box: same JSON contract, and the line reports the communicator's own rank count."""
    import sys
    d = _run_bench({"SSLAM_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29627"},
                   [sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "4", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["rccl_ranks"] == 1 and d["config"]["collation_backend"] == "rccl (direct)"


def test_bench_falls_back_to_the_host_exchange_when_rccl_cannot_be_loaded():
    """No scaling line lost to the collation transport: with librccl.so not loadable every rank agrees (over the gloo rendezvous)
    to exchange the rows through the host, and the line says which transport ran and why."""
    import sys
    d = _run_bench({"SSLAM_BENCH_FORCE_DIST": "1", "SSLAM_RCCL_LIB": "/nonexistent/librccl.so", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": "29633"},
                   [sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "4", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["rccl_ranks"] is None
    assert d["config"]["collation_backend"].startswith("gloo (host round trip; RCCL not available: rank 0:"), d["config"]["collation_backend"]
