"""The multi-stream frame pipeline (frame_shard.FrameStreamPipeline) must give exactly what the
sequential host API gives, frame by frame: same keypoints, same match indices."""
import importlib

import numpy as np
import pytest

import frames
from conftest import ROOT, load_pkg

pytestmark = pytest.mark.gpu


def test_pipeline_equals_sequential_api(native):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    W = load_pkg("weights"); fs = load_pkg("frame_shard")
    AL = load_pkg("aliked").AlikedHIP; LG = load_pkg("lightglue").LightGlueHIP
    K, B, H, Wd = 512, 4, 200, 320
    sd_a = W.random_aliked_state_dict(0)
    sd_l = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
    imgs = [frames.structured_frame(i, h=H, w=Wd) for i in range(2 * B)]

    # sequential reference through the host entry points
    ctx0 = native.default_context(0)
    det0 = AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=ctx0)
    mat0 = LG(sd_l, max_kpts=K, ctx=ctx0)
    feats = [det0.extract(im, K) for im in imgs]
    ref = [None] + [mat0.match(feats[i - 1][0], feats[i - 1][1], feats[i][0], feats[i][1], min_conf=0.2)
                    for i in range(1, len(imgs))]

    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        se = [torch.cuda.Stream() for _ in range(2)]
        sm = [torch.cuda.Stream() for _ in range(3)]
        dets = [AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=native.Context(0, stream=s.cuda_stream)) for s in se]
        mats = [LG(sd_l, max_kpts=K, ctx=native.Context(0, stream=s.cuda_stream)) for s in sm]
        pipe = fs.FrameStreamPipeline(dets, mats, fs.ShardPlan(1, 0, B), K, 0.2, streams_e=se, streams_m=sm)
        for rnd in range(2):
            chunk = torch.from_numpy(np.stack(imgs[rnd * B:(rnd + 1) * B])).cuda()
            torch.cuda.synchronize()
            pipe.round(chunk, H, Wd, 3)
            res = pipe.results()
            cnt = pipe.count.cpu().numpy()[:, 0]
            xy = pipe.xy.cpu().numpy()
            for s in range(B):
                f = rnd * B + s
                assert cnt[s] == len(feats[f][0])
                np.testing.assert_array_equal(xy[s, :cnt[s]], feats[f][0])
                if f == 0:
                    continue
                np.testing.assert_array_equal(res[s][0], ref[f][0])
                np.testing.assert_allclose(res[s][1], ref[f][1], atol=1e-6)
    assert sum(len(r[0]) for r in ref[1:]) >= 0


def _run_bench(extra_env, cmd):
    import json, os, subprocess, sys
    env = dict(os.environ, SSLAM_BENCH_FRAMES="6", SSLAM_BENCH_NE="1", SSLAM_BENCH_NM="2", **extra_env)
    out = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]               # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_contract_single_rank():
    import sys
    d = _run_bench({}, [sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "structured_input"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["unit"] == "frames/s"
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 code path (frame sharding, all-gather collation, boundary pair, max-over-ranks
    timing) on a 1-GPU box: two ranks on the same device, gloo instead of RCCL."""
    import sys
    d = _run_bench({"SSLAM_DIST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"},
                   [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29611", "bench.py", "--gpus", "2",
                    "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["frames_per_step_per_gpu"] == 6


def test_two_rank_pipeline_equals_sequential_api():
    """N > 1 data path: tests/dist_pipeline_check.py under torch.distributed.run, 2 gloo ranks on one GPU."""
    import os, subprocess, sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29613", "tests/dist_pipeline_check.py"],
                         cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert out.stdout.count("pairs identical to the sequential API") == 2
