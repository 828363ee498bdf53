"""Multi-GPU path on CPU: shard plan + all-gather collation with world_size 2 over gloo."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_shard_plan_covers_stream_once():
    fs = load_pkg("frame_shard")
    world, B = 4, 3
    plans = [fs.ShardPlan(world, r, B) for r in range(world)]
    seen = []
    for rnd in range(5):
        for p in plans:
            seen += list(p.frames(rnd))
    assert seen == list(range(5 * world * B))            # contiguous chunks, every frame exactly once
    for f in range(1, 40):
        rnd, rank, slot = plans[0].owner(f)
        assert plans[rank].frames(rnd)[slot] == f
    assert plans[0].halo(0) is None
    assert plans[1].halo(0) == B - 1 and plans[0].halo(1) == world * B - 1
    assert plans[2].frames_per_round() == 12


def test_pack_unpack_roundtrip():
    fs = load_pkg("frame_shard")
    K = 16
    xy = torch.rand(K, 2); desc = torch.rand(K, 128)
    blk = fs.pack_rows(torch.tensor(11, dtype=torch.int32), xy, desc, K)
    n, x2, d2 = fs.unpack_rows(blk)
    assert n == 11 and torch.equal(x2, xy[:11]) and torch.equal(d2, desc[:11])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import importlib
    fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K = 3, 8
    plan = fs.ShardPlan(world, rank, B)
    for rnd in range(2):
        blocks = torch.zeros(B, K + 1, fs.ROW)
        for s, f in enumerate(plan.frames(rnd)):
            g = torch.Generator().manual_seed(1000 + f)
            n = 1 + f % K
            blocks[s] = fs.pack_rows(n, torch.rand(K, 2, generator=g), torch.rand(K, 128, generator=g), K)
        full = fs.collate(blocks, plan)
        torch.save(full, Path(out_dir) / f"r{rank}_round{rnd}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_collate_world2_gloo(tmp_path):
    """Every rank ends with the same shared map, in global frame order, ragged counts intact."""
    fs = load_pkg("frame_shard")
    world, B, K = 2, 3, 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rnd in range(2):
        maps = [torch.load(tmp_path / f"r{r}_round{rnd}.pt") for r in range(world)]
        assert torch.equal(maps[0], maps[1])
        assert maps[0].shape == (world * B, K + 1, fs.ROW)
        for i in range(world * B):
            f = rnd * world * B + i
            g = torch.Generator().manual_seed(1000 + f)
            xy, desc = torch.rand(K, 2, generator=g), torch.rand(K, 128, generator=g)
            n, x2, d2 = fs.unpack_rows(maps[0][i])
            assert n == 1 + f % K and torch.equal(x2, xy[:n]) and torch.equal(d2, desc[:n])


def test_collate_single_rank_is_identity():
    fs = load_pkg("frame_shard")
    x = torch.rand(2, 5, fs.ROW)
    assert fs.collate(x, fs.ShardPlan(1, 0, 2)) is x
