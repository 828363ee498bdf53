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


def test_record_roundtrip():
    fs = load_pkg("frame_shard")
    K = 16
    rng = np.random.default_rng(0)
    xy = rng.random((K, 2), np.float32); desc = rng.random((K, 128), np.float32)
    rec = fs.pack_record(11, xy, desc, K)
    assert rec.shape == (fs.record_floats(K),) and rec.nbytes % 16 == 0
    n, x2, d2 = fs.unpack_record(rec, K)
    assert n == 11 and np.array_equal(x2, xy[:11]) and np.array_equal(d2, desc[:11])
    n, x2, d2 = fs.unpack_record(torch.from_numpy(rec), K)           # a gathered torch tensor unpacks the same
    assert n == 11 and np.array_equal(x2, xy[:11])


def _frame_record(fs, f, K):
    rng = np.random.default_rng(1000 + f)
    return fs.pack_record(1 + f % K, rng.random((K, 2), np.float32), rng.random((K, 128), np.float32), K)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import importlib
    import fake_device
    fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K = 3, 8
    plan = fs.ShardPlan(world, rank, B)
    comm = fs.GlooRowsComm(rank, world)
    assert comm.count() == world
    ctx = fake_device.FakeContext()                        # "device memory" of this rank: the exchange works on raw addresses
    rb = fs.record_floats(K) * 4
    local, gathered = ctx.malloc(B * rb), ctx.malloc(world * B * rb)
    for rnd in range(2):
        records = np.stack([_frame_record(fs, f, K) for f in plan.frames(rnd)])
        ctx.h2d(local, records)
        # the pipeline's form (FrameStreamPipeline.round): the round is gathered in two parts, as the extracts of each half
        # finish, into a buffer the caller owns
        ctx.view(gathered, world * B * rb)[:] = 0xff
        for lo, hi in ((0, 2), (2, 3)):
            comm.all_gather_rows(ctx, local, gathered, B, lo, hi, rb)
        full = np.empty((world * B, fs.record_floats(K)), np.float32)
        ctx.d2h(full, gathered)
        np.save(Path(out_dir) / f"r{rank}_round{rnd}.npy", full)
        # THE exchange of the pipeline: each half is one contiguous all-gather into block [half] of a map laid out
        # [half][rank][rows]; read back through the same row arithmetic as FrameStreamPipeline.map_row it is the same round
        ctx.view(gathered, world * B * rb)[:] = 0xff
        for lo, hi in ((0, 2), (2, 3)):
            comm.all_gather(ctx, local + lo * rb, gathered + world * lo * rb, (hi - lo) * rb)
        blocked = np.empty_like(full)
        ctx.d2h(blocked, gathered)
        rows = [world * lo + r * (hi - lo) + (s - lo) for r in range(world) for s in range(B)
                for lo, hi in ((0, 2), (2, 3)) if lo <= s < hi]
        assert np.array_equal(blocked[rows].view(np.int32), full.view(np.int32)), f"rank {rank} round {rnd}: half-blocked gather differs"
        comm.all_gather(ctx, local, gathered, 0)                              # an empty part is a no-op on every rank
        # whole blocks in one call give the same map
        ctx.view(gathered, world * B * rb)[:] = 0xff
        comm.all_gather_rows(ctx, local, gathered, B, 0, B, rb)
        again = np.empty_like(full)
        ctx.d2h(again, gathered)
        assert np.array_equal(again.view(np.int32), full.view(np.int32)), f"rank {rank} round {rnd}: part-wise gather differs"
        comm.all_gather_rows(ctx, local, gathered, B, 2, 2, rb)             # an empty part is a no-op on every rank
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def test_collate_world2_gloo(tmp_path):
    """The exchange of the frame-sharded pipeline's collation (`comm.all_gather_rows`, here frame_shard.GlooRowsComm over a
    stand-in device memory) with world size 2: every rank ends with the same shared map, in global frame order, ragged
    counts intact."""
    fs = load_pkg("frame_shard")
    world, B, K = 2, 3, 8
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rnd in range(2):
        maps = [np.load(tmp_path / f"r{r}_round{rnd}.npy") for r in range(world)]
        assert np.array_equal(maps[0].view(np.int32), maps[1].view(np.int32))
        assert maps[0].shape == (world * B, fs.record_floats(K))
        for i in range(world * B):
            f = rnd * world * B + i
            want = _frame_record(fs, f, K)
            assert np.array_equal(maps[0][i].view(np.int32), want.view(np.int32))     # bit for bit
            n, _, _ = fs.unpack_record(maps[0][i], K)
            assert n == 1 + f % K


def test_a_multi_rank_pipeline_needs_an_exchange():
    fs = load_pkg("frame_shard")
    with pytest.raises(ValueError, match="comm"):
        fs.FrameStreamPipeline([], [], fs.ShardPlan(2, 0, 3), 8)


# ---------------------------------------------------------------------------------------------------------------------------
# The WHOLE pipeline choreography at world size 8 on CPU: FrameStreamPipeline.round itself (frame ownership, record sets,
# per-half collation, the halo record - rank 0's from the previous round's map, the others' from the neighbour's last frame
# of the same round -, batch spans and their order) over stand-ins for the extractor / matcher / device memory
# (tests/fake_device.py: everything executes at enqueue time, so stream ORDERING is not what this covers - the shared-GPU
# and RCCL tests of tests/test_pipeline_gpu.py do) and the gloo exchange.  The driver's 8-GPU SCALE run is the first
# execution of this choreography on eight real devices; this is its rehearsal.
def _pipeline_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import importlib
    import fake_device as fd
    import lg_inputs
    torch.set_num_threads(1)
    fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K, P, ROUNDS, H, Wd = 3, 64, 2, 3, 4, 4
    n_frames = ROUNDS * world * B
    chain = lg_inputs.make_chain(n_frames, K, seed=9, period=4)
    plan = fs.ShardPlan(world, rank, B)
    dets = [fd.FakeAliked(fd.FakeContext(), chain, K, by_image=True, max_frames=2) for _ in range(2)]
    mats = [fd.FakeLightGlue(fd.FakeContext(), K, max_pairs=P) for _ in range(2)]
    pipe = fs.FrameStreamPipeline(dets, mats, plan, K, 0.3, batch_pairs=P, comm=fs.GlooRowsComm(rank, world))
    ctx = pipe.ctx

    def image(f):
        img = np.zeros((H, Wd, 3), np.uint8)
        img.reshape(-1)[:2] = (f % 256, f // 256)
        return img
    checked = 0
    for rnd in range(ROUNDS):
        mine = list(plan.frames(rnd))
        pipe.round(ctx.upload(np.stack([image(f) for f in mine])), H, Wd, 3)
        pipe.sync()
        info = pipe.infos()
        res = pipe.results()
        feats = pipe.features()
        smap = np.empty((world * B, pipe.REC), np.float32)
        ctx.d2h(smap, pipe.shared_map_ptr)
        for s, f in enumerate(mine):
            np.testing.assert_array_equal(feats[s][0], chain[f][0])
            np.testing.assert_array_equal(feats[s][1], chain[f][1])
            if f == 0:
                continue
            want_ij, want_sc = fd.fake_match(chain[f - 1][0], chain[f - 1][1], chain[f][0], chain[f][1], 0.3)
            assert info[s, 0] == len(want_ij) > 8, (rank, rnd, f, info[s])
            np.testing.assert_array_equal(res[s][0], want_ij, err_msg=f"rank {rank} round {rnd} frame {f} (pair with frame {f - 1})")
            checked += 1
        assert sorted(pipe.map_rows().tolist()) == list(range(world * B))
        for j in range(world * B):                           # the collated map: every rank's frames of the round ([half][rank][rows])
            f = rnd * world * B + j
            n, xy, desc = fs.unpack_record(smap[pipe.map_row(j)], K)
            assert n == K
            np.testing.assert_array_equal(xy, chain[f][0]); np.testing.assert_array_equal(desc, chain[f][1])
    np.save(Path(out_dir) / f"checked_{rank}.npy", np.array([checked]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_pipeline_choreography_world_n_on_cpu(tmp_path, world):
    mp.spawn(_pipeline_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    checked = [int(np.load(tmp_path / f"checked_{r}.npy")[0]) for r in range(world)]
    assert checked[0] == 3 * 3 - 1 and all(c == 3 * 3 for c in checked[1:])          # every pair of every rank, frame 0 excepted
