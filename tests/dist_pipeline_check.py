"""Launched by tests/test_pipeline_gpu.py under torch.distributed.run: every rank runs the frame-sharded
pipeline for THREE rounds enqueued back to back with no host synchronisation in between (so two rounds
are in flight: record-set double buffering, the per-half all-gathers on the collation stream, the halo
record of the previous round, extracts of round r+1 under the matches of round r all get exercised), and
checks its own frames against the sequential host API - keypoints, the match indices of EVERY pair including
the ones that straddle a rank / round boundary (which need the gathered features of the neighbour) - and the
collated shared map of every round.

    SSLAM_DIST_BACKEND = gloo (default; ranks may share GPU 0: frame_shard.GlooRowsComm, the rows through the host)
                       | rccl (RCCL driven directly, opencv-simpleslam_amd/rccl.py: one GPU per rank)
Both are a `comm` handed to the SAME pipeline code; records and the gathered map live in C-ABI memory either way, torch
only carries the rendezvous (and, for gloo, the rows) on the CPU and never touches the GPU.
"""
import importlib
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import frames                                                      # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("SSLAM_DIST_BACKEND", "gloo")
assert backend in ("gloo", "rccl"), backend
# the system HIP runtime first (the library), then torch for the CPU-side rendezvous only
pkg = importlib.import_module("opencv-simpleslam_amd")
nat = pkg._native
dev = int(os.environ.get("LOCAL_RANK", rank)) % nat.device_count() if backend == "rccl" else 0
ctx_first = nat.default_context(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
if backend == "rccl":
    rccl = importlib.import_module("opencv-simpleslam_amd.rccl")

    def _exchange(payload):
        box = [payload]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    comm = rccl.RcclComm.create(rank, world, _exchange)
    # count what the collation really calls in librccl (each half round must be ONE ncclAllGather, no
    # broadcast groups): a counting shim in front of the two entry points
    L = rccl.lib()
    rccl_calls = {"ncclAllGather": 0, "ncclBroadcast": 0}

    class _Shim:
        def __init__(self, L):
            self.__dict__["_L"] = L

        def __getattr__(self, name):
            f = getattr(self._L, name)
            if name in rccl_calls:
                def counted(*a, _f=f, _n=name):
                    rccl_calls[_n] += 1
                    return _f(*a)
                return counted
            return f
    rccl._lib = _Shim(L)
else:
    rccl_calls = None
    comm = importlib.import_module("opencv-simpleslam_amd.frame_shard").GlooRowsComm(rank, world)
pkg = importlib.import_module("opencv-simpleslam_amd")
nat = pkg._native
W = importlib.import_module("opencv-simpleslam_amd.weights")
fs = importlib.import_module("opencv-simpleslam_amd.frame_shard")
AL = importlib.import_module("opencv-simpleslam_amd.aliked").AlikedHIP
LG = importlib.import_module("opencv-simpleslam_amd.lightglue").LightGlueHIP
K, B, H, Wd, ROUNDS = 384, 3, 160, 256, 3
sd_a = W.random_aliked_state_dict(0)
sd_l = W.random_lightglue_state_dict(1, match_gain=4.0, match_bias=3.0)
n_frames = ROUNDS * world * B
imgs = [frames.structured_frame(i, h=H, w=Wd) for i in range(n_frames)]
ctx0 = nat.default_context(dev)
det0 = AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=ctx0)
mat0 = LG(sd_l, max_kpts=K, ctx=ctx0, filter_threshold=0.0)     # every mutual arg-max: non-vacuous
feats = [det0.extract(im, K) for im in imgs]
ref = [None] + [mat0.match(feats[i - 1][0], feats[i - 1][1], feats[i][0], feats[i][1], min_conf=0.0)
                for i in range(1, n_frames)]
plan = fs.ShardPlan(world, rank, B)
dets = [AL(sd_a, max_num_keypoints=K, max_h=H, max_w=Wd, ctx=nat.Context(dev), max_frames=2) for _ in range(2)]   # chunks of 2 frames per extractor call (ragged last chunk)
mats = [LG(sd_l, max_kpts=K, ctx=nat.Context(dev), max_pairs=2, filter_threshold=0.0) for _ in range(2)]
pipe = fs.FrameStreamPipeline(dets, mats, plan, K, 0.0, batch_pairs=2, collate_always=True, comm=comm)    # (one rank: still the collective path)
ctx = pipe.ctx
col = nat.Context(dev)                      # collector stream: snapshots a round's outputs without a host sync
chunks = [ctx.upload(np.stack([imgs[f] for f in plan.frames(r)])) for r in range(ROUNDS)]
hist = []
for rnd in range(ROUNDS):
    pipe.round(chunks[rnd], H, Wd, 3)
    pset = pipe.last_set
    for ev in pipe.ev_batch[pset][:pipe.n_batches[pset]] + pipe.ev_ext[pset] + [pipe.ev_collated[pset]]:
        col.wait(ev)
    h = dict(rec=col.malloc(B * pipe.REC * 4), ij=col.malloc(B * K * 8), info=col.malloc(B * 16),
             smap=col.malloc(world * B * pipe.REC * 4))
    col.d2d_async(h["rec"], pipe.rec_ptr(pset * B), B * pipe.REC * 4)
    col.d2d_async(h["ij"], pipe.ij, B * K * 8)
    col.d2d_async(h["info"], pipe.info, B * 16)
    col.d2d_async(h["smap"], pipe.shared_map_ptr, world * B * pipe.REC * 4)
    ev = col.event(); col.record(ev)        # the pipeline's next-but-one round must not overwrite before the copies ran
    for d in pipe.dets:
        d.ctx.wait(ev)
    pipe.cctx.wait(ev)
    hist.append(h)
pipe.sync(); col.sync()
checked = 0
for rnd, h in enumerate(hist):
    mine = list(plan.frames(rnd))
    rec = np.empty((B, pipe.REC), np.float32); ij = np.empty((B, K, 2), np.int32); info = np.empty((B, 4), np.int32)
    smap = np.empty((world * B, pipe.REC), np.float32)
    ctx.d2h(rec, h["rec"]); ctx.d2h(ij, h["ij"]); ctx.d2h(info, h["info"]); ctx.d2h(smap, h["smap"])
    for s, f in enumerate(mine):
        n, xy, desc = fs.unpack_record(rec[s], K)
        assert n == len(feats[f][0]), (rank, f)
        np.testing.assert_array_equal(xy, feats[f][0])
        if f == 0:
            continue
        assert info[s, 0] >= 0, (rank, f)
        np.testing.assert_array_equal(ij[s, :info[s, 0]], ref[f][0], err_msg=f"rank {rank} round {rnd} frame {f}")
        checked += 1
    # the collated map holds every rank's features of this round ([half][rank][rows]: pipe.map_row)
    for j in range(world * B):
        f = rnd * world * B + j
        n, xy, desc = fs.unpack_record(smap[pipe.map_row(j)], K)
        assert n == len(feats[f][0]), (rank, rnd, j)
        np.testing.assert_array_equal(xy, feats[f][0])
        np.testing.assert_array_equal(desc, feats[f][1])
for h in hist:
    for p_ in h.values():
        col.free(p_)
for c_ in chunks:
    ctx.free(c_)
assert sum(len(r[0]) for r in ref[1:]) > 5, "vacuous: the sequential reference found no matches"
if rccl_calls is not None:
    assert rccl_calls == {"ncclAllGather": 2 * ROUNDS, "ncclBroadcast": 0}, rccl_calls      # one all-gather per half round
    print(f"rank {rank}: librccl calls over {ROUNDS} rounds: {rccl_calls}", flush=True)
dist.barrier()
print(f"rank {rank} ({backend}, device {dev}): {checked} pairs identical to the sequential API over {ROUNDS} un-synchronised rounds", flush=True)
comm.close()
dist.destroy_process_group()
