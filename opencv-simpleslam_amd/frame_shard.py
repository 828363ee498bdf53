"""Frame-sharded extract + match pipeline (one process per GPU).

The reference runs one frame at a time in one process
(slam/monocular/main_revamped.py:321-328: feature_extractor(frame t) then
feature_matcher(t-1 -> t)).  Frames are independent for extraction and pairs
are independent for matching, so a stream shards across the GPUs of a node
with no data-path collective: in every round rank r owns the contiguous chunk

    frames [ (round * world + r) * B , ... + B )

and matches each of its frames against the previous one.  The only exchange
is the collation of {count, keypoints, descriptors} of all frames back into
the shared map every rank keeps (RCCL all-gather over xGMI through
torch.distributed, padded to max_features rows + one header row); the pair
that straddles a chunk boundary is matched after that gather against the
neighbour's last frame.

Everything on the device is enqueued on one HIP stream (the torch current
stream the Context was created on); counts stay device-resident, so a round
has no host synchronisation.

`ShardPlan` and `collate` are pure host/tensor logic and are covered by the
world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DESC_DIM = 128
ROW = 2 + DESC_DIM            # x, y, descriptor


@dataclass(frozen=True)
class ShardPlan:
    world: int
    rank: int
    frames_per_rank: int       # B

    def frames(self, rnd: int):
        """Global frame ids this rank extracts in round `rnd`."""
        first = (rnd * self.world + self.rank) * self.frames_per_rank
        return range(first, first + self.frames_per_rank)

    def owner(self, frame: int):
        """(round, rank, slot) that extracts `frame`."""
        chunk, slot = divmod(frame, self.frames_per_rank)
        rnd, rank = divmod(chunk, self.world)
        return rnd, rank, slot

    def halo(self, rnd: int):
        """Frame this rank's first frame of round `rnd` is matched against, or None (frame 0)."""
        f = self.frames(rnd)[0] - 1
        return f if f >= 0 else None

    def frames_per_round(self):
        return self.world * self.frames_per_rank


def pack_rows(count, xy, desc, max_kpts):
    """[max_kpts+1, 130] float32 block: row 0 = header (count), rows 1.. = (x, y, desc)."""
    import torch
    blk = torch.zeros((max_kpts + 1, ROW), dtype=torch.float32, device=xy.device)
    blk[0, 0] = count.to(torch.float32) if hasattr(count, "to") else float(count)
    blk[1:, :2] = xy[:max_kpts]
    blk[1:, 2:] = desc[:max_kpts]
    return blk


def unpack_rows(blk):
    n = int(blk[0, 0].item())
    return n, blk[1:1 + n, :2], blk[1:1 + n, 2:]


def collate(local_blocks, plan: ShardPlan, group=None):
    """All-gather the per-frame blocks of one round.

    local_blocks: [B, max_kpts+1, 130] tensor of this rank's frames (frame order).
    Returns [world*B, max_kpts+1, 130] in GLOBAL frame order of the round.  With
    world == 1 this is the input (no collective)."""
    import torch
    if plan.world == 1:
        return local_blocks
    import torch.distributed as dist
    out = torch.empty((plan.world * local_blocks.shape[0],) + tuple(local_blocks.shape[1:]),
                      dtype=local_blocks.dtype, device=local_blocks.device)
    dist.all_gather_into_tensor(out, local_blocks.contiguous(), group=group)   # rank-major = frame order
    return out


class FrameStreamPipeline:
    """Device-resident extract(t) + match(t-1 -> t) over this rank's frame chunks."""

    def __init__(self, detector, matcher, plan: ShardPlan, max_kpts: int, min_conf: float = 0.7):
        import torch
        self.torch = torch
        self.det, self.mat, self.plan = detector, matcher, plan
        self.K = int(max_kpts)
        self.min_conf = float(min_conf)
        dev = torch.device("cuda", torch.cuda.current_device())
        B, K = plan.frames_per_rank, self.K
        # per-frame feature blocks of the current round (+ slot B: the halo frame)
        self.blocks = torch.zeros((B + 1, K + 1, ROW), dtype=torch.float32, device=dev)
        self.xy = torch.zeros((B + 1, K, 2), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((B + 1, K, DESC_DIM), dtype=torch.float32, device=dev)
        self.score = torch.zeros((B + 1, K), dtype=torch.float32, device=dev)
        self.count = torch.zeros((B + 1, 1), dtype=torch.int32, device=dev)
        self.ij = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
        self.msc = torch.zeros((B, K), dtype=torch.float32, device=dev)
        self.info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        self.have_halo = False
        self.shared_map = None          # last collated round [world*B, K+1, 130]

    def _match(self, a, b, out):
        self.mat.match_dev(self.xy[a], self.desc[a], self.K, self.xy[b], self.desc[b], self.K,
                           self.ij[out], self.msc[out], self.info[out], min_conf=self.min_conf,
                           m_dev=self.count[a], n_dev=self.count[b])

    def round(self, frames_dev, H, W, C):
        """frames_dev: uint8 [B, H, W, C] device tensor holding this rank's chunk.
        Enqueues B extracts + B matches (+ the collation when world > 1)."""
        torch, plan, B = self.torch, self.plan, self.plan.frames_per_rank
        for s in range(B):
            self.det.extract_dev(frames_dev[s], H, W, C, self.xy[s], self.desc[s], self.score[s],
                                 self.count[s], max_kpts=self.K)
            if s > 0:
                self._match(s - 1, s, s)
        if plan.world == 1:
            if self.have_halo:                       # previous round's last frame -> this round's first
                self._match(B, 0, 0)
            self.xy[B].copy_(self.xy[B - 1]); self.desc[B].copy_(self.desc[B - 1])
            self.count[B].copy_(self.count[B - 1])
            self.have_halo = True
            return
        # collate: every rank ends up with all frames of the round (the shared map)
        for s in range(B):
            self.blocks[s, 0, 0] = self.count[s, 0].to(torch.float32)
            self.blocks[s, 1:, :2] = self.xy[s]
            self.blocks[s, 1:, 2:] = self.desc[s]
        self.shared_map = collate(self.blocks[:B], plan)
        # boundary pair: my first frame vs the previous chunk's last frame
        prev = plan.rank * B - 1                     # index inside the gathered round
        if prev >= 0:
            src = self.shared_map[prev]
        elif self.have_halo:
            src = self.prev_round_last
        else:
            src = None
        if src is not None:
            self.xy[B].copy_(src[1:, :2]); self.desc[B].copy_(src[1:, 2:])
            self.count[B, 0] = src[0, 0].to(torch.int32)
            self._match(B, 0, 0)
        self.prev_round_last = self.shared_map[plan.world * B - 1].clone()
        self.have_halo = True

    def results(self):
        """Host copy of the last round's matches: list of (ij [K,2], scores [K]) per local frame."""
        self.torch.cuda.synchronize()
        info = self.info.cpu().numpy()
        ij = self.ij.cpu().numpy()
        sc = self.msc.cpu().numpy()
        return [(ij[s, :info[s, 0]].copy(), sc[s, :info[s, 0]].copy()) for s in range(len(info))]
