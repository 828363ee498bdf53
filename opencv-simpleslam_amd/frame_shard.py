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

Inside one GPU several frames are in flight as well: NE extractor and NM
matcher instances, each on its own HIP stream, chained by events
(match(t-1, t) waits for extract(t-1) and extract(t)).  Each pair / frame is a
chain of ~130 / ~35 short kernels that is bound by per-block latency, not by
the chip, so independent chains overlap almost freely (measured: 6-7 matcher
streams = 1.5x the pairs/s of one; 8 streams on the runtime's 4 hardware queues is
the measured optimum - a ninth stream unbalances the queues and costs > 10 %).  Counts stay device-resident, so a round
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
    """Device-resident extract(t) + match(t-1 -> t) over this rank's frame chunks.

    `detectors` / `matchers` are lists of instances, each created on its own Context / HIP
    stream (`streams_e[i]`, `streams_m[j]` are the matching torch streams).  Frame s of a round
    is extracted on extractor s % NE and pair (s-1, s) is matched on matcher s % NM, chained by
    events, so several frames and pairs are in flight on one GPU."""

    def __init__(self, detectors, matchers, plan: ShardPlan, max_kpts: int, min_conf: float = 0.7,
                 streams_e=None, streams_m=None):
        import torch
        self.torch = torch
        self.dets = list(detectors) if isinstance(detectors, (list, tuple)) else [detectors]
        self.mats = list(matchers) if isinstance(matchers, (list, tuple)) else [matchers]
        self.plan = plan
        cur = torch.cuda.current_stream()
        self.se = list(streams_e) if streams_e else [cur] * len(self.dets)
        self.sm = list(streams_m) if streams_m else [cur] * len(self.mats)
        assert len(self.se) == len(self.dets) and len(self.sm) == len(self.mats)
        self.K = int(max_kpts)
        self.min_conf = float(min_conf)
        dev = torch.device("cuda", torch.cuda.current_device())
        B, K = plan.frames_per_rank, self.K
        # per-frame feature slots of the current round (+ slot B: the halo frame)
        self.blocks = torch.zeros((B + 1, K + 1, ROW), dtype=torch.float32, device=dev)
        self.xy = torch.zeros((B + 1, K, 2), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((B + 1, K, DESC_DIM), dtype=torch.float32, device=dev)
        self.score = torch.zeros((B + 1, K), dtype=torch.float32, device=dev)
        self.count = torch.zeros((B + 1, 1), dtype=torch.int32, device=dev)
        self.ij = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
        self.msc = torch.zeros((B, K), dtype=torch.float32, device=dev)
        self.info = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        self.ev_ext = [torch.cuda.Event() for _ in range(B)]
        self.ev_mdone = [torch.cuda.Event() for _ in self.sm]
        self.have_halo = False
        self.started = False
        self.shared_map = None          # last collated round [world*B, K+1, 130]

    def _match(self, m, a, b, out):
        self.mats[m].match_dev(self.xy[a], self.desc[a], self.K, self.xy[b], self.desc[b], self.K,
                               self.ij[out], self.msc[out], self.info[out], min_conf=self.min_conf,
                               m_dev=self.count[a], n_dev=self.count[b])

    def round(self, frames_dev, H, W, C):
        """frames_dev: uint8 [B, H, W, C] device tensor holding this rank's chunk.
        Enqueues B extracts + B matches (+ the collation when world > 1)."""
        torch, plan, B = self.torch, self.plan, self.plan.frames_per_rank
        NE, NM = len(self.dets), len(self.mats)
        if self.started:     # slots are overwritten: the previous round's readers must be done
            for st in set(self.se):
                for ev in self.ev_mdone:
                    st.wait_event(ev)
        self.started = True
        for s in range(B):
            e = s % NE
            with torch.cuda.stream(self.se[e]):
                self.dets[e].extract_dev(frames_dev[s], H, W, C, self.xy[s], self.desc[s], self.score[s],
                                         self.count[s], max_kpts=self.K)
                self.ev_ext[s].record(self.se[e])
        single = plan.world == 1
        for s in range(B):
            m = s % NM
            st = self.sm[m]
            st.wait_event(self.ev_ext[s])
            if s > 0:
                st.wait_event(self.ev_ext[s - 1])
            with torch.cuda.stream(st):
                if s > 0:
                    self._match(m, s - 1, s, s)
                elif single and self.have_halo:      # previous round's last frame -> this round's first
                    self._match(m, B, 0, 0)          # (slot B was filled on this same stream, sm[0])
        if single:
            st = self.sm[0]
            st.wait_event(self.ev_ext[B - 1])
            with torch.cuda.stream(st):
                self.xy[B].copy_(self.xy[B - 1]); self.desc[B].copy_(self.desc[B - 1])
                self.count[B].copy_(self.count[B - 1])
            for m in range(NM):
                self.ev_mdone[m].record(self.sm[m])
            self.have_halo = True
            return
        # ---- multi-GPU: collate as soon as the EXTRACTS are done; this round's matches keep running
        # on their own streams underneath the all-gather.  The collation is issued on the last
        # extractor stream, not on a stream of its own: streams share 4 hardware queues round-robin,
        # and one more active stream unbalances them (measured: 2 + 7 streams 490 frames/s against
        # 552 for 2 + 6).
        cst = self.se[-1]
        for st in set(self.se) - {cst}:
            cst.wait_stream(st)
        with torch.cuda.stream(cst):
            self.blocks[:B, 0, 0] = self.count[:B, 0].to(torch.float32)      # three fused copies, not 3 B
            self.blocks[:B, 1:, :2] = self.xy[:B]
            self.blocks[:B, 1:, 2:] = self.desc[:B]
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
                # slot B may still be read by the previous round's boundary match on sm[0]
                ev0 = torch.cuda.Event(); ev0.record(self.sm[0])
                cst.wait_event(ev0)
                self.xy[B].copy_(src[1:, :2]); self.desc[B].copy_(src[1:, 2:])
                self.count[B, 0] = src[0, 0].to(torch.int32)
                ev = torch.cuda.Event(); ev.record(cst)
                self.sm[0].wait_event(ev)
                with torch.cuda.stream(self.sm[0]):
                    self._match(0, B, 0, 0)
            self.prev_round_last = self.shared_map[plan.world * B - 1].clone()
        self.have_halo = True
        for m in range(NM):
            self.ev_mdone[m].record(self.sm[m])

    def results(self):
        """Host copy of the last round's matches: list of (ij [K,2], scores [K]) per local frame."""
        self.torch.cuda.synchronize()
        info = self.info.cpu().numpy()
        ij = self.ij.cpu().numpy()
        sc = self.msc.cpu().numpy()
        return [(ij[s, :info[s, 0]].copy(), sc[s, :info[s, 0]].copy()) for s in range(len(info))]
