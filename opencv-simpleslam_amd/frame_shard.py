"""Frame-sharded extract + match pipeline (one process per GPU).

The reference runs one frame at a time in one process
(slam/monocular/main_revamped.py:321-328: feature_extractor(frame t) then
feature_matcher(t-1 -> t)).  Frames are independent for extraction and pairs
are independent for matching, so a stream shards across the GPUs of a node
with no data-path collective: in every round rank r owns the contiguous chunk

    frames [ (round * world + r) * B , ... + B )

and matches each of its frames against the previous one.  The only exchange
is the collation of the per-frame feature records of all frames back into
the shared map every rank keeps (RCCL all-gather over xGMI through
torch.distributed); the pair that straddles a chunk boundary is matched after
that gather against the neighbour's last frame.

Inside one GPU a round is
  * B extracts, dealt round-robin over the extractor instances (each on its own HIP stream:
    a frame is a chain of ~45 short kernels that does not fill the chip), and
  * ceil(B / P) BATCHED matches of P pairs each (`sslam_lightglue_match_batch_dev`: every launch
    of the forward covers all P pairs, which is what fills the 256 CUs), dealt over the matcher
    instances, chained to the extracts by events.
Counts stay device-resident, so a round has no host synchronisation.

Per-frame feature record (the unit of the exchange, device resident, float32 words):

    [ xy : K x 2 | descriptors : K x 128 | count (int32 bits) | 3 pad ]       REC = 130 K + 4

The extractor writes straight into the record of its slot and the matcher reads straight from it,
so collation is ONE all-gather of the round's B records with no packing pass.

On one GPU the pipeline runs on the C-ABI alone (streams, events, buffers through
`_native.Context`); torch is imported only for the N > 1 collective.  `ShardPlan`, the record
helpers and `collate` are pure host logic and are covered by the world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DESC_DIM = 128
ROW = 2 + DESC_DIM            # x, y, descriptor
REC_TAIL = 4                  # count + padding (keeps records 16-byte multiples)


def record_floats(max_kpts: int) -> int:
    return int(max_kpts) * ROW + REC_TAIL


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


def pack_record(count, xy, desc, max_kpts):
    """Host-side construction of one record (tests / tools): numpy float32 [REC]."""
    K = int(max_kpts)
    rec = np.zeros(record_floats(K), np.float32)
    xy = np.asarray(xy, np.float32).reshape(-1, 2)[:K]
    desc = np.asarray(desc, np.float32).reshape(-1, DESC_DIM)[:K]
    rec[:2 * len(xy)] = xy.ravel()
    rec[2 * K:2 * K + DESC_DIM * len(desc)] = desc.ravel()
    rec[K * ROW:K * ROW + 1].view(np.int32)[0] = int(count)
    return rec


def unpack_record(rec, max_kpts):
    """(count, xy [count,2], desc [count,128]) views of one record (numpy array or torch tensor)."""
    K = int(max_kpts)
    if hasattr(rec, "detach"):
        rec = rec.detach().cpu().numpy()
    rec = np.ascontiguousarray(rec, np.float32)
    n = int(rec[K * ROW:K * ROW + 1].view(np.int32)[0])
    if n < 0:                                        # (al_finalize_kernel: the extractor's range flag for this frame)
        raise RangeOverflowError("ALIKED split-precision range overflow: this frame's features are void (|activation| >= 65520 "
                                 "does not fit the fp16 planes)")
    return n, rec[:2 * K].reshape(K, 2)[:n], rec[2 * K:K * ROW].reshape(K, DESC_DIM)[:n]


def collate(local_records, plan: ShardPlan, group=None, out=None, part=None, always=False):
    """All-gather the per-frame records of one round (or of a part of it).

    local_records: [B, REC] float32 tensor of this rank's frames (frame order).
    Returns [world*B, REC] in GLOBAL frame order of the round.  With world == 1 this is the
    input (no collective).
    out:  optional preallocated [world*B, REC] tensor to gather into (the caller keeps it alive: no
          allocation and no copy per round);
    part: optional (lo, hi) - gather only local frames lo..hi-1 of every rank, into rows
          r*B + lo .. r*B + hi - 1 of `out`, so a round can be collated in pieces as its extracts finish.
    always: run the collective with one rank too (tests: the RCCL path on a single GPU)."""
    if plan.world == 1 and not always:
        return local_records
    import torch
    import torch.distributed as dist
    B = local_records.shape[0]
    if out is None:
        out = torch.empty((plan.world * B,) + tuple(local_records.shape[1:]),
                          dtype=local_records.dtype, device=local_records.device)
    if part is None or (part[0] == 0 and part[1] == B):
        dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)   # rank-major = frame order
    else:
        lo, hi = part
        views = [out[r * B + lo:r * B + hi] for r in range(plan.world)]             # each contiguous
        dist.all_gather(views, local_records[lo:hi].contiguous(), group=group)
    return out


class RangeOverflowError(RuntimeError):
    pass


class FrameStreamPipeline:
    """Device-resident extract(t) + match(t-1 -> t) over this rank's frame chunks.

    `detectors` / `matchers` are lists of AlikedHIP / LightGlueHIP instances, each created on its
    own `_native.Context` (= its own HIP stream).  Frame s of a round is extracted on extractor
    s % NE - or, when the extractors were created with max_frames = EF > 1, chunk c of EF consecutive frames
    goes through one batched launch sequence on extractor c % NE; pairs are matched in batches of `batch_pairs` on matcher (batch index) % NM (each
    matcher needs max_pairs >= batch_pairs)."""

    def __init__(self, detectors, matchers, plan: ShardPlan, max_kpts: int, min_conf: float = 0.7,
                 batch_pairs: int | None = None, group=None, use_graphs: bool = True, collate_always: bool = False,
                 comm=None):
        self.dets = list(detectors) if isinstance(detectors, (list, tuple)) else [detectors]
        self.mats = list(matchers) if isinstance(matchers, (list, tuple)) else [matchers]
        self.plan, self.group = plan, group
        # collate_always: take the multi-GPU path (torch-owned slab, collation stream, per-half all-gathers, halo record)
        # with ONE rank too - how the tests run the RCCL branch on a single GPU
        # comm: an `rccl.RcclComm` - the collation runs on RCCL directly, over memory of the C-ABI (no torch in the data
        # path); without it the exchange goes through torch.distributed (`group`; gloo in the CPU / shared-GPU tests)
        self.distributed = plan.world > 1 or bool(collate_always)
        self.comm = comm if self.distributed else None
        # every round cycles through the same record slots: each extractor / matcher call sequence is
        # replayed as a cached hipGraph (one hipGraphLaunch instead of 45 / 190 launches per call)
        for x in self.dets + self.mats:
            x.use_graphs(bool(use_graphs))
        self.K = int(max_kpts)
        if self.K % 2:
            raise ValueError("max_kpts must be even (16-byte aligned descriptor rows inside a record)")
        self.min_conf = float(min_conf)
        self.EF = min(min(getattr(d, "max_frames", 1) for d in self.dets), plan.frames_per_rank)   # frames per extractor call
        self.P = int(batch_pairs or min(m.max_pairs for m in self.mats))
        if any(m.max_pairs < self.P for m in self.mats):
            raise ValueError(f"batch_pairs={self.P} exceeds a matcher's max_pairs")
        B, K = plan.frames_per_rank, self.K
        self.REC = record_floats(K)
        self.ctx = self.dets[0].ctx                      # allocations / copies that belong to no instance
        # Two record sets (round parity) so that the extracts of round r+1 run underneath the matches of
        # round r; + two halo slots for the N > 1 boundary record.  Outputs are double-buffered alike.
        self.NSLOT = 2 * B + 2
        dev_bytes = self.NSLOT * self.REC * 4
        self.torch = None
        self.cctx = None
        if self.distributed and self.comm is not None:
            from . import _native
            self.slab = self.ctx.malloc(dev_bytes)
            self.ctx.memset_async(self.slab, 0, dev_bytes)
            self.cctx = _native.Context(self.ctx.device)         # the collation has a stream of its own (see below)
            gbytes = plan.world * B * self.REC * 4
            self._gathered_ptr = [self.ctx.malloc(gbytes), self.ctx.malloc(gbytes)]
            for g_ in self._gathered_ptr:
                self.ctx.memset_async(g_, 0, gbytes)
            self.halves = [(0, (B + 1) // 2), ((B + 1) // 2, B)] if B > 1 else [(0, B)]
            self.ctx.sync()
        elif self.distributed:
            # the exchange goes through torch.distributed: its tensors own the record slab
            import torch
            self.torch = torch
            dev = torch.device("cuda", self.ctx.device)
            self._slab_t = torch.zeros((self.NSLOT, self.REC), dtype=torch.float32, device=dev)
            self.slab = int(self._slab_t.data_ptr())
            # the collation has a stream of its own (a context of this package, seen by torch as an external
            # stream): on an extractor's stream the first half-round gather would hold back that extractor's
            # second half
            from . import _native
            self.cctx = _native.Context(self.ctx.device)
            self._cstream = torch.cuda.ExternalStream(self.cctx.stream, device=self.ctx.device)
            # gathered rounds, one buffer per round parity: the previous round's last record (the halo of
            # this round's first pair on rank 0) is read where it was gathered - no clone, no allocation
            self._gathered = [torch.zeros((plan.world * B, self.REC), dtype=torch.float32, device=dev) for _ in range(2)]
            self.halves = [(0, (B + 1) // 2), ((B + 1) // 2, B)] if B > 1 else [(0, B)]
            torch.cuda.synchronize()
        else:
            self.slab = self.ctx.malloc(dev_bytes)
            self.ctx.memset_async(self.slab, 0, dev_bytes)
        self.score = self.ctx.malloc(2 * B * K * 4)
        self._ij = self.ctx.malloc(2 * B * K * 8)
        self._msc = self.ctx.malloc(2 * B * K * 4)
        self._info = self.ctx.malloc(2 * B * 16)
        self.ctx.memset_async(self._info, 0, 2 * B * 16)
        self.ctx.sync()
        self.NBATCH = (B + self.P - 1) // self.P + 1            # (+1: a round without halo starts at pair 1)
        self.ev_ext = [[self.ctx.event() for _ in range(B)] for _ in range(2)]
        self.ev_batch = [[self.ctx.event() for _ in range(self.NBATCH)] for _ in range(2)]
        self.n_batches = [0, 0]                                  # batches enqueued in the last round of each parity
        self.ev_halo = [self.ctx.event(), self.ctx.event()]      # N > 1: halo slot (round & 1) is in place
        self.ev_collated = [self.ctx.event(), self.ctx.event()]  # N > 1: the all-gather of that parity has read its set
        self.have_halo = False
        self.rounds = 0
        self.batches = 0                                         # global batch counter (matcher round-robin)
        self.shared_map = None          # last collated round [world*B, REC] (torch tensor; torch.distributed path only)
        self.shared_map_ptr = 0         # ... its device address (both paths)

    # ---- record addressing (slot = index into the slab; set p holds slots p*B .. p*B + B-1)
    def rec_ptr(self, slot: int) -> int:
        return self.slab + slot * self.REC * 4

    def xy_ptr(self, slot: int) -> int:
        return self.rec_ptr(slot)

    def desc_ptr(self, slot: int) -> int:
        return self.rec_ptr(slot) + self.K * 2 * 4

    def count_ptr(self, slot: int) -> int:
        return self.rec_ptr(slot) + self.K * ROW * 4

    @property
    def last_set(self) -> int:
        return (self.rounds - 1) & 1

    @property
    def ij(self) -> int:                  # outputs of the last enqueued round
        return self._ij + self.last_set * self.plan.frames_per_rank * self.K * 8

    @property
    def msc(self) -> int:
        return self._msc + self.last_set * self.plan.frames_per_rank * self.K * 4

    @property
    def info(self) -> int:
        return self._info + self.last_set * self.plan.frames_per_rank * 16

    def round(self, frames_dev, H, W, C):
        """frames_dev: device pointer (int, or any object with data_ptr()) of this rank's chunk,
        uint8 [B, H, W, C].  Enqueues B extracts + the batched matches (+ the collation when
        world > 1); returns without synchronising.  Round r uses record set r & 1: its extracts wait
        only for the readers of that set (the matches and collation of round r-2, and pair 0 of
        round r-1, which reads the last record of round r-2), so they overlap the matches of r-1."""
        plan, B, K = self.plan, self.plan.frames_per_rank, self.K
        NE, NM, P = len(self.dets), len(self.mats), self.P
        base = frames_dev if isinstance(frames_dev, int) else int(frames_dev.data_ptr())
        fbytes = int(H) * int(W) * int(C)
        rnd = self.rounds
        p = rnd & 1
        s_base = p * B                                   # first slot of this round's set
        halo_slot = 2 * B + p
        single = not self.distributed
        for d in self.dets:
            if rnd >= 2:
                for j in range(self.n_batches[p]):
                    d.ctx.wait(self.ev_batch[p][j])
                if not single:
                    d.ctx.wait(self.ev_collated[p])
            if rnd >= 1 and self.n_batches[1 - p]:
                d.ctx.wait(self.ev_batch[1 - p][0])
        EF = self.EF
        if EF == 1:
            for s in range(B):
                d = self.dets[s % NE]
                d.extract_dev(base + s * fbytes, H, W, C, self.xy_ptr(s_base + s), self.desc_ptr(s_base + s),
                              self.score + (s_base + s) * K * 4, self.count_ptr(s_base + s), max_kpts=K)
                d.ctx.record(self.ev_ext[p][s])
        else:
            # batched extractors: chunk c = frames [c EF, (c + 1) EF) goes through ONE launch sequence on extractor
            # c % NE (sslam_aliked_extract_batch_dev); every frame of the chunk gets its event behind it
            for c, lo in enumerate(range(0, B, EF)):
                fr = range(lo, min(B, lo + EF))
                d = self.dets[c % NE]
                d.extract_batch_dev([base + s * fbytes for s in fr], H, W, C, [self.xy_ptr(s_base + s) for s in fr],
                                    [self.desc_ptr(s_base + s) for s in fr],
                                    [self.score + (s_base + s) * K * 4 for s in fr],
                                    [self.count_ptr(s_base + s) for s in fr], max_kpts=K)
                for s in fr:
                    d.ctx.record(self.ev_ext[p][s])
        have_halo = self.have_halo
        prev_slot = (1 - p) * B + B - 1                  # last frame of the previous round (one GPU)
        if not single and self.comm is not None:
            # ---- multi-GPU, RCCL directly: the same choreography as below on raw device memory
            G = self._gathered_ptr[p]
            rb = self.REC * 4
            prev = plan.rank * B - 1
            if prev < 0:
                # rank 0: the halo is the LAST record of the previous round's collation - in place already, so the copy
                # (and the halo event the first batch waits for) goes in front of this round's gathers
                if have_halo:
                    self.cctx.d2d_async(self.rec_ptr(halo_slot), self._gathered_ptr[1 - p] + (plan.world * B - 1) * rb, rb)
                self.cctx.record(self.ev_halo[p])
            for (lo, hi) in self.halves:
                for s in range(lo, hi):
                    self.cctx.wait(self.ev_ext[p][s])
                self.comm.all_gather_rows(self.cctx.stream, self.rec_ptr(s_base), G, B, lo, hi, rb)
            self.shared_map_ptr = G
            if prev >= 0:
                # other ranks: the last frame of the neighbour, gathered in this round's second half
                self.cctx.d2d_async(self.rec_ptr(halo_slot), G + prev * rb, rb)
                have_halo = True
                self.cctx.record(self.ev_halo[p])
            self.cctx.record(self.ev_collated[p])
            prev_slot = halo_slot
        elif not single:
            # ---- multi-GPU: collate each HALF of the round as soon as its extracts are done (the first
            # all-gather runs under the second half's extracts, the second under the matches), on the
            # collation stream; the matches run on their own streams underneath.
            torch = self.torch
            G = self._gathered[p]
            if rnd >= 2:
                # G was last written two rounds ago; its readers since: that round's halo copy (same stream) and
                # the NEXT round's halo copy on rank 0 (same stream too) - stream order covers both
                pass
            prev = plan.rank * B - 1                         # index inside the gathered round
            with torch.cuda.stream(self._cstream):
                if prev < 0:
                    # rank 0: the halo is the LAST record of the previous round's collation - in place already, so the
                    # copy (and the halo event the first batch waits for) goes in front of this round's gathers
                    if have_halo:
                        self._slab_t[halo_slot].copy_(self._gathered[1 - p][plan.world * B - 1])
                    self.cctx.record(self.ev_halo[p])
                for (lo, hi) in self.halves:
                    for s in range(lo, hi):
                        self.cctx.wait(self.ev_ext[p][s])
                    collate(self._slab_t[s_base:s_base + B], plan, self.group, out=G, part=(lo, hi), always=True)
                self.shared_map = G
                self.shared_map_ptr = int(G.data_ptr())
                if prev >= 0:
                    # other ranks: the last frame of the neighbour, gathered in this round's second half
                    self._slab_t[halo_slot].copy_(G[prev])
                    have_halo = True
                    self.cctx.record(self.ev_halo[p])
            self.cctx.record(self.ev_collated[p])
            prev_slot = halo_slot
        # ---- batched matches: pair s = (s-1, s); pair 0 = (previous frame, 0)
        out_base = p * B
        s0 = 0 if have_halo else 1
        spans = []
        while s0 < B:
            s1 = min(B, (s0 // P + 1) * P)
            spans.append((s0, s1))
            s0 = s1
        if not single and plan.rank > 0 and len(spans) > 1 and spans[0][0] == 0:
            # multi-GPU: the batch with the halo pair (previous frame = a neighbour's record, in place only after this
            # round's all-gathers) goes LAST; the others need local extracts only and start under the collation
            # (r03: enqueued first, it held every matcher stream back until the whole round was extracted and gathered -
            # 957 -> see DESIGN section 7)
            spans = spans[1:] + spans[:1]
        j = 0
        for s0, s1 in spans:
            m = self.mats[self.batches % NM]
            pairs = []
            for s in range(s0, s1):
                a = prev_slot if s == 0 else s_base + s - 1
                b = s_base + s
                pairs.append((self.xy_ptr(a), self.desc_ptr(a), K, self.xy_ptr(b), self.desc_ptr(b), K,
                              self.count_ptr(a), self.count_ptr(b)))
                m.ctx.wait(self.ev_ext[p][s])
            if s0 == 0:
                # the previous frame's record: the last extract of the previous round (one GPU) or
                # this round's collation (N > 1)
                m.ctx.wait(self.ev_ext[1 - p][B - 1] if single else self.ev_halo[p])
            else:
                m.ctx.wait(self.ev_ext[p][s0 - 1])
            m.match_batch_dev(pairs, self._ij + (out_base + s0) * K * 8, self._msc + (out_base + s0) * K * 4,
                              self._info + (out_base + s0) * 16, K, min_conf=self.min_conf)
            m.ctx.record(self.ev_batch[p][j])
            self.batches += 1
            j += 1
        self.n_batches[p] = j
        self.have_halo = True
        self.rounds += 1

    def sync(self):
        for c in {id(x.ctx): x.ctx for x in self.dets + self.mats}.values():
            c.sync()
        if self.cctx is not None:
            self.cctx.sync()

    def _checked_infos(self):
        """[B, 4] int32 {matches, layers, n0, n1} of the last round.  A match count of -1 is the matcher's
        verdict that a finite activation left the fp16 range of the split-precision path while that pair
        was processed (csrc/gemm_f16x3.hpp: its matches are not fp32-grade): never handed on silently."""
        self.sync()
        info = np.empty((self.plan.frames_per_rank, 4), np.int32)
        self.ctx.d2h(info, self.info)
        bad = np.flatnonzero(info[:, 0] < 0)
        if len(bad):
            for m in self.mats:
                m.range_overflow()                           # reported here: clear the instances' sticky words
            raise RangeOverflowError(
                f"LightGlue split-precision range overflow in pair(s) {bad.tolist()} of the last round (|activation| >= "
                f"65520 does not fit the fp16 planes): rescale the descriptors or run the matchers with set_precision('f32')")
        return info

    def results(self):
        """Host copy of the last round's matches: list of (ij [K,2], scores [K]) per local frame."""
        info = self._checked_infos()
        B, K = self.plan.frames_per_rank, self.K
        ij = np.empty((B, K, 2), np.int32); sc = np.empty((B, K), np.float32)
        self.ctx.d2h(ij, self.ij); self.ctx.d2h(sc, self.msc)
        return [(ij[s, :info[s, 0]].copy(), sc[s, :info[s, 0]].copy()) for s in range(B)]

    def infos(self):
        return self._checked_infos()

    def range_overflow(self) -> bool:
        """True if any matcher or extractor of the pipeline raised its range flag since the last poll (every round, not
        only the last one whose `info` is still on the device); synchronises their streams, clears."""
        return any([m.range_overflow() for m in self.mats] + [d.range_overflow() for d in self.dets])

    def features(self):
        """Host copy of the last round's features: list of (xy [n,2], desc [n,128]) per local frame."""
        self.sync()
        B = self.plan.frames_per_rank
        slab = np.empty((B, self.REC), np.float32)
        self.ctx.d2h(slab, self.rec_ptr(self.last_set * B))
        return [unpack_record(slab[s], self.K)[1:] for s in range(B)]
