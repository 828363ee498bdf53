"""Frame-sharded extract + match pipeline (one process per GPU).

The reference runs one frame at a time in one process
(slam/monocular/main_revamped.py:321-328: feature_extractor(frame t) then
feature_matcher(t-1 -> t)).  Frames are independent for extraction and pairs
are independent for matching, so a stream shards across the GPUs of a node
with no data-path collective: in every round rank r owns the contiguous chunk

    frames [ (round * world + r) * B , ... + B )

and matches each of its frames against the previous one.  The only exchange
is the collation of the per-frame feature records of all frames back into
the shared map every rank keeps (RCCL all-gather over xGMI, driven directly:
rccl.py); the pair that straddles a chunk boundary is matched after that
gather against the neighbour's last frame.

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

The pipeline runs on the C-ABI alone at every N (streams, events, buffers through `_native.Context`); the exchange is a
`comm` object with ONE method, `all_gather` - `rccl.RcclComm` (one `ncclAllGather` per half round) in production, `GlooRowsComm` (torch.distributed over
gloo, rows through the host) where ranks share a GPU or there is none: the choreography around it is one code path.
`ShardPlan`, the record helpers and `GlooRowsComm` are covered by the world_size-2 gloo tests on CPU.
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


class GlooRowsComm:
    """The collation's exchange over torch.distributed (gloo, host memory), with `rccl.RcclComm.all_gather_rows`'s signature:
    for ranks that SHARE a GPU (RCCL wants one device per rank) and for the CPU tests.  The rows make a round trip through
    the host, synchronously behind whatever is already enqueued on the collation context - slow, and the same data in the
    same places as the RCCL form, so `FrameStreamPipeline.round` is one code path for both."""

    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = int(rank), int(world), group

    def count(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def all_gather(self, cctx, src_ptr: int, dst_ptr: int, nbytes: int):
        """THE exchange of the pipeline: every rank contributes `nbytes` at src_ptr; rank r's land at dst_ptr + r * nbytes on
        every rank (one contiguous all-gather).  `cctx`: the context whose stream the exchange is ordered on."""
        if nbytes <= 0:
            return
        import torch
        import torch.distributed as dist
        send = np.empty(nbytes, np.uint8)
        cctx.d2h(send, src_ptr)                                  # (waits for the stream: the extracts' events were enqueued on it)
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, torch.from_numpy(send), group=self.group)
        for r, t in enumerate(parts):
            cctx.h2d(dst_ptr + r * nbytes, t.numpy())

    def all_gather_rows(self, cctx, src_ptr: int, dst_ptr: int, rows_per_rank: int, lo: int, hi: int, row_bytes: int):
        """The rank-major form for a caller that wants PART of every rank's block in frame order: every rank contributes rows
        lo .. hi-1 of its `rows_per_rank` local rows (src_ptr = row 0 of the local block); they land in rows
        r * rows_per_rank + lo .. of dst_ptr on every rank.  (Not what the pipeline uses: its gathered round is laid out so
        that each half is one `all_gather`.)"""
        if hi <= lo:
            return
        import torch
        import torch.distributed as dist
        n = (hi - lo) * row_bytes
        send = np.empty(n, np.uint8)
        cctx.d2h(send, src_ptr + lo * row_bytes)
        parts = [torch.empty(n, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, torch.from_numpy(send), group=self.group)
        for r, t in enumerate(parts):
            cctx.h2d(dst_ptr + (r * rows_per_rank + lo) * row_bytes, t.numpy())

    def close(self):
        pass


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
                 batch_pairs: int | None = None, use_graphs: bool = True, collate_always: bool = False, comm=None):
        self.dets = list(detectors) if isinstance(detectors, (list, tuple)) else [detectors]
        self.mats = list(matchers) if isinstance(matchers, (list, tuple)) else [matchers]
        self.plan = plan
        # comm: the exchange of the collation - `rccl.RcclComm` (RCCL over xGMI on memory of the C-ABI, no tensor library in
        # the data path) or `GlooRowsComm` (ranks sharing a GPU / tests); needed when world > 1
        # collate_always: take the multi-GPU path (collation stream, per-half all-gathers, halo record) with ONE rank too -
        # how the tests run the RCCL branch on a single GPU
        self.distributed = plan.world > 1 or bool(collate_always)
        if self.distributed and comm is None:
            raise ValueError("a frame-sharded pipeline over more than one rank needs a `comm` (rccl.RcclComm, or GlooRowsComm "
                             "for ranks that share a GPU)")
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
        self.cctx = None
        self.slab = self.ctx.malloc(dev_bytes)
        self.ctx.memset_async(self.slab, 0, dev_bytes)
        if self.distributed:
            # the collation has a stream of its own (a context of the same kind as the extractors'): on an extractor's stream
            # the first half-round gather would hold back that extractor's second half
            self.cctx = type(self.ctx)(self.ctx.device)
            # gathered rounds, one buffer per round parity: the previous round's last record (the halo of this round's
            # first pair on rank 0) is read where it was gathered - no copy of the map, no allocation.  Layout
            # [half][rank][rows of that half] (`map_row`): each half of a round is then ONE contiguous all-gather
            # (ncclAllGather: on the full xGMI mesh every rank's block goes straight to every peer, one hop)
            gbytes = plan.world * B * self.REC * 4
            self._gathered_ptr = [self.ctx.malloc(gbytes), self.ctx.malloc(gbytes)]
            for g_ in self._gathered_ptr:
                self.ctx.memset_async(g_, 0, gbytes)
        self.halves = [(0, (B + 1) // 2), ((B + 1) // 2, B)] if B > 1 else [(0, B)]
        self.ctx.sync()
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
        self.shared_map_ptr = 0         # device address of the last collated round [world*B, REC] in `map_row` order (N > 1)

    def map_row(self, frame_in_round: int) -> int:
        """Row of the collated round (`shared_map_ptr`) that holds frame `frame_in_round` = rank * B + slot of the round: the map
        is laid out [half][rank][rows of that half] so that each half-round collation is one contiguous all-gather."""
        B, world = self.plan.frames_per_rank, self.plan.world
        r, s = divmod(int(frame_in_round), B)
        for lo, hi in self.halves:
            if lo <= s < hi:
                return world * lo + r * (hi - lo) + (s - lo)
        raise IndexError(frame_in_round)

    def map_rows(self):
        """`map_row` of every frame of a round, in frame order: `shared_map[pipe.map_rows()]` is the round in frame order."""
        return np.array([self.map_row(j) for j in range(self.plan.world * self.plan.frames_per_rank)], np.int64)

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
        if not single:
            # ---- multi-GPU: collate each HALF of the round as soon as its extracts are done (the first all-gather runs
            # under the second half's extracts, the second under the matches) on the collation stream; the matches run on
            # their own streams underneath.  (G was last written two rounds ago; its readers since - that round's halo copy
            # and, on rank 0, the next round's - are on this same stream.)
            G = self._gathered_ptr[p]
            rb = self.REC * 4
            prev = plan.rank * B - 1                         # index inside the gathered round
            if prev < 0:
                # rank 0: the halo is the LAST record of the previous round's collation - in place already, so the copy
                # (and the halo event the first batch waits for) goes in front of this round's gathers
                if have_halo:
                    self.cctx.d2d_async(self.rec_ptr(halo_slot),
                                        self._gathered_ptr[1 - p] + self.map_row(plan.world * B - 1) * rb, rb)
                self.cctx.record(self.ev_halo[p])
            for (lo, hi) in self.halves:
                for s in range(lo, hi):
                    self.cctx.wait(self.ev_ext[p][s])
                # this half of every rank's records -> block [half] of the map, rank-major: ONE all-gather
                self.comm.all_gather(self.cctx, self.rec_ptr(s_base + lo), G + plan.world * lo * rb, (hi - lo) * rb)
            self.shared_map_ptr = G
            if prev >= 0:
                # other ranks: the last frame of the neighbour, gathered in this round's second half
                self.cctx.d2d_async(self.rec_ptr(halo_slot), G + self.map_row(prev) * rb, rb)
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
            # 957 -> see HISTORY.md section 7)
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
        was processed (csrc/gemm_f16x3.hpp: its matches are not fp32-grade): never handed on silently.  Neither is a frame
        an EXTRACTOR voided (keypoint count -1 in its record, which the matcher clamps to an empty frame): every extractor's
        sticky range word is polled here."""
        self.sync()
        info = np.empty((self.plan.frames_per_rank, 4), np.int32)
        self.ctx.d2h(info, self.info)
        if any([d.range_overflow() for d in self.dets]):             # (every instance polled: the poll clears its sticky word)
            # an extractor voided a frame (count -1 in its record; the matcher reads such a frame as EMPTY, so `info` shows 0
            # matches, not -1): name the frames of the round that is still on the device
            B, void = self.plan.frames_per_rank, []
            cnt = np.empty(4, np.int32)
            for s in range(B):
                self.ctx.d2h(cnt, self.count_ptr(self.last_set * B + s))
                if cnt[0] < 0:
                    void.append(s)
            raise RangeOverflowError(
                f"ALIKED split-precision range overflow: the features of frame(s) {void if void else '(of an earlier round)'} "
                f"are void (|activation| >= 65520 does not fit the fp16 planes) and so are the matches against them")
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
