"""RCCL (librccl.so of the ROCm install) driven directly through ctypes: the collation of the frame-sharded pipeline
without a tensor library in the data path.

`torch.distributed`'s `nccl` backend is RCCL too, but a process that initialises torch's GPU side runs on the HIP
runtime bundled with the torch wheel (same SONAME as the system one: whichever loads first serves the whole process);
measured on one MI355X, that alone costs the pipeline 3 - 4 % (bench: 992 -> 961 frames/s with torch initialised
first), the process group and the torch-side collation another 3 %.  Here the ranks keep the system runtime their
kernels were built against: records live in memory of the C-ABI, the gathers are `ncclAllGather` calls enqueued on
the collation stream of the pipeline (one per half round: frame_shard.py lays the gathered round out [half][rank][rows]), and torch - if it is there at all - only carries the 128-byte communicator id
between the ranks over gloo (CPU).

    comm = RcclComm.create(rank, world, exchange)     # exchange(bytes | None) -> bytes: rank 0's id to everybody
    comm.all_gather(cctx, src_ptr, dst_ptr, nbytes)      # cctx: the _native.Context of the collation stream; dst block r = rank r's bytes
"""
from __future__ import annotations

import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128
_NCCL_FLOAT32 = 7


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("SSLAM_RCCL_LIB", "/opt/rocm/lib/librccl.so")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ncclCommCount.restype = C.c_int
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        L.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd",
                  "ncclBroadcast", "ncclAllGather"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def new_unique_id() -> bytes:
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES))


class RcclComm:
    """One communicator of `world` ranks (one GPU per rank; the device must be current in the calling thread, which it
    is once a Context of this package exists for it)."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("an ncclUniqueId is 128 bytes")
        uid = _UniqueId()
        C.memmove(C.addressof(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        h = C.c_void_p()
        _check(lib().ncclCommInitRank(C.byref(h), int(world), uid, int(rank)), "ncclCommInitRank")
        self.handle, self.rank, self.world = h, int(rank), int(world)

    @classmethod
    def create(cls, rank: int, world: int, exchange):
        """`exchange(payload)`: called with rank 0's id bytes on rank 0 and with None elsewhere; returns rank 0's bytes
        on every rank (e.g. a `torch.distributed.broadcast_object_list` over gloo, or a file / socket)."""
        uid = exchange(new_unique_id() if rank == 0 else None)
        return cls(rank, world, uid)

    def count(self) -> int:
        """Number of ranks, as the communicator itself reports it (ncclCommCount)."""
        n = C.c_int()
        _check(lib().ncclCommCount(self.handle, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def all_gather(self, cctx, src_ptr: int, dst_ptr: int, nbytes: int):
        """THE exchange of the pipeline's collation: every rank contributes `nbytes` (a multiple of 4) at src_ptr; rank r's
        land at dst_ptr + r * nbytes on every rank - ONE `ncclAllGather`, enqueued on the stream of `cctx` (a
        `_native.Context`; or a raw hipStream_t as an integer)."""
        if nbytes <= 0:
            return
        if nbytes % 4:
            raise ValueError("all_gather: float32 words are exchanged (nbytes must be a multiple of 4)")
        stream = cctx if isinstance(cctx, int) else int(cctx.stream)
        _check(lib().ncclAllGather(C.c_void_p(src_ptr), C.c_void_p(dst_ptr), nbytes // 4, _NCCL_FLOAT32, self.handle,
                                   C.c_void_p(stream)), "ncclAllGather")

    def all_gather_rows(self, cctx, src_ptr: int, dst_ptr: int, rows_per_rank: int, lo: int, hi: int, row_bytes: int):
        """The rank-major form for a caller that wants PART of every rank's block in frame order (a ragged split: the
        pipeline itself never needs it, its map is laid out so that each half round is one `all_gather`): every rank
        contributes rows lo .. hi-1 of its `rows_per_rank` local rows (src_ptr = row 0 of the local block); they land in
        rows r * rows_per_rank + lo .. of dst_ptr on every rank.  Whole blocks are one `ncclAllGather`; a part is a group
        of `ncclBroadcast`s (ring / tree traffic - avoid on the data path).  float32 rows."""
        if hi <= lo:
            return
        stream = cctx if isinstance(cctx, int) else int(cctx.stream)
        L = lib()
        count = (hi - lo) * row_bytes // 4
        if lo == 0 and hi == rows_per_rank:                    # whole blocks: one all-gather, rank-major = frame order
            _check(L.ncclAllGather(C.c_void_p(src_ptr), C.c_void_p(dst_ptr), count, _NCCL_FLOAT32, self.handle,
                                   C.c_void_p(stream)), "ncclAllGather")
            return
        send = src_ptr + lo * row_bytes
        _check(L.ncclGroupStart(), "ncclGroupStart")
        try:
            for r in range(self.world):                        # rank r's part, broadcast into its rows on everybody
                recv = dst_ptr + (r * rows_per_rank + lo) * row_bytes
                _check(L.ncclBroadcast(C.c_void_p(send), C.c_void_p(recv), count, _NCCL_FLOAT32, r, self.handle,
                                       C.c_void_p(stream)), "ncclBroadcast")
        finally:
            _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        if getattr(self, "handle", None):
            lib().ncclCommDestroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
