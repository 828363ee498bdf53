"""ctypes binding of libsslam_hip.so (include/sslam_hip.h).

There is NO CPU fallback: if the library is missing or no gfx950 device is
visible, the product path raises.  (The oracle under /oracle is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libsslam_hip.so"

c_void_pp = C.POINTER(C.c_void_p)
c_int_p = C.POINTER(C.c_int)
c_float_p = C.POINTER(C.c_float)


class NativeError(RuntimeError):
    pass


_lib = None


def _signatures():
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    return {
        "sslam_abi_version": (i32, []),
        "sslam_last_error": (C.c_char_p, []),
        "sslam_device_count": (i32, [c_int_p]),
        "sslam_ctx_create": (i32, [i32, vp, c_void_pp]),
        "sslam_ctx_destroy": (i32, [vp]),
        "sslam_ctx_sync": (i32, [vp]),
        "sslam_ctx_stream": (vp, [vp]),
        "sslam_timer_start": (i32, [vp]),
        "sslam_timer_stop": (i32, [vp, c_float_p]),
        "sslam_malloc": (i32, [vp, sz, c_void_pp]),
        "sslam_free": (i32, [vp, vp]),
        "sslam_memcpy_h2d": (i32, [vp, vp, vp, sz]),
        "sslam_memcpy_d2h": (i32, [vp, vp, vp, sz]),
        "sslam_memcpy_d2d_async": (i32, [vp, vp, vp, sz]),
        "sslam_host_alloc": (i32, [vp, sz, c_void_pp]),
        "sslam_host_free": (i32, [vp, vp]),
        "sslam_memcpy_h2d_async": (i32, [vp, vp, vp, sz]),
        "sslam_memcpy_d2h_async": (i32, [vp, vp, vp, sz]),
        "sslam_memset_async": (i32, [vp, vp, i32, sz]),
        "sslam_event_create": (i32, [vp, c_void_pp]),
        "sslam_event_destroy": (i32, [vp]),
        "sslam_event_record": (i32, [vp, vp]),
        "sslam_ctx_wait_event": (i32, [vp, vp]),
        "sslam_timing_event_create": (i32, [vp, c_void_pp]),
        "sslam_event_elapsed_ms": (i32, [vp, vp, c_float_p]),
        "sslam_ba_residual_jacobian_host": (i32, [vp, i32] + [vp] * 3 + [i32, vp, vp, i32, vp, vp] + [vp] * 4),
        "sslam_ba_residual_jacobian_dev": (i32, [vp, i32] + [vp] * 3 + [i32, vp, vp, i32, vp, vp] + [vp] * 4),
        "sslam_ba_solve_host": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, C.c_double, i32, vp]),
        "sslam_fmat_ransac_host": (i32, [vp, i32, vp, vp, C.c_double, C.c_double, i32, vp, vp, vp]),
        "sslam_fmat_ransac_dev": (i32, [vp, i32, vp, vp, vp, vp, C.c_double, C.c_double, i32, vp, vp, vp, vp]),
        "sslam_reproject_match_host": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, C.c_double, C.c_double,
                                             vp, vp, vp]),
        "sslam_reproject_match_dev": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, C.c_double, C.c_double,
                                            vp, vp, vp]),
        "sslam_aliked_create": (i32, [vp, vp, sz, i32, i32, i32, c_void_pp]),
        "sslam_aliked_create_batched": (i32, [vp, vp, sz, i32, i32, i32, i32, c_void_pp]),
        "sslam_aliked_extract_batch_dev": (i32, [vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
        "sslam_aliked_destroy": (i32, [vp]),
        "sslam_aliked_extract_host": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, c_int_p]),
        "sslam_aliked_extract_dev": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
        "sslam_aliked_debug_read": (i32, [vp, i32, vp, sz]),
        "sslam_aliked_use_graphs": (i32, [vp, i32]),
        "sslam_aliked_range_overflow": (i32, [vp, c_int_p]),
        "sslam_lightglue_use_graphs": (i32, [vp, i32]),
        "sslam_lightglue_create": (i32, [vp, vp, sz, i32, c_void_pp]),
        "sslam_lightglue_create_batched": (i32, [vp, vp, sz, i32, i32, c_void_pp]),
        "sslam_lightglue_destroy": (i32, [vp]),
        "sslam_lightglue_capacity": (i32, [vp, c_int_p]),
        "sslam_lightglue_batch_capacity": (i32, [vp, c_int_p]),
        "sslam_lightglue_match_batch_dev": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i32]),
        "sslam_lightglue_range_overflow": (i32, [vp, c_int_p]),
        "sslam_lightglue_debug_key_split": (i32, [vp, i32]),
        "sslam_lightglue_debug_big_gemm": (i32, [vp, i32]),
        "sslam_lightglue_debug_split_form": (i32, [vp, i32]),
        "sslam_lightglue_set_conf": (i32, [vp, f32, f32, f32, i32]),
        "sslam_lightglue_match_host": (i32, [vp, vp, vp, i32, vp, vp, i32, f32, vp, vp, c_int_p, c_int_p]),
        "sslam_lightglue_match_host_sized": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, f32, vp, vp, c_int_p, c_int_p]),
        "sslam_lightglue_match_dev": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, vp, f32, vp, vp, vp]),
        "sslam_lightglue_debug_read": (i32, [vp, i32, vp, sz]),
        "sslam_lightglue_debug_layers": (i32, [vp, i32, i32]),
        "sslam_lightglue_profile": (i32, [vp, i32]),
        "sslam_lightglue_set_precision": (i32, [vp, i32]),
        "sslam_lightglue_profile_read": (i32, [vp, c_float_p, c_int_p]),
    }


def _declare(lib):
    for name, (res, args) in _signatures().items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def declared_symbols():
    """Names this binding expects (tests compare with include/sslam_hip.h)."""
    return list(_signatures().keys())


def lib():
    """Load the shared library (once).  Raises NativeError if it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError(
                f"{LIB_PATH} not found - build it with `python opencv-simpleslam_amd/build.py` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        try:
            handle = C.CDLL(str(LIB_PATH))
        except OSError as e:
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        _declare(handle)
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().sslam_last_error()
        raise NativeError(f"{what or 'libsslam_hip'} failed (rc={rc}): "
                          f"{msg.decode() if msg else 'unknown error'}")


def ptr(a):
    """void* of a numpy array (must be C-contiguous) or an int device address or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array passed to the native library must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):          # torch tensor (device or host)
        return C.c_void_p(a.data_ptr())
    raise TypeError(f"cannot take a pointer of {type(a)}")


class Context:
    """One HIP device + stream (sslam_ctx).  Not thread-safe."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = lib()
        h = C.c_void_p()
        check(L.sslam_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)),
              "sslam_ctx_create")
        self.handle = h
        self.device = int(device)
        self.scratch = {}                 # named device buffers owned by this context (freed in close)
        self._pinned = []                 # page-locked host blocks (address, ctypes view) from host_alloc

    def close(self):
        if getattr(self, "handle", None):
            for buf in getattr(self, "scratch", {}).values():
                for ptr_ in buf.get("_ptrs", ()):
                    lib().sslam_free(self.handle, C.c_void_p(ptr_))
            self.scratch = {}
            for addr, _buf in getattr(self, "_pinned", ()):
                lib().sslam_host_free(self.handle, C.c_void_p(addr))
            self._pinned = []
            lib().sslam_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().sslam_ctx_sync(self.handle), "sslam_ctx_sync")

    @property
    def stream(self) -> int:
        return int(lib().sslam_ctx_stream(self.handle) or 0)

    def timer_start(self):
        check(lib().sslam_timer_start(self.handle), "sslam_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib().sslam_timer_stop(self.handle, C.byref(ms)), "sslam_timer_stop")
        return float(ms.value)

    # raw device memory (used by tests / bench when torch is not wanted)
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().sslam_malloc(self.handle, int(nbytes), C.byref(p)), "sslam_malloc")
        return int(p.value)

    def free(self, dptr: int):
        check(lib().sslam_free(self.handle, C.c_void_p(dptr)), "sslam_free")

    def h2d(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        check(lib().sslam_memcpy_h2d(self.handle, C.c_void_p(dptr), ptr(arr), arr.nbytes), "h2d")

    def d2h(self, arr: np.ndarray, dptr: int):
        check(lib().sslam_memcpy_d2h(self.handle, ptr(arr), C.c_void_p(dptr), arr.nbytes), "d2h")

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """Page-locked host memory as a uint8 array.  The block belongs to the CONTEXT: `Context.close()` frees it, after
        which the array and every view of it dangle - keep the context alive as long as they are used."""
        p = C.c_void_p()
        check(lib().sslam_host_alloc(self.handle, int(nbytes), C.byref(p)), "sslam_host_alloc")
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        arr = np.frombuffer(buf, np.uint8)
        self._pinned.append((p.value, buf))
        return arr

    def h2d_async(self, dptr: int, arr: np.ndarray):
        """Enqueue only; `arr` (contiguous, page-locked for a real overlap) must stay untouched until the stream passed it."""
        check(lib().sslam_memcpy_h2d_async(self.handle, C.c_void_p(dptr), ptr(arr), arr.nbytes), "h2d_async")

    def d2h_async(self, arr: np.ndarray, dptr: int, nbytes: int | None = None):
        check(lib().sslam_memcpy_d2h_async(self.handle, ptr(arr), C.c_void_p(dptr),
                                           arr.nbytes if nbytes is None else int(nbytes)), "d2h_async")

    def d2d_async(self, dst: int, src: int, nbytes: int):
        check(lib().sslam_memcpy_d2d_async(self.handle, C.c_void_p(dst), C.c_void_p(src), int(nbytes)), "d2d")

    def memset_async(self, dst: int, value: int, nbytes: int):
        check(lib().sslam_memset_async(self.handle, C.c_void_p(dst), int(value), int(nbytes)), "memset")

    def event(self) -> int:
        e = C.c_void_p()
        check(lib().sslam_event_create(self.handle, C.byref(e)), "sslam_event_create")
        return int(e.value)

    def timing_event(self) -> int:
        e = C.c_void_p()
        check(lib().sslam_timing_event_create(self.handle, C.byref(e)), "sslam_timing_event_create")
        return int(e.value)

    @staticmethod
    def elapsed_ms(start_event: int, stop_event: int) -> float:
        ms = C.c_float()
        check(lib().sslam_event_elapsed_ms(C.c_void_p(start_event), C.c_void_p(stop_event), C.byref(ms)), "sslam_event_elapsed_ms")
        return float(ms.value)

    def record(self, event: int):
        """Record `event` on this context's stream."""
        check(lib().sslam_event_record(self.handle, C.c_void_p(event)), "sslam_event_record")

    def wait(self, event: int):
        """This context's stream waits for `event` (the host does not block)."""
        check(lib().sslam_ctx_wait_event(self.handle, C.c_void_p(event)), "sslam_ctx_wait_event")

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        d = self.malloc(max(arr.nbytes, 1))
        if arr.nbytes:
            self.h2d(d, arr)
        return d


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().sslam_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    """Process-wide context per device (the reference creates its models once
    at slam/core/features_utils.py:18-30 and keeps them for the whole run)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
