"""ctypes binding of the C ABI in include/ucoslam_hip.h (libucoslam_hip.so, built in-tree by build.py).

There is no CPU fallback: if the shared library is missing or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libucoslam_hip.so")

UH_OK, UH_EINVAL, UH_ENODEVICE, UH_ENOTBUILT, UH_ENOMEM, UH_ECAPACITY = 0, -1, -2, -3, -4, -5


class UcoslamHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Load the HIP library (importing torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UcoslamHipError(UH_ENODEVICE, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is the product; there is no CPU path)")
    try:
        import torch  # noqa: F401  (loads libamdhip64 once for the whole process)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    _declare(L)
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise UcoslamHipError(rc, lib().uh_last_error().decode("utf-8", "replace"))


VP, I, SZ = C.c_void_p, C.c_int, C.c_size_t


def _declare(L: C.CDLL):
    def sig(name, restype, *argtypes):
        fn = getattr(L, name)
        fn.restype = restype
        fn.argtypes = list(argtypes)

    sig("uh_last_error", C.c_char_p)
    sig("uh_version", I)
    sig("uh_ctx_create", I, I, VP, C.POINTER(VP))
    sig("uh_ctx_create_private", I, I, C.POINTER(VP))
    sig("uh_ctx_create_private_cus", I, I, I, I, C.POINTER(VP))
    sig("uh_ctx_destroy", None, VP)
    sig("uh_ctx_synchronize", I, VP)
    sig("uh_ctx_stream", VP, VP)
    sig("uh_prof_enable", I, VP, I)
    sig("uh_prof_reset", I, VP)
    sig("uh_prof_report", I, VP, C.c_char_p, SZ)
    # kNN
    sig("uh_knn_create", I, VP, C.POINTER(VP))
    sig("uh_knn_destroy", None, VP)
    sig("uh_knn_build", I, VP, VP, I, SZ, I)
    sig("uh_knn_build_dev", I, VP, VP, I)
    sig("uh_knn_set_shard", I, VP, I, I)
    sig("uh_knn_size", I, VP)
    sig("uh_knn_set_queries_per_wave", I, VP, I)
    sig("uh_knn_search", I, VP, VP, I, SZ, I, VP, VP, I, I)
    sig("uh_knn_search_dev", I, VP, VP, I, I, VP, VP, I, I)
    sig("uh_knn_build_kmeans", I, VP, VP, I, I, I)
    sig("uh_knn_kmeans_blob", I, VP, C.POINTER(VP), C.POINTER(C.c_uint64))
    sig("uh_knn_kmeans_build_host", I, VP, I, I, I, VP, C.c_uint64, C.POINTER(C.c_uint64))
    sig("uh_knn_search_kmeans", I, VP, VP, I, I, I, I, VP, VP)
    sig("uh_knn_search_kmeans_dev", I, VP, VP, I, I, I, I, VP, VP)
    sig("uh_knn_scan_shard_dev", I, VP, VP, I, I, I, VP, VP, I)
    sig("uh_knn_replay_dev", I, VP, VP, I, I, I, I, VP, VP, I, I, VP, VP)
    sig("uh_knn_set_row_offset", I, VP, I)
    sig("uh_knn_set_valid_rows_dev", I, VP, VP)
    sig("uh_knn_to_stream", I, VP, VP, C.c_uint64, C.POINTER(C.c_uint64))
    sig("uh_knn_from_stream", I, VP, VP, C.c_uint64)
    sig("uh_knn_replay_tiles_dev", I, VP, VP, I, I, I, I, VP, VP, I, I, VP, VP, VP)
    for extra in _EXTRA_DECLS:
        extra(L, sig)


_EXTRA_DECLS = []


class Context:
    """uh_ctx: one GPU + one HIP stream. `stream` is a torch stream's `.cuda_stream` integer (0/None = the
    default stream); private=True makes the context create and own a non-blocking stream."""

    def __init__(self, device: int = 0, stream: int | None = None, private: bool = False, cus: tuple | None = None):
        self._h = VP()
        if cus is not None:   # (first mask bit, number of bits): a private stream on that share of the compute units
            check(lib().uh_ctx_create_private_cus(device, int(cus[0]), int(cus[1]), C.byref(self._h)))
        elif private:
            check(lib().uh_ctx_create_private(device, C.byref(self._h)))
        else:
            check(lib().uh_ctx_create(device, VP(stream) if stream else None, C.byref(self._h)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(lib().uh_ctx_synchronize(self._h))

    # per-kernel HIP-event timing on the context stream (measurement only)
    def prof_enable(self, on: bool = True):
        check(lib().uh_prof_enable(self._h, int(on)))

    def prof_reset(self):
        check(lib().uh_prof_reset(self._h))

    def prof_report(self) -> dict:
        """{kernel: (calls, total_ms)} accumulated since the last reset."""
        n = lib().uh_prof_report(self._h, None, 0)
        buf = C.create_string_buffer(max(n, 1) + 16)
        lib().uh_prof_report(self._h, buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.rsplit(" ", 2)
            out[name] = (int(calls), float(ms))
        return out

    def close(self):
        if self._h:
            lib().uh_ctx_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def np_ptr(a):
    return a.ctypes.data_as(VP)


def dev_ptr(t):
    """Device pointer of a torch CUDA tensor."""
    return VP(t.data_ptr())
