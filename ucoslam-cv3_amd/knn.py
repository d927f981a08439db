"""Host-side mirror of xflann::Index (Linear, Hamming) on top of the C ABI.

Reference surface: 3rdparty/xflann/xflann/index.h:41-135 (`build`, `search`), params
`LinearParams(store_data)` / `KnnSearchParams(maxChecks, sorted, threads)` (types.h:83,152).
`maxChecks` and `threads` have no meaning for an exact GPU scan and are accepted but ignored.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VP, check, dev_ptr, lib, np_ptr


def shard_bounds(nt: int, nshards: int):
    """Contiguous split of the train rows used by every rank: shard s = [nt*s/n, nt*(s+1)/n)."""
    return [nt * s // nshards for s in range(nshards + 1)]


class Index:
    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_knn_create(ctx.handle, C.byref(self._h)))
        self._keep = None  # keeps device train rows alive for build_dev

    # -- xflann::Index::build(Matrix features, LinearParams) -------------------------------------
    def build(self, features, store_data: int = 1):
        if _is_torch(features):
            t = features
            assert t.is_cuda and t.dtype.is_floating_point is False and t.dim() == 2 and t.shape[1] == 32
            t = t.contiguous()
            self._keep = t
            check(lib().uh_knn_build_dev(self._h, dev_ptr(t), t.shape[0]))
        else:
            a = np.asarray(features)
            if a.dtype != np.uint8 or a.ndim != 2:
                raise _lib.UcoslamHipError(_lib.UH_EINVAL, "features must be a 2-D uint8 matrix")
            if a.shape[0] and a.strides[1] != 1:
                a = np.ascontiguousarray(a)
            stride = a.strides[0] if a.shape[0] else 32
            check(lib().uh_knn_build(self._h, np_ptr(a), a.shape[0], stride, a.shape[1]))
        return self

    def set_shard(self, begin: int, end: int):
        check(lib().uh_knn_set_shard(self._h, begin, end))

    def set_queries_per_wave(self, qpw: int):
        """1 (default), 2 or 4 queries per wave in the exact search: identical rows; fewer, register-tiled waves move less L1/L2 traffic,
        which is what latency-bound work on another stream (the local BA) needs — see uh_knn_set_queries_per_wave."""
        check(lib().uh_knn_set_queries_per_wave(self._h, qpw))
        return self

    def size(self) -> int:
        return lib().uh_knn_size(self._h)

    # -- xflann::Index::search(features, nn, indices, distances, KnnSearchParams) ----------------
    def search(self, queries, nn: int, sorted: bool = False, max_dist: int = -1):
        if _is_torch(queries):
            import torch

            q = queries.contiguous()
            nq = q.shape[0]
            idx = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
            dist = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
            check(lib().uh_knn_search_dev(self._h, dev_ptr(q), nq, nn, dev_ptr(idx), dev_ptr(dist), int(sorted), max_dist))
            return idx, dist
        q = np.asarray(queries)
        if q.dtype != np.uint8 or q.ndim != 2 or q.shape[1] != 32:
            raise _lib.UcoslamHipError(_lib.UH_EINVAL, "queries must be uint8 [nq,32]")
        if q.shape[0] and q.strides[1] != 1:
            q = np.ascontiguousarray(q)
        nq = q.shape[0]
        idx = np.empty((nq, nn), np.int32)
        dist = np.empty((nq, nn), np.int32)
        stride = q.strides[0] if nq else 32
        check(lib().uh_knn_search(self._h, np_ptr(q), nq, stride, nn, np_ptr(idx), np_ptr(dist), int(sorted), max_dist))
        return idx, dist

    # -- xflann::Index::build(features, HKMeansParams(k, maxIters)) / search(KnnSearchParams(maxChecks, sorted)) -------------
    def build_kmeans(self, features, k: int = 32, maxIters: int = 0):
        """The approximate index FrameMatcher_Flann uses (framematcher.cpp:213).  features: uint8 [n,32] on the host."""
        a = np.ascontiguousarray(np.asarray(features), np.uint8)
        if a.ndim != 2 or (a.shape[0] and a.shape[1] != 32):
            raise _lib.UcoslamHipError(_lib.UH_EINVAL, "features must be uint8 [n,32]")
        self._km_rows = a
        check(lib().uh_knn_build_kmeans(self._h, np_ptr(a) if a.shape[0] else None, a.shape[0], k, maxIters))
        return self

    def kmeans_blob(self) -> np.ndarray:
        data, size = VP(), C.c_uint64()
        check(lib().uh_knn_kmeans_blob(self._h, C.byref(data), C.byref(size)))
        if not size.value:
            return np.zeros(0, np.uint8)
        return np.frombuffer((C.c_char * size.value).from_address(data.value), np.uint8).copy()

    # -- xflann::Index::toStream / fromStream (index.cpp:153-188), k-means index only (Linear has none in the reference) ---------
    def toStream(self) -> bytes:
        size = C.c_uint64()
        check(lib().uh_knn_to_stream(self._h, None, 0, C.byref(size)))
        out = np.zeros(size.value, np.uint8)
        check(lib().uh_knn_to_stream(self._h, np_ptr(out), size.value, C.byref(size)))
        return out.tobytes()

    def fromStream(self, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        check(lib().uh_knn_from_stream(self._h, np_ptr(buf), len(buf)))
        return self

    def search_kmeans(self, queries, nn: int, maxChecks: int = 16, sorted: bool = False):
        if _is_torch(queries):
            import torch

            q = queries.contiguous()
            nq = q.shape[0]
            idx = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
            dist = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
            check(lib().uh_knn_search_kmeans_dev(self._h, dev_ptr(q), nq, nn, maxChecks, int(sorted), dev_ptr(idx), dev_ptr(dist)))
            return idx, dist
        q = np.ascontiguousarray(np.asarray(queries), np.uint8)
        if q.ndim != 2 or (q.shape[0] and q.shape[1] != 32):
            raise _lib.UcoslamHipError(_lib.UH_EINVAL, "queries must be uint8 [nq,32]")
        nq = q.shape[0]
        idx = np.empty((nq, nn), np.int32)
        dist = np.empty((nq, nn), np.int32)
        check(lib().uh_knn_search_kmeans(self._h, np_ptr(q) if nq else None, nq, nn, maxChecks, int(sorted), np_ptr(idx), np_ptr(dist)))
        return idx, dist

    # -- sharded path (multi-GPU row of the scope table) -------------------------------------------
    def scan_shard(self, queries, nn: int, cap: int, max_dist: int = -1):
        import torch

        q = queries.contiguous()
        nq = q.shape[0]
        cand = torch.empty((nq, cap), dtype=torch.int64, device=q.device)
        counts = torch.empty((nq,), dtype=torch.int32, device=q.device)
        check(lib().uh_knn_scan_shard_dev(self._h, dev_ptr(q), nq, nn, max_dist, dev_ptr(cand), dev_ptr(counts), cap))
        return cand, counts

    def replay(self, queries, nn: int, cand_all, counts_all, sorted: bool = False, max_dist: int = -1):
        """cand_all: [nshards, nq, cap] int64, counts_all: [nshards, nq] int32 (shard order = index order)."""
        import torch

        q = queries.contiguous()
        nq = q.shape[0]
        nshards, _, cap = cand_all.shape
        cand_all = cand_all.contiguous()
        counts_all = counts_all.contiguous()
        idx = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
        dist = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
        check(lib().uh_knn_replay_dev(self._h, dev_ptr(q), nq, nn, int(sorted), max_dist, dev_ptr(cand_all),
                                      dev_ptr(counts_all), nshards, cap, dev_ptr(idx), dev_ptr(dist)))
        return idx, dist

    def set_valid_rows(self, d_count=None):
        """Query rows [*d_count, nq) of the following scan_shard calls are not scanned and get empty lists (None: every row counts).
        d_count: a one-element int32 device tensor the caller keeps alive."""
        self._valid_rows = d_count
        check(lib().uh_knn_set_valid_rows_dev(self._h, dev_ptr(d_count) if d_count is not None else None))
        return self

    def set_row_offset(self, offset: int):
        """This index holds ONE tile of a sharded train set: row 0 has the global index `offset` (used by scan_shard)."""
        check(lib().uh_knn_set_row_offset(self._h, offset))
        return self

    def replay_tiles(self, queries, nn: int, cand_all, counts_all, sorted: bool = False, max_dist: int = -1):
        """replay() for ranks that hold only their own tile: never rescans; returns (indices, distances, overflow) where the 1-element
        int32 tensor `overflow` is non-zero if some gathered list exceeded its cap (retry the frame with a larger cap)."""
        import torch

        q = queries.contiguous()
        nq = q.shape[0]
        nshards, _, cap = cand_all.shape
        cand_all = cand_all.contiguous()
        counts_all = counts_all.contiguous()
        idx = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
        dist = torch.empty((nq, nn), dtype=torch.int32, device=q.device)
        overflow = torch.zeros((1,), dtype=torch.int32, device=q.device)
        check(lib().uh_knn_replay_tiles_dev(self._h, dev_ptr(q), nq, nn, int(sorted), max_dist, dev_ptr(cand_all),
                                            dev_ptr(counts_all), nshards, cap, dev_ptr(idx), dev_ptr(dist), dev_ptr(overflow)))
        return idx, dist, overflow

    def close(self):
        if self._h:
            lib().uh_knn_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kmeans_build_host(features, k: int = 32, maxIters: int = 0) -> np.ndarray:
    """Block data of HKMeansParams(k, maxIters) over `features` — the host-side build alone, no GPU."""
    a = np.ascontiguousarray(np.asarray(features), np.uint8)
    size = C.c_uint64()
    check(lib().uh_knn_kmeans_build_host(np_ptr(a), a.shape[0], k, maxIters, None, 0, C.byref(size)))
    out = np.zeros(size.value, np.uint8)
    check(lib().uh_knn_kmeans_build_host(np_ptr(a), a.shape[0], k, maxIters, np_ptr(out), size.value, C.byref(size)))
    return out


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")
