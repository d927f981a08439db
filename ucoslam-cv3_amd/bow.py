"""Host-side mirror of fbow::Vocabulary / fbow::fBow on top of the C ABI.

Reference surface (3rdparty/fbow/fbow/fbow.h:54-116): `readFromFile/fromStream`, `transform(features)` (L2-normalised bag),
`transform(features, level, fBow&, fBow2&)` (raw weights + features grouped by the node reached at `level`),
`getK/getDescSize/getDescType/size`, `fBow::score(a, b)`.  fBow = {word id: weight}, fBow2 = {node id: [feature indices]}.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib
from ._lib import I, SZ, VP, check, lib, np_ptr

PARAMS_FMT = "<50s2xII4xQQQQQiiI4x"          # fbow::Vocabulary::params, 120 bytes
assert struct.calcsize(PARAMS_FMT) == 120
STREAM_SIG = 55824124


def _declare(L, sig):
    sig("uh_bow_create", I, VP, C.POINTER(VP))
    sig("uh_bow_destroy", None, VP)
    sig("uh_bow_load", I, VP, VP, SZ)
    sig("uh_bow_set", I, VP, VP, VP)
    sig("uh_bow_get_params", I, VP, VP)
    sig("uh_bow_transform", I, VP, VP, I, SZ, I, I, VP, VP, VP, VP)
    sig("uh_bow_transform_dev", I, VP, VP, I, I, VP, VP, VP, VP)
    sig("uh_bow_score", C.c_double, VP, VP, I, VP, VP, I)
    sig("uh_bowdb_create", I, VP, C.POINTER(VP))
    sig("uh_bowdb_destroy", None, VP)
    sig("uh_bowdb_size", I, VP)
    sig("uh_bowdb_add", I, VP, C.c_uint32, VP, VP, I)
    sig("uh_bowdb_del", I, VP, C.c_uint32)
    sig("uh_bowdb_query", I, VP, VP, VP, I, VP, I, C.c_float, VP, VP, VP, I)


_lib._EXTRA_DECLS.append(_declare)


class fBow(dict):
    """std::map<uint32_t,float> (iteration in key order where it matters)."""

    @staticmethod
    def score(a: "fBow", b: "fBow") -> float:
        ka = np.array(sorted(a), np.uint32)
        kb = np.array(sorted(b), np.uint32)
        wa = np.array([a[k] for k in ka], np.float32)
        wb = np.array([b[k] for k in kb], np.float32)
        return float(lib().uh_bow_score(np_ptr(ka), np_ptr(wa), len(ka), np_ptr(kb), np_ptr(wb), len(kb)))


class Vocabulary:
    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_bow_create(ctx.handle, C.byref(self._h)))
        self._params = None

    def fromStream(self, data: bytes):
        buf = np.frombuffer(data, np.uint8)
        check(lib().uh_bow_load(self._h, np_ptr(buf), len(data)))
        self._read_params()
        return self

    def readFromFile(self, path: str):
        try:
            with open(path, "rb") as f:
                data = f.read()
        except OSError:
            raise RuntimeError("Vocabulary::readFromFile could not open:" + path)    # fbow.cpp:156
        return self.fromStream(data)

    def _read_params(self):
        raw = np.zeros(120, np.uint8)
        check(lib().uh_bow_get_params(self._h, np_ptr(raw)))
        f = struct.unpack(PARAMS_FMT, raw.tobytes())
        self._params = dict(desc_name=f[0].split(b"\0")[0].decode(), aligment=f[1], nblocks=f[2], desc_size_bytes_wp=f[3],
                            block_size_bytes_wp=f[4], feature_off_start=f[5], child_off_start=f[6], total_size=f[7],
                            desc_type=f[8], desc_size=f[9], m_k=f[10])

    def getK(self):
        return self._params["m_k"]

    def getDescSize(self):
        return self._params["desc_size"]

    def getDescType(self):
        return self._params["desc_type"]

    def size(self):
        return self._params["nblocks"] * self._params["m_k"]     # fbow.h: size() = blocks * k

    def _descend(self, features, level):
        f = np.asarray(features)
        if f.ndim != 2 or f.shape[0] == 0:
            raise RuntimeError("Vocabulary::transform No input data")                                   # fbow.cpp:52
        if f.dtype != np.uint8:
            raise RuntimeError("Vocabulary::transform features are of different type than vocabulary")  # fbow.cpp:53
        if f.strides[1] != 1:
            f = np.ascontiguousarray(f)
        n = f.shape[0]
        word, weight = np.empty(n, np.uint32), np.empty(n, np.float32)
        node, valid = np.empty(n, np.uint32), np.empty(n, np.uint8)
        check(lib().uh_bow_transform(self._h, np_ptr(f), n, f.strides[0], f.shape[1], level, np_ptr(word), np_ptr(weight), np_ptr(node),
                                     np_ptr(valid)))
        return word, weight, node, valid

    def transform_triplets(self, d_features, level: int):
        """The per-descriptor result of the tree descent for device-resident descriptors (torch uint8 CUDA [n,32]): int32 [n,4] =
        (word id, float weight bits, node id at `level`, node valid).  This is what the ranks of a sharded transform exchange
        ("gather (word, weight, node) triplets; host accumulates in feature order")."""
        import torch

        from ._lib import dev_ptr

        f = d_features.contiguous()
        n = f.shape[0]
        out = torch.zeros((n, 4), dtype=torch.int32, device=f.device)
        if n == 0:
            return out
        word = torch.empty(n, dtype=torch.int32, device=f.device)
        weight = torch.empty(n, dtype=torch.float32, device=f.device)
        node = torch.empty(n, dtype=torch.int32, device=f.device)
        valid = torch.empty(n, dtype=torch.uint8, device=f.device)
        check(lib().uh_bow_transform_dev(self._h, dev_ptr(f), n, level, dev_ptr(word), dev_ptr(weight), dev_ptr(node), dev_ptr(valid)))
        out[:, 0] = word; out[:, 1] = weight.view(torch.int32); out[:, 2] = node; out[:, 3] = valid.to(torch.int32)
        return out

    @staticmethod
    def maps_from_triplets(trip):
        """(fBow, fBow2) of transform(features, level) from the [n,4] int32 triplets of ALL descriptors in feature order."""
        t = np.ascontiguousarray(np.asarray(trip), np.int32)
        word, weight = t[:, 0].view(np.uint32), t[:, 1].copy().view(np.float32)
        node, valid = t[:, 2].view(np.uint32), t[:, 3]
        r1 = fBow()
        for w, wt in zip(word.tolist(), weight):
            if w != 0xFFFFFFFF:
                r1[w] = np.float32(r1.get(w, np.float32(0)) + wt)
        r2 = {}
        for i, (nd, ok) in enumerate(zip(node.tolist(), valid)):
            if ok:
                r2.setdefault(nd, []).append(i)
        return r1, r2

    def transform(self, features, level: int | None = None):
        """transform(features) -> fBow (L2-normalised); transform(features, level) -> (fBow, fBow2) raw weights."""
        word, weight, node, valid = self._descend(features, -1 if level is None else level)
        r1 = fBow()
        for w, wt in zip(word.tolist(), weight):
            if w != 0xFFFFFFFF:
                r1[w] = np.float32(r1.get(w, np.float32(0)) + wt)        # float accumulation in feature order
        if level is None:
            norm = 0.0
            for v in r1.values():
                norm += float(v) * float(v)
            if norm > 0.0:
                inv = 1.0 / np.sqrt(norm)
                for k in r1:
                    r1[k] = np.float32(float(r1[k]) * inv)               # e.second *= inv_norm (float *= double)
            return r1
        r2 = {}
        for i, (nd, ok) in enumerate(zip(node.tolist(), valid)):
            if ok:
                r2.setdefault(nd, []).append(i)
        return r1, r2

    def close(self):
        if self._h:
            lib().uh_bow_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KPFrameDataBase:
    """ucoslam::KPFrameDataBase (src/map_types/keyframedatabase.cpp:150-275) over uh_bowdb_*: the keyframes' bags of words live
    in HBM; add(frame_id, bow) / delete(frame_id) / relocalizationCandidates(bow, covis_neighbors, ...).  `bow` is the fBow of
    Vocabulary.transform(desc, 3) (a dict word -> float32 weight).  `covis_neighbors(frame_id)` returns the (neighbour id, weight)
    list of CovisGraph::getNeighborsWeights(id, true): sorted by decreasing weight."""

    def __init__(self, ctx: _lib.Context):
        self._h = VP()
        check(lib().uh_bowdb_create(ctx.handle, C.byref(self._h)))

    @staticmethod
    def _arrays(bow):
        ks = np.fromiter(sorted(bow), np.uint32, len(bow))
        return ks, np.array([bow[int(k)] for k in ks], np.float32)

    def add(self, frame_id: int, bow):
        ks, ws = self._arrays(bow)
        check(lib().uh_bowdb_add(self._h, frame_id, np_ptr(ks) if len(ks) else None, np_ptr(ws) if len(ks) else None, len(ks)))

    def delete(self, frame_id: int):
        check(lib().uh_bowdb_del(self._h, frame_id))

    def size(self):
        return lib().uh_bowdb_size(self._h)

    def scoredFrames(self, bow, minScore=0.0, excludedFrames=()):
        """relocalizationCandidates up to `frame_score` (:195-238): [(frame id, common words, score)] in ascending id order."""
        ks, ws = self._arrays(bow)
        ex = np.array(sorted(excludedFrames), np.uint32)
        cap = max(self.size(), 1)
        ids, nobs, sc = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
        n = lib().uh_bowdb_query(self._h, np_ptr(ks) if len(ks) else None, np_ptr(ws) if len(ks) else None, len(ks),
                                 np_ptr(ex) if len(ex) else None, len(ex), float(minScore), np_ptr(ids), np_ptr(nobs), np_ptr(sc), cap)
        if n < 0:
            check(n)
        return [(int(ids[i]), int(nobs[i]), float(sc[i])) for i in range(n)]

    def relocalizationCandidates(self, bow, covis_neighbors, sorted_=True, minScore=0.0, excludedFrames=()):
        fs = self.scoredFrames(bow, minScore, excludedFrames)
        if len(fs) == 0:
            return []
        if len(fs) == 1:
            return [fs[0][0]]
        frame_score = {f: s for f, _, s in fs}
        acc, best = [], np.float64(np.float32(minScore))              # double bestAccScore = minScore (a float)
        for f, _, s in fs:                                           # :241-256, frames in ascending id (std::map order)
            a = s
            for nb, _w in list(covis_neighbors(f))[:10]:
                if nb in frame_score:
                    a += frame_score[nb]
            acc.append((f, a))
            if a > best:
                best = a
        keep = np.float64(np.float32(0.75)) * best                   # 0.75f * bestAccScore
        acc = [fa for fa in acc if not fa[1] < keep]
        if sorted_:
            acc = _stable_desc(acc)
        return [f for f, _ in acc]

    def close(self):
        if self._h:
            lib().uh_bowdb_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stable_desc(pairs):
    """std::sort with a.second > b.second is not stable; equal accumulated scores are vanishingly rare for real bags, and the
    reference's own order among them is unspecified — a stable sort is one of the valid outcomes."""
    return sorted(pairs, key=lambda fa: -fa[1])


def write_vocabulary_stream(params120: bytes, blob: bytes) -> bytes:
    """Vocabulary::toStream (fbow.cpp:171-179)."""
    return struct.pack("<Q", STREAM_SIG) + params120 + blob
