"""Host-side mirror of ucoslam::FrameMatcher (TYPE_FLANN) on top of the C ABI.

Reference: src/utils/framematcher.h:33-60, framematcher.cpp:140-319.  `setParams(trainFrame, mode, minDescDist,
nn_match_ratio, checkOrientation, maxOctaveDiff)` builds the index over the selected train descriptors, `match(queryFrame,
mode)` / `matchEpipolar(queryFrame, mode, F12)` search nn=10 neighbours (unsorted rows) and run the reference's filter chain.
The index is the reference's: xflann HKMeans(32,0) searched with maxChecks=16 (approximate, framematcher.cpp:121-122,213,239),
reproduced bit for bit (uh_knn_build_kmeans / uh_knn_search_kmeans).  `exact=True` swaps in the brute-force scan, i.e. the
candidate set the approximate search tries to recover (a superset in quality, not the reference's rows).

A frame is a dict with: desc [N,32] uint8, ids [N] uint32 (0xFFFFFFFF = unassigned), nonmaxima [N] bool (FLAG_NONMAXIMA),
octave [N] int32, angle [N] float32, pt [N,2] float32 (und_kpts), scaleFactors [levels] float32.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import I, VP, check, lib, np_ptr
from .knn import Index

MODE_ALL, MODE_ASSIGNED, MODE_UNASSIGNED = 0, 1, 2
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class _Args(C.Structure):
    _fields_ = [("nq", C.c_int32), ("nn", C.c_int32), ("indices", VP), ("distances", VP), ("map_idx_query", VP), ("map_idx_train", VP),
                ("q_octave", VP), ("q_angle", VP), ("q_pt", VP), ("t_octave", VP), ("t_angle", VP), ("t_pt", VP),
                ("scale_factors", VP), ("F12", VP), ("min_desc_dist", C.c_float), ("nn_match_ratio", C.c_float),
                ("check_orientation", C.c_int32), ("max_octave_diff", C.c_int32),
                ("n_query_kpts", C.c_int32), ("n_train_kpts", C.c_int32), ("n_levels", C.c_int32)]


class _BowFrame(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_ids", VP), ("node_ptr", VP), ("feat_idx", VP), ("n_kpts", C.c_int32), ("desc", VP),
                ("octave", VP), ("angle", VP), ("pt", VP), ("used", VP)]


class _BowArgs(C.Structure):
    _fields_ = [("query", _BowFrame), ("train", _BowFrame), ("scale_factors", VP), ("n_levels", C.c_int32), ("F12", VP),
                ("min_desc_dist", C.c_float), ("nn_match_ratio", C.c_float), ("check_orientation", C.c_int32), ("max_octave_diff", C.c_int32)]


def _declare(L, sig):
    sig("uh_match_filter", I, C.POINTER(_Args), VP, I)
    sig("uh_filter_ambiguous", I, VP, I, I)
    sig("uh_bowmatch_create", I, VP, C.POINTER(VP))
    sig("uh_bowmatch_destroy", None, VP)
    sig("uh_bowmatch_match", I, VP, C.POINTER(_BowArgs), VP, I)


_lib._EXTRA_DECLS.append(_declare)


def manage_mode(mode, frame):
    """FrameMatcher_Flann::manageMode (framematcher.cpp:166-199): rows used and their keypoint indices."""
    n = len(frame["ids"])
    if mode == MODE_ALL:
        sel = np.arange(n)
    else:
        assigned = frame["ids"] != 0xFFFFFFFF
        keep = (assigned if mode == MODE_ASSIGNED else ~assigned) & ~frame["nonmaxima"].astype(bool)
        sel = np.nonzero(keep)[0]
    return sel.astype(np.uint32), np.ascontiguousarray(frame["desc"][sel])


def match_filter(indices, distances, query, train, map_q=None, map_t=None, min_desc_dist=np.inf, nn_match_ratio=0.8,
                 check_orientation=True, max_octave_diff=1, F12=None):
    """The filter chain alone (uh_match_filter); pure host code, runs without a GPU."""
    indices = np.ascontiguousarray(indices, np.int32)
    distances = np.ascontiguousarray(distances, np.int32)
    nq, nn = indices.shape
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return np_ptr(a)

    args = _Args(nq, nn, np_ptr(indices), np_ptr(distances), arr(map_q, np.uint32) if map_q is not None else None,
                 arr(map_t, np.uint32) if map_t is not None else None, arr(query["octave"], np.int32), arr(query["angle"], np.float32),
                 arr(query["pt"], np.float32), arr(train["octave"], np.int32), arr(train["angle"], np.float32), arr(train["pt"], np.float32),
                 arr(query["scaleFactors"], np.float32), arr(F12, np.float32) if F12 is not None else None,
                 float(min(min_desc_dist, np.finfo(np.float32).max)), float(nn_match_ratio), int(check_orientation), int(max_octave_diff),
                 len(query["octave"]), len(train["octave"]), len(query["scaleFactors"]))   # sizes: every mapped index is range-checked
    out = np.zeros(max(nq, 1), DMATCH_DTYPE)
    n = lib().uh_match_filter(C.byref(args), np_ptr(out), len(out))
    if n < 0:
        check(n)
    return out[:n].copy()


class FrameMatcher:
    NN = 10           # framematcher.cpp:122: nn=10 nearest neighbours per query
    MAX_SEARCH = 16   # framematcher.cpp:121: maxChecks of the approximate search

    def __init__(self, ctx: _lib.Context, exact: bool = False):
        self.ctx = ctx
        self.index = Index(ctx)
        self.exact = exact
        self._train = None
        self._built = 0

    def setParams(self, trainFrame, mode=MODE_ALL, minDescDist=np.inf, nn_match_ratio=0.8, checkOrientation=True, maxOctaveDiff=1):
        self._p = dict(min_desc_dist=minDescDist, nn_match_ratio=nn_match_ratio, check_orientation=checkOrientation,
                       max_octave_diff=maxOctaveDiff)
        self._map_t, desc = manage_mode(mode, trainFrame)
        if self.exact:
            self.index.build(desc)
        else:
            self.index.build_kmeans(desc, 32, 0)    # trainIndex.build(desc, xflann::HKMeansParams(32,0))
        self._built = len(desc)
        self._train = trainFrame

    def match(self, queryFrame, mode=MODE_ALL):
        return self.matchEpipolar(queryFrame, mode, None)

    def matchEpipolar(self, queryFrame, mode=MODE_ALL, F12=None):
        map_q, qdesc = manage_mode(mode, queryFrame)
        if self._built == 0 or len(qdesc) == 0:
            return np.zeros(0, DMATCH_DTYPE)        # search() returns false on an unbuilt index -> {} (framematcher.cpp:239-240)
        if self.exact:
            idx, dist = self.index.search(qdesc, self.NN, sorted=False)
        else:
            idx, dist = self.index.search_kmeans(qdesc, self.NN, self.MAX_SEARCH, sorted=False)
        return match_filter(idx, dist, queryFrame, self._train, map_q, self._map_t, F12=F12, **self._p)


def is_used(frame, mode):
    """FrameMatcher_BoW::isUsed (framematcher.cpp:373-379) for every keypoint of a frame."""
    used = ~frame["nonmaxima"].astype(bool)
    if mode == MODE_ASSIGNED:
        used &= frame["ids"] != 0xFFFFFFFF
    elif mode == MODE_UNASSIGNED:
        used &= frame["ids"] == 0xFFFFFFFF
    return used.astype(np.uint8)


class FrameMatcherBoW:
    """ucoslam::FrameMatcher (TYPE_BOW): framematcher.cpp:395-535.  A frame is the dict FrameMatcher takes plus
    `bowvector_level`: {node id: [keypoint indices]} — the fBow2 of Vocabulary.transform(desc, 3) (keyframedatabase.cpp:319)."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_bowmatch_create(ctx.handle, C.byref(self._h)))
        self._train = None

    def setParams(self, trainFrame, mode=MODE_ALL, minDescDist=np.inf, nn_match_ratio=0.8, checkOrientation=True, maxOctaveDiff=1):
        self._p = (float(min(minDescDist, np.finfo(np.float32).max)), float(nn_match_ratio), int(checkOrientation), int(maxOctaveDiff))
        self._train, self._train_mode = trainFrame, mode

    def match(self, queryFrame, mode=MODE_ALL):
        return self.matchEpipolar(queryFrame, mode, None)

    def matchEpipolar(self, queryFrame, mode=MODE_ALL, F12=None):
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return np_ptr(a) if a.size else None

        def frame(f, m):
            bv = f["bowvector_level"]
            ids = sorted(bv)                                    # std::map order
            ptr = np.zeros(len(ids) + 1, np.int32)
            for i, k in enumerate(ids):
                ptr[i + 1] = ptr[i] + len(bv[k])
            feat = np.concatenate([np.asarray(bv[k], np.uint32) for k in ids]) if ids else np.zeros(0, np.uint32)
            n = len(f["desc"])
            return _BowFrame(len(ids), arr(ids, np.uint32), arr(ptr, np.int32), arr(feat, np.uint32), n, arr(f["desc"], np.uint8),
                             arr(f["octave"], np.int32), arr(f["angle"], np.float32), arr(f["pt"], np.float32), arr(is_used(f, m), np.uint8))

        md, ratio, co, mod = self._p
        sf = np.ascontiguousarray(queryFrame["scaleFactors"], np.float32)
        args = _BowArgs(frame(queryFrame, mode), frame(self._train, self._train_mode), np_ptr(sf), len(sf),
                        arr(F12, np.float32) if F12 is not None else None, md, ratio, co, mod)
        out = np.zeros(max(len(queryFrame["desc"]), 1), DMATCH_DTYPE)
        n = lib().uh_bowmatch_match(self._h, C.byref(args), np_ptr(out), len(out))
        if n < 0:
            check(n)
        return out[:n].copy()

    def close(self):
        if self._h:
            lib().uh_bowmatch_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
