"""Host-side mirror of ucoslam::Map::matchFrameToMapPoints on top of the C ABI (uh_projmatch_*).

Reference: src/map.cpp:651-770 — `matchFrameToMapPoints(used_frames, curframe, pose_f2g, minDescDist, maxRepjDist,
markMapPointsAsVisible, useAllPoints, excludedPoints)` returns vector<cv::DMatch> [queryIdx = keypoint, trainIdx = map point id].
The selection of candidate map points (getMapPointsInFrames, lastFIdxSeen filter, :657-668) is map bookkeeping and stays with the
caller; here the frame and the candidate points come flattened (see include/ucoslam_hip.h, uh_proj_frame / uh_map_points).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import I, VP, check, lib, np_ptr
from .orb import KEYPOINT_DTYPE

DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class _ProjFrame(C.Structure):
    _fields_ = [("und_kpts", VP), ("n_kpts", C.c_int32), ("desc", VP), ("scale_factors", VP), ("n_levels", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("min_x", C.c_int32), ("min_y", C.c_int32), ("max_x", C.c_int32), ("max_y", C.c_int32)]


class _MapPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("ids", VP), ("pos3d", VP), ("normal", VP), ("min_dist", VP), ("max_dist", VP), ("desc", VP)]


class _PrevPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("ids", VP), ("pos3d", VP), ("octave", VP), ("desc", VP)]


class _TrackArgs(C.Structure):
    _fields_ = [("pose0", VP), ("intr4", VP), ("inv_sigma_levels", VP), ("n_levels", C.c_int32), ("prev", C.POINTER(_PrevPoints)), ("prev_map_row", VP),
                ("map", C.POINTER(_MapPoints)), ("map_weight", VP), ("prev_min_desc_dist", C.c_float), ("prev_max_repj_dist", C.c_float),
                ("map_min_desc_dist", C.c_float), ("map_radius_tracked", C.c_float), ("map_radius_lost", C.c_float), ("min_inliers", C.c_int32)]


class _TrackResult(C.Structure):
    _fields_ = [("matches_prev", VP), ("bad_prev", VP), ("cap_prev", C.c_int32), ("matches_map", VP), ("cap_map", C.c_int32),
                ("matches_all", VP), ("bad_all", VP), ("cap_all", C.c_int32),
                ("n_prev", C.c_int32), ("n_map", C.c_int32), ("n_all", C.c_int32), ("tracked", C.c_int32), ("inliers1", C.c_int32), ("inliers2", C.c_int32),
                ("iters1", C.c_int32 * 4), ("iters2", C.c_int32 * 4), ("pose1", C.c_float * 16), ("pose2", C.c_float * 16)]


def _declare(L, sig):
    sig("uh_track_pose", I, VP, VP, C.POINTER(_TrackArgs), C.POINTER(_TrackResult))
    sig("uh_projmatch_create", I, VP, C.POINTER(VP))
    sig("uh_projmatch_destroy", None, VP)
    sig("uh_projmatch_set_frame", I, VP, C.POINTER(_ProjFrame))
    sig("uh_projmatch_match", I, VP, VP, C.POINTER(_MapPoints), C.c_float, C.c_float, VP, C.c_int32, VP, VP, VP)
    sig("uh_projmatch_match_prev", I, VP, VP, C.POINTER(_PrevPoints), C.c_float, C.c_float, VP, C.c_int32, VP, VP)
    sig("uh_projmatch_debug_tree", I, VP, C.POINTER(C.c_int32), C.POINTER(VP), C.POINTER(VP), VP, C.POINTER(C.c_int32))
    sig("uh_kdtree_build_host", I, VP, C.c_int32, C.POINTER(C.c_int32), VP, VP, VP, C.POINTER(C.c_int32))
    sig("uh_kdtree_build_dev", I, VP, VP, C.c_int32, C.c_int32, C.POINTER(C.c_int32), VP, VP, VP, C.POINTER(C.c_int32))
    sig("uh_kdtree_sort_restated_host", I, VP, C.c_int32, VP)
    sig("uh_projmatch_set_frame_dev", I, VP, VP, C.POINTER(_ProjFrame))


_lib._EXTRA_DECLS.append(_declare)

INT_MAX = 2 ** 31 - 1
KDNODE_DTYPE = np.dtype([("divlow", "<f4"), ("divhigh", "<f4"), ("left", "<i4"), ("right", "<i4"), ("leaf_begin", "<i4"),
                         ("leaf_count", "<i2"), ("col", "<i2")])


def kdtree_build_host(xy):
    """The kd-tree uh_projmatch_set_frame builds for these (x, y) points — host only, no GPU."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    nodes = np.zeros(2 * n + 2, KDNODE_DTYPE)
    leaf = np.zeros(max(n, 1), np.uint32)
    box = np.zeros(4, np.float64)
    nn, depth = C.c_int32(), C.c_int32()
    check(lib().uh_kdtree_build_host(np_ptr(xy) if n else None, n, C.byref(nn), np_ptr(nodes), np_ptr(leaf), np_ptr(box), C.byref(depth)))
    return dict(nodes=nodes[:nn.value], leaf_idx=leaf[:n], root_box=box, depth=depth.value)


def kdtree_build_dev(ctx, xy, threads=0):
    """The same tree from the device builder (csrc/kdbuild.hpp), n <= 4096 — test hook."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    nodes = np.zeros(2 * n + 2, KDNODE_DTYPE)
    leaf = np.zeros(max(n, 1), np.uint32)
    box = np.zeros(4, np.float64)
    nn, depth = C.c_int32(), C.c_int32()
    check(lib().uh_kdtree_build_dev(ctx.handle, np_ptr(xy) if n else None, n, threads, C.byref(nn), np_ptr(nodes), np_ptr(leaf), np_ptr(box), C.byref(depth)))
    return dict(nodes=nodes[:nn.value], leaf_idx=leaf[:n], root_box=box, depth=depth.value)


def kdtree_sort_restated_host(keys):
    """The permutation the device builder's restatement of libstdc++'s std::sort gives these float keys (host code, no GPU)."""
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(max(len(keys), 1), np.uint32)
    check(lib().uh_kdtree_sort_restated_host(np_ptr(keys) if len(keys) else None, len(keys), np_ptr(perm)))
    return perm[: len(keys)]


class ProjectionMatcher:
    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_projmatch_create(ctx.handle, C.byref(self._h)))
        self._keep = None

    def setFrame(self, und_kpts, desc, scale_factors, fx, fy, cx, cy, min_xy=(0, 0), max_xy=(INT_MAX, INT_MAX)):
        """und_kpts: KEYPOINT_DTYPE array (cv::KeyPoint layout); desc: [n,32] uint8."""
        k = np.ascontiguousarray(und_kpts, KEYPOINT_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(len(k), 32) if len(k) else np.zeros((0, 32), np.uint8)
        s = np.ascontiguousarray(scale_factors, np.float32)
        f = _ProjFrame(np_ptr(k) if len(k) else None, len(k), np_ptr(d) if len(k) else None, np_ptr(s), len(s), fx, fy, cx, cy,
                       int(min_xy[0]), int(min_xy[1]), int(max_xy[0]), int(max_xy[1]))
        check(lib().uh_projmatch_set_frame(self._h, C.byref(f)))
        self.n_kpts = len(k)

    def setFrameDev(self, frame, scale_factors, fx, fy, cx, cy, min_xy=(0, 0), max_xy=(INT_MAX, INT_MAX), und_kpts=None):
        """Adopt the frame ORBextractor.extractFrameDev left on the device (no keypoints / descriptors cross the host link).
        und_kpts: the undistorted keypoints (KEYPOINT_DTYPE) — needed when the frame's tree is built by the host core
        (DeviceFrame.setTreeBuilder(True)), ignored otherwise."""
        s = np.ascontiguousarray(scale_factors, np.float32)
        k = np.ascontiguousarray(und_kpts) if und_kpts is not None else None
        f = _ProjFrame(np_ptr(k) if k is not None and len(k) else None, len(k) if k is not None else 0, None, np_ptr(s), len(s), fx, fy, cx, cy,
                       int(min_xy[0]), int(min_xy[1]), int(max_xy[0]), int(max_xy[1]))
        check(lib().uh_projmatch_set_frame_dev(self._h, frame._h, C.byref(f)))
        self._keep = frame

    def matchFrameToMapPoints(self, pose_f2g, ids, pos3d, normal, min_dist, max_dist, mp_desc, minDescDist, maxRepjDist):
        """Returns dict(matches DMATCH_DTYPE[k], best_kp int32[n], best_dist float32[n], visible uint8[n])."""
        pose = np.ascontiguousarray(pose_f2g, np.float32).reshape(16)
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        a = [np.ascontiguousarray(pos3d, np.float32).reshape(n, 3), np.ascontiguousarray(normal, np.float32).reshape(n, 3),
             np.ascontiguousarray(min_dist, np.float32), np.ascontiguousarray(max_dist, np.float32),
             np.ascontiguousarray(mp_desc, np.uint8).reshape(n, 32)]
        mp = _MapPoints(n, np_ptr(ids) if n else None, *[np_ptr(x) if n else None for x in a])
        out = np.zeros(max(n, 1), DMATCH_DTYPE)
        best_kp = np.full(max(n, 1), -1, np.int32)
        best_d = np.zeros(max(n, 1), np.float32)
        vis = np.zeros(max(n, 1), np.uint8)
        rc = lib().uh_projmatch_match(self._h, np_ptr(pose), C.byref(mp), float(minDescDist), float(maxRepjDist), np_ptr(out), len(out),
                                      np_ptr(best_kp), np_ptr(best_d), np_ptr(vis))
        if rc < 0:
            check(rc)
        return dict(matches=out[:rc].copy(), best_kp=best_kp[:n], best_dist=best_d[:n], visible=vis[:n])

    def matchFrameToPrevFrame(self, pose_f2g, ids, pos3d, octave, prev_desc, minDescDist, maxRepjDist):
        """The tracker's projection search against the previous frame (system.cpp:5930-6460; called with
        maxDescDistance*1.5, projDistThr at :6559-6565).  One item per previous-frame keypoint that carries a valid, non-bad map
        point: its id, coordinates, the keypoint's octave and descriptor.  The current frame is the one given to setFrame,
        pose_f2g its (predicted) pose.  Returns dict(matches DMATCH_DTYPE[k], best_kp int32[n], best_dist float32[n])."""
        pose = np.ascontiguousarray(pose_f2g, np.float32).reshape(16)
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        a = [np.ascontiguousarray(pos3d, np.float32).reshape(n, 3), np.ascontiguousarray(octave, np.int32),
             np.ascontiguousarray(prev_desc, np.uint8).reshape(n, 32)]
        pp = _PrevPoints(n, np_ptr(ids) if n else None, *[np_ptr(x) if n else None for x in a])
        out = np.zeros(max(n, 1), DMATCH_DTYPE)
        best_kp = np.full(max(n, 1), -1, np.int32)
        best_d = np.zeros(max(n, 1), np.float32)
        rc = lib().uh_projmatch_match_prev(self._h, np_ptr(pose), C.byref(pp), float(minDescDist), float(maxRepjDist), np_ptr(out), len(out),
                                           np_ptr(best_kp), np_ptr(best_d))
        if rc < 0:
            check(rc)
        return dict(matches=out[:rc].copy(), best_kp=best_kp[:n], best_dist=best_d[:n])

    def trackPose(self, pnp, pose0, intr4, inv_sigma_levels, prev, mp, prev_map_row=None, map_weight=None, prev_min_desc_dist=75.0, prev_max_repj_dist=15.0,
                  map_min_desc_dist=100.0, map_radius_tracked=4.0, map_radius_lost=15.0, min_inliers=30):
        """uh_track_pose: the previous-frame search, solvePnp, the decision, the local-map search, the union and the second solvePnp of one
        frame as ONE call (system.cpp:5930-6954).  prev: dict(ids, pos3d, octave, desc); mp: dict(ids, pos3d, normal, min_dist, max_dist, desc);
        the frame is the device-resident one given to setFrameDev.  Returns dict(matches_prev, bad_prev, matches_map, matches_all, bad_all,
        tracked, inliers1, inliers2, iters1, iters2, pose1, pose2)."""
        pose = np.ascontiguousarray(pose0, np.float32).reshape(16)
        intr = np.ascontiguousarray(intr4, np.float32).reshape(4)
        isl = np.ascontiguousarray(inv_sigma_levels, np.float32)
        pid = np.ascontiguousarray(prev["ids"], np.uint32)
        n_p = len(pid)
        pa = [np.ascontiguousarray(prev["pos3d"], np.float32).reshape(n_p, 3), np.ascontiguousarray(prev["octave"], np.int32), np.ascontiguousarray(prev["desc"], np.uint8).reshape(n_p, 32)]
        pp = _PrevPoints(n_p, np_ptr(pid) if n_p else None, *[np_ptr(x) if n_p else None for x in pa])
        mid = np.ascontiguousarray(mp["ids"], np.uint32)
        n_m = len(mid)
        ma = [np.ascontiguousarray(mp["pos3d"], np.float32).reshape(n_m, 3), np.ascontiguousarray(mp["normal"], np.float32).reshape(n_m, 3),
              np.ascontiguousarray(mp["min_dist"], np.float32), np.ascontiguousarray(mp["max_dist"], np.float32), np.ascontiguousarray(mp["desc"], np.uint8).reshape(n_m, 32)]
        mpp = _MapPoints(n_m, np_ptr(mid) if n_m else None, *[np_ptr(x) if n_m else None for x in ma])
        row = np.ascontiguousarray(prev_map_row, np.int32) if prev_map_row is not None else None
        wgt = np.ascontiguousarray(map_weight, np.float32) if map_weight is not None else None
        args = _TrackArgs(np_ptr(pose), np_ptr(intr), np_ptr(isl), len(isl), C.pointer(pp), np_ptr(row) if row is not None and n_p else None, C.pointer(mpp),
                          np_ptr(wgt) if wgt is not None and n_m else None, float(prev_min_desc_dist), float(prev_max_repj_dist), float(map_min_desc_dist),
                          float(map_radius_tracked), float(map_radius_lost), int(min_inliers))
        m1 = np.zeros(max(n_p, 1), DMATCH_DTYPE); b1 = np.zeros(max(n_p, 1), np.uint8)
        m2 = np.zeros(max(n_m, 1), DMATCH_DTYPE)
        mA = np.zeros(max(n_p + n_m, 1), DMATCH_DTYPE); bA = np.zeros(max(n_p + n_m, 1), np.uint8)
        res = _TrackResult(np_ptr(m1), np_ptr(b1), len(m1), np_ptr(m2), len(m2), np_ptr(mA), np_ptr(bA), len(mA))
        check(lib().uh_track_pose(self._h, pnp._h, C.byref(args), C.byref(res)))
        return dict(matches_prev=m1[: res.n_prev].copy(), bad_prev=b1[: res.n_prev].copy(), matches_map=m2[: res.n_map].copy(), matches_all=mA[: res.n_all].copy(),
                    bad_all=bA[: res.n_all].copy(), tracked=bool(res.tracked), inliers1=res.inliers1, inliers2=res.inliers2, iters1=np.array(res.iters1[:], np.int32),
                    iters2=np.array(res.iters2[:], np.int32), pose1=np.array(res.pose1[:], np.float32), pose2=np.array(res.pose2[:], np.float32))

    def debug_tree(self):
        nn, nodes, leaf, depth = C.c_int32(), VP(), VP(), C.c_int32()
        box = np.zeros(4, np.float64)
        check(lib().uh_projmatch_debug_tree(self._h, C.byref(nn), C.byref(nodes), C.byref(leaf), np_ptr(box), C.byref(depth)))
        n = nn.value
        nd = np.frombuffer((C.c_char * (24 * n)).from_address(nodes.value), KDNODE_DTYPE).copy() if n else np.zeros(0, KDNODE_DTYPE)
        li = np.frombuffer((C.c_char * (4 * self.n_kpts)).from_address(leaf.value), np.uint32).copy() if self.n_kpts else np.zeros(0, np.uint32)
        return dict(nodes=nd, leaf_idx=li, root_box=box, depth=depth.value)

    def close(self):
        if self._h:
            lib().uh_projmatch_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
