"""Host-side mirror of the `.slm` checkpoint sections the hot path owns (uh_system_stream_* / uh_params_from_stream).

Reference: System::saveToFile / readFromFile (src/utils/system.cpp:8099-8720): u64 182312 | Map | Params | se3 pose | i64 current keyframe |
bool initialised | STATE | MODES | Frame current | Frame previous | FrameExtractor | MapManager | cv::Mat | i64 | u64.  Map / Frame /
MapManager are the host's containers (opaque bytes here); Params (src/ucoslamtypes.cpp:63-180) and the FrameExtractor block
(ucoslam_cv3_amd.orb.ORBextractor.frameExtractorToStream) belong to the path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import I, VP, check, lib

U64 = C.c_uint64


class ParamsView(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in ("detect_markers", "detect_keypoints", "kp_non_maxima_suppression", "force_initialization_from_markers",
                                         "remove_keypoints_into_markers", "run_sequential", "auto_adjust_kp_sensitivity", "relocalization_with_keypoints",
                                         "relocalization_with_markers")] + [("kp_descriptor_type", C.c_int8)] + \
               [(n, C.c_float) for n in ("target_focus", "kf_min_confidence", "max_desc_distance", "baseline_median_depth_ratio_min", "aruco_marker_size",
                                         "kf_culling", "th_ref_ratio", "scale_factor", "min_base_line", "kpt_image_scale_factor")] + \
               [(n, C.c_int32) for n in ("max_new_points", "proj_dist_thr", "nthreads_feature_detector", "max_visible_frames_per_marker", "min_num_proj_points",
                                         "max_features", "n_octave_levels")] + [("global_optimizer", C.c_char * 32)]


class SystemState(C.Structure):
    _fields_ = [("cur_pose_rt", C.c_float * 6), ("current_keyframe", C.c_int64), ("is_initialized", C.c_uint8), ("state", C.c_int32), ("mode", C.c_int32)]


class SystemTail(C.Structure):
    _fields_ = [("mat_rows", C.c_int32), ("mat_cols", C.c_int32), ("mat_type", C.c_int32), ("mat_data_offset", U64), ("mat_data_bytes", U64),
                ("last_value_i64", C.c_int64), ("last_value_u64", U64)]


class SystemParts(C.Structure):
    _fields_ = [("map", VP), ("map_bytes", U64), ("params", VP), ("params_bytes", U64), ("state", SystemState), ("cur_frame", VP), ("cur_frame_bytes", U64),
                ("prev_frame", VP), ("prev_frame_bytes", U64), ("extractor", VP), ("extractor_bytes", U64), ("map_manager", VP), ("map_manager_bytes", U64),
                ("tail", SystemTail), ("mat_data", VP)]


def _declare(L, sig):
    sig("uh_params_from_stream", I, VP, U64, C.POINTER(ParamsView), C.POINTER(U64))
    sig("uh_system_stream_begin", I, VP, U64, C.POINTER(U64))
    sig("uh_system_stream_state", I, VP, U64, C.POINTER(ParamsView), C.POINTER(SystemState), C.POINTER(U64), C.POINTER(U64))
    sig("uh_system_stream_tail", I, VP, U64, C.POINTER(SystemTail), C.POINTER(U64))
    sig("uh_system_to_stream", I, C.POINTER(SystemParts), VP, U64, C.POINTER(U64))


_lib._EXTRA_DECLS.append(_declare)


def _buf(b):
    return (C.c_char * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


def params_from_stream(data: bytes):
    """Params::fromStream: returns (ParamsView, bytes consumed)."""
    pv, used = ParamsView(), U64()
    check(lib().uh_params_from_stream(_buf(data), len(data), C.byref(pv), C.byref(used)))
    return pv, used.value


def system_stream_begin(data: bytes) -> int:
    off = U64()
    check(lib().uh_system_stream_begin(_buf(data), len(data), C.byref(off)))
    return off.value


def system_stream_state(data: bytes):
    """On the bytes behind the host's Map block: (ParamsView, SystemState, params_bytes, consumed)."""
    pv, st, pb, used = ParamsView(), SystemState(), U64(), U64()
    check(lib().uh_system_stream_state(_buf(data), len(data), C.byref(pv), C.byref(st), C.byref(pb), C.byref(used)))
    return pv, st, pb.value, used.value


def system_stream_tail(data: bytes):
    """On the bytes behind the host's MapManager block: (SystemTail, consumed)."""
    t, used = SystemTail(), U64()
    check(lib().uh_system_stream_tail(_buf(data), len(data), C.byref(t), C.byref(used)))
    return t, used.value


def system_to_stream(map_bytes, state: SystemState, cur_frame, prev_frame, extractor, map_manager, tail: SystemTail, mat_data=b"", params=None) -> bytes:
    """System::saveToFile's byte stream from the host's blocks (opaque), the path's blocks and the state members."""
    keep = [_buf(b) if b is not None else None for b in (map_bytes, params, cur_frame, prev_frame, extractor, map_manager, mat_data)]
    ptr = lambda k, b: (C.cast(k, VP) if b is not None and len(b) else None)
    p = SystemParts(ptr(keep[0], map_bytes), len(map_bytes or b""), ptr(keep[1], params), len(params or b""), state, ptr(keep[2], cur_frame), len(cur_frame or b""),
                    ptr(keep[3], prev_frame), len(prev_frame or b""), ptr(keep[4], extractor), len(extractor or b""), ptr(keep[5], map_manager), len(map_manager or b""),
                    tail, ptr(keep[6], mat_data))
    size = U64()
    check(lib().uh_system_to_stream(C.byref(p), None, 0, C.byref(size)))
    out = np.zeros(size.value, np.uint8)
    check(lib().uh_system_to_stream(C.byref(p), out.ctypes.data_as(VP), size.value, C.byref(size)))
    return out.tobytes()
