"""Host-side mirror of the reference extractor plugin on top of the C ABI.

Reference surface: `Feature2DSerializable` (src/featureextractors/feature2dserializable.h:30-95):
`create(DescriptorTypes::DESC_ORB)`, `detectAndCompute(image, mask, keypoints, descriptors, FeatParams)`,
`getParams`, `getMinDescDistance` (ORB: 50, ORBextractor.h:105), `setSensitivity`, `toStream/fromStream`.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib
from ._lib import I, SZ, VP, check, dev_ptr, lib, np_ptr

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28


class FeatParams(C.Structure):
    """Feature2DSerializable::FeatParams (feature2dserializable.h:34-60), same defaults."""
    _fields_ = [("nthreads", C.c_int32), ("maxFeatures", C.c_int32), ("nOctaveLevels", C.c_int32),
                ("scaleFactor", C.c_float), ("sensitivity", C.c_float)]

    def __init__(self, maxFeatures=4000, nOctaveLevels=8, scaleFactor=1.2, nthreads=-1, sensitivity=0.0):
        super().__init__(nthreads, maxFeatures, nOctaveLevels, scaleFactor, sensitivity)


class Camera(C.Structure):
    """uh_camera: ImageParams' CameraMatrix (CV_32F) and Distorsion (k1 k2 p1 p2 [k3 [k4 k5 k6]])."""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("dist", C.c_float * 8), ("n_dist", C.c_int32)]

    def __init__(self, fx, fy, cx, cy, dist=()):
        d = (C.c_float * 8)(*[float(v) for v in dist])
        super().__init__(fx, fy, cx, cy, d, len(dist))


def undistort_points_host(cam: Camera, xy):
    """undistortPoints(points, ImageParams) (misc.cpp:269-293) on the host: n x 2 float32 in, n x 2 out."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    check(lib().uh_undistort_points_host(C.byref(cam), np_ptr(xy) if len(xy) else None, len(xy), np_ptr(out) if len(xy) else None))
    return out


class FrameExtractorState(C.Structure):
    """uh_frame_extractor_state: what FrameExtractor::toStream writes between the extractor's own stream and the two sub-streams."""
    _fields_ = [("counter", C.c_uint32), ("remove_from_markers", C.c_uint8), ("detect_markers", C.c_uint8), ("detect_keypoints", C.c_uint8),
                ("marker_size", C.c_float), ("feat_params", FeatParams), ("max_desc_distance", C.c_float)]


def _declare(L, sig):
    sig("uh_frame_extractor_to_stream", I, VP, C.c_char_p, C.POINTER(FrameExtractorState), VP, C.c_uint64, VP, C.c_uint64, VP, C.c_uint64, C.POINTER(C.c_uint64))
    sig("uh_frame_extractor_from_stream", I, VP, VP, C.c_uint64, I, C.POINTER(FrameExtractorState), VP, C.c_uint64, C.POINTER(C.c_uint64),
        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("uh_orb_to_stream", I, VP, C.c_char_p, VP, C.c_uint64, C.POINTER(C.c_uint64))
    sig("uh_orb_from_stream", I, VP, VP, C.c_uint64, VP, C.c_uint64, C.POINTER(C.c_uint64))
    sig("uh_orb_create", I, VP, C.POINTER(VP))
    sig("uh_orb_destroy", None, VP)
    sig("uh_orb_set_params", I, VP, C.POINTER(FeatParams))
    sig("uh_orb_get_params", I, VP, C.POINTER(FeatParams))
    sig("uh_orb_set_blur", I, VP, I)
    sig("uh_orb_set_sensitivity", I, VP, C.c_float)
    sig("uh_orb_set_nonmaxima", I, VP, I)
    sig("uh_orb_set_level_range", I, VP, I, I)
    sig("uh_orb_max_keypoints", I, VP)
    sig("uh_orb_extract", I, VP, VP, I, I, SZ, VP, VP, I, C.POINTER(I))
    sig("uh_orb_extract_dev", I, VP, VP, I, I, SZ, SZ, I, VP, VP, I, VP)
    sig("uh_orb_debug_level", I, VP, I, I, I, VP, C.POINTER(I), C.POINTER(I))
    sig("uh_orb_set_camera", I, VP, C.POINTER(Camera))
    sig("uh_orb_extract_frame", I, VP, VP, I, I, SZ, I, VP, VP, VP, I, C.POINTER(I))
    sig("uh_undistort_points_host", I, C.POINTER(Camera), VP, I, VP)
    sig("uh_dev_frame_create", I, VP, C.POINTER(VP))
    sig("uh_dev_frame_destroy", None, VP)
    sig("uh_dev_frame_set_tree_builder", I, VP, C.c_int32)
    sig("uh_dev_frame_upload", I, VP, VP, C.c_int32, VP)
    sig("uh_orb_extract_frame_dev", I, VP, VP, I, I, SZ, I, VP, VP, VP, I, C.POINTER(I), VP)
    sig("uh_orb_extract_frame_dev_begin", I, VP, VP, I, I, SZ, I, VP, VP, VP, I, C.POINTER(I), VP, C.POINTER(VP))
    sig("uh_orb_extract_frame_dev_end", I, VP, C.POINTER(I))
    sig("uh_dev_frame_tree", I, VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), VP, VP, VP, VP, VP, C.POINTER(C.c_int32))


_lib._EXTRA_DECLS.append(_declare)


class DeviceFrame:
    """uh_dev_frame: the Frame FrameExtractor::process produces (descriptors, undistorted keypoints, kd-tree), resident in HBM."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_dev_frame_create(ctx.handle, C.byref(self._h)))

    def setTreeBuilder(self, on_host: bool):
        """Who builds the kd-tree: the build launches behind the extraction (False, default) or the host core inside
        ProjectionMatcher.setFrameDev(..., und_kpts=...) (True: no build launch; the descriptors stay on the device either way)."""
        check(lib().uh_dev_frame_set_tree_builder(self._h, 1 if on_host else 0))
        return self

    def upload(self, und_kpts, desc):
        """uh_dev_frame_upload: a frame from elsewhere (KEYPOINT_DTYPE array with undistorted x / y, n x 32 descriptors) into this object."""
        k = np.ascontiguousarray(und_kpts, KEYPOINT_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(len(k), 32)
        check(lib().uh_dev_frame_upload(self._h, np_ptr(k) if len(k) else None, len(k), np_ptr(d) if len(k) else None))
        return self

    def tree(self):
        """The kd-tree of the latest extraction, copied out (waits for the build launch): dict(nodes, leaf_idx, leaf_xy, leaf_octave, root_box, depth)."""
        from .projmatch import KDNODE_DTYPE

        n, nn, depth = C.c_int32(), C.c_int32(), C.c_int32()
        box = np.zeros(4, np.float64)
        check(lib().uh_dev_frame_tree(self._h, C.byref(n), C.byref(nn), None, None, None, None, np_ptr(box), C.byref(depth)))
        nodes = np.zeros(max(nn.value, 1), KDNODE_DTYPE)
        leaf = np.zeros(max(n.value, 1), np.uint32)
        xy = np.zeros((max(n.value, 1), 2), np.float32)
        oc = np.zeros(max(n.value, 1), np.int32)
        check(lib().uh_dev_frame_tree(self._h, C.byref(n), C.byref(nn), np_ptr(nodes), np_ptr(leaf), np_ptr(xy), np_ptr(oc), np_ptr(box), C.byref(depth)))
        return dict(nodes=nodes[: nn.value], leaf_idx=leaf[: n.value], leaf_xy=xy[: n.value], leaf_octave=oc[: n.value], root_box=box, depth=depth.value)

    def close(self):
        if self._h:
            lib().uh_dev_frame_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ORBextractor:
    """GPU ORB extractor with the reference's Feature2DSerializable surface."""

    F2D_ORB = 0                       # Feature2DSerializable::F2S_Type
    STREAM_SIG = 1828374733           # feature2dserializable.cpp:78

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_orb_create(ctx.handle, C.byref(self._h)))

    @staticmethod
    def create(ctx: _lib.Context, desc_type: str = "orb") -> "ORBextractor":
        if desc_type != "orb":
            raise RuntimeError("Invalid input descriptor")          # feature2dserializable.cpp:71
        return ORBextractor(ctx)

    def getMinDescDistance(self) -> float:
        return 50.0

    def getParams(self) -> FeatParams:
        fp = FeatParams()
        check(lib().uh_orb_get_params(self._h, C.byref(fp)))
        return fp

    def setSensitivity(self, v: float):
        check(lib().uh_orb_set_sensitivity(self._h, float(v)))

    def doGaussianBlur(self, flag: bool):
        check(lib().uh_orb_set_blur(self._h, int(flag)))

    def setNonMaxima(self, flag: bool):
        """The reference's debug switch: debug::Debug::addString("orb_nonmaxima") (ORBextractor.cpp:1146-1148)."""
        check(lib().uh_orb_set_nonmaxima(self._h, int(flag)))

    def setLevelRange(self, first: int = 0, end: int = -1):
        """Pyramid-level shard [first, end) of a multi-GPU extraction (parallel.sharded_extract); (0, -1) = all levels."""
        check(lib().uh_orb_set_level_range(self._h, first, end))

    # detectAndCompute(image, mask, keypoints, descriptors, params): mask is ignored (ORBextractor.cpp:1253)
    def detectAndCompute(self, image, mask=None, params: FeatParams | None = None):
        if params is not None:
            check(lib().uh_orb_set_params(self._h, C.byref(params)))
        img = np.asarray(image)
        if img.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise _lib.UcoslamHipError(_lib.UH_EINVAL, "image must be CV_8UC1 (2-D uint8)")   # assert at :1268
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        cap = max(lib().uh_orb_max_keypoints(self._h), 1)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(lib().uh_orb_extract(self._h, np_ptr(img), img.shape[1], img.shape[0], img.strides[0], np_ptr(kps), np_ptr(desc),
                                   cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def setCamera(self, cam: Camera | None):
        check(lib().uh_orb_set_camera(self._h, C.byref(cam) if cam is not None else None))
        return self

    def extractFrameDev(self, image, frame: "DeviceFrame", params: FeatParams | None = None):
        """extractFrame that also leaves the frame (descriptors, undistorted keypoints, kd-tree) in `frame` on the device
        (uh_orb_extract_frame_dev); same host outputs."""
        return self.extractFrame(image, params, True, frame)

    def extractFrameDevBegin(self, image, frame: "DeviceFrame", params: FeatParams | None = None):
        """First half of extractFrameDev (uh_orb_extract_frame_dev_begin): returns the undistorted keypoints — x, y, octave only — as soon as the
        selection is done; build the tree with them (ProjectionMatcher.setFrameDev(..., und_kpts=...)), then call extractFrameDevEnd()."""
        if params is not None:
            check(lib().uh_orb_set_params(self._h, C.byref(params)))
        img = np.ascontiguousarray(image)
        cn = 1 if img.ndim == 2 else img.shape[2]
        cap = max(lib().uh_orb_max_keypoints(self._h), 1)
        self._pend = (np.zeros(cap, KEYPOINT_DTYPE), np.zeros((cap, 32), np.uint8), np.zeros((cap, 2), np.float32), img)
        n, early = C.c_int(0), VP()
        check(lib().uh_orb_extract_frame_dev_begin(self._h, np_ptr(img), img.shape[1], img.shape[0], img.strides[0], cn, np_ptr(self._pend[0]), np_ptr(self._pend[1]),
                                                   np_ptr(self._pend[2]), cap, C.byref(n), frame._h, C.byref(early)))
        if not n.value:
            return np.zeros(0, KEYPOINT_DTYPE)
        return np.frombuffer((C.c_char * (n.value * KEYPOINT_DTYPE.itemsize)).from_address(early.value), KEYPOINT_DTYPE).copy()

    def extractFrameDevEnd(self):
        """Second half: (keypoints, descriptors, und_xy) of the extraction extractFrameDevBegin started."""
        n = C.c_int(0)
        check(lib().uh_orb_extract_frame_dev_end(self._h, C.byref(n)))
        kps, desc, und, _ = self._pend
        self._pend = None
        return kps[: n.value].copy(), desc[: n.value].copy(), und[: n.value].copy()

    def extractFrame(self, image, params: FeatParams | None = None, undistorted=True, _frame=None):
        """The frame as the camera delivers it (H x W gray, H x W x 3 BGR or H x W x 4 BGRA, uint8) -> (keypoints, descriptors, und_xy):
        FrameExtractor's cvtColor + detectAndCompute + undistortPoints (frameextractor.cpp:2960, :3985) in one call."""
        if params is not None:
            check(lib().uh_orb_set_params(self._h, C.byref(params)))
        img = np.asarray(image)
        cn = 1 if img.ndim == 2 else img.shape[2]
        if img.dtype != np.uint8 or img.ndim not in (2, 3) or cn not in (1, 3, 4):
            raise _lib.UcoslamHipError(_lib.UH_EINVAL, "image must be uint8, H x W [x 3 | x 4]")
        if img.strides[-1] != 1 or (img.ndim == 3 and img.strides[1] != cn):
            img = np.ascontiguousarray(img)
        cap = max(lib().uh_orb_max_keypoints(self._h), 1)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        und = np.zeros((cap, 2), np.float32) if undistorted else None
        n = C.c_int(0)
        if _frame is not None:
            check(lib().uh_orb_extract_frame_dev(self._h, np_ptr(img), img.shape[1], img.shape[0], img.strides[0], cn, np_ptr(kps), np_ptr(desc),
                                                 np_ptr(und) if undistorted else None, cap, C.byref(n), _frame._h))
        else:
            check(lib().uh_orb_extract_frame(self._h, np_ptr(img), img.shape[1], img.shape[0], img.strides[0], cn, np_ptr(kps), np_ptr(desc),
                                             np_ptr(und) if undistorted else None, cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy(), (und[: n.value].copy() if undistorted else None)

    def extract_batch(self, frames, params: FeatParams | None = None, out=None):
        """frames: torch uint8 CUDA tensor [B,H,W] (resident in HBM). Returns (kps [B,cap,7] f32 view, desc [B,cap,32], counts [B])."""
        import torch

        if params is not None:
            check(lib().uh_orb_set_params(self._h, C.byref(params)))
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3
        frames = frames.contiguous()
        B, H, W = frames.shape
        cap = max(lib().uh_orb_max_keypoints(self._h), 1)
        if out is None:
            kps = torch.empty((B, cap, 7), dtype=torch.float32, device=frames.device)
            desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=frames.device)
            counts = torch.empty((B,), dtype=torch.int32, device=frames.device)
        else:
            kps, desc, counts = out
        check(lib().uh_orb_extract_dev(self._h, dev_ptr(frames), W, H, W, W * H, B, dev_ptr(kps), dev_ptr(desc), cap, dev_ptr(counts)))
        return kps, desc, counts

    def debug_level(self, frame: int, level: int, which: int = 0):
        w, h = C.c_int(0), C.c_int(0)
        check(lib().uh_orb_debug_level(self._h, frame, level, which, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        check(lib().uh_orb_debug_level(self._h, frame, level, which, np_ptr(out), C.byref(w), C.byref(h)))
        return out

    # Feature2DSerializable::toStream / fromStream (feature2dserializable.cpp:76-113) + ORBextractor::toStream_impl / fromStream_impl
    def toStream(self, str_params: str = "") -> bytes:
        size = C.c_uint64()
        sp = str_params.encode()
        check(lib().uh_orb_to_stream(self._h, sp, None, 0, C.byref(size)))
        out = np.zeros(size.value, np.uint8)
        check(lib().uh_orb_to_stream(self._h, sp, np_ptr(out), size.value, C.byref(size)))
        return out.tobytes()

    @staticmethod
    def fromStream(ctx: _lib.Context, data: bytes):
        """-> (extractor, str_params, bytes consumed); raises like the reference on a wrong signature."""
        ext = ORBextractor(ctx)
        buf = np.frombuffer(data, np.uint8)
        sp = C.create_string_buffer(4096)
        used = C.c_uint64()
        check(lib().uh_orb_from_stream(ext._h, np_ptr(buf), len(buf), sp, len(sp), C.byref(used)))
        return ext, sp.value.decode(), used.value

    # FrameExtractor::toStream / fromStream (frameextractor.cpp:651-1136): the extractor block of a reference .slm checkpoint
    def frameExtractorToStream(self, state: "FrameExtractorState", str_params: str = "", aruco: bytes | None = None, params: bytes | None = None) -> bytes:
        size = C.c_uint64()
        a = np.frombuffer(aruco, np.uint8) if aruco else None
        p = np.frombuffer(params, np.uint8) if params else None
        args = (self._h, str_params.encode(), C.byref(state), np_ptr(a) if a is not None else None, len(aruco or b""), np_ptr(p) if p is not None else None, len(params or b""))
        check(lib().uh_frame_extractor_to_stream(*args, None, 0, C.byref(size)))
        out = np.zeros(size.value, np.uint8)
        check(lib().uh_frame_extractor_to_stream(*args, np_ptr(out), size.value, C.byref(size)))
        return out.tobytes()

    @staticmethod
    def frameExtractorFromStream(ctx: _lib.Context, data: bytes, allow_markers: bool = False):
        """-> (extractor, FrameExtractorState, str_params, aruco sub-stream bytes, Params sub-stream bytes, bytes consumed)."""
        ext = ORBextractor(ctx)
        buf = np.frombuffer(data, np.uint8)
        st = FrameExtractorState()
        sp = C.create_string_buffer(4096)
        ao, ab, po, pb, used = (C.c_uint64() for _ in range(5))
        check(lib().uh_frame_extractor_from_stream(ext._h, np_ptr(buf), len(buf), int(allow_markers), C.byref(st), sp, len(sp), C.byref(ao), C.byref(ab),
                                                   C.byref(po), C.byref(pb), C.byref(used)))
        return ext, st, sp.value.decode(), data[ao.value:ao.value + ab.value], data[po.value:po.value + pb.value], used.value

    def close(self):
        if self._h:
            lib().uh_orb_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
