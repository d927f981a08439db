"""FrameExtractor's two per-frame steps around the extractor — cv::cvtColor(COLOR_BGR2GRAY) of a three-channel frame
(src/utils/frameextractor.cpp:2960,3046) and undistortPoints(keypoints, ImageParams) (src/basictypes/misc.cpp:269-293, called at
frameextractor.cpp:3985) — through uh_orb_extract_frame / uh_undistort_points_host against oracle/orb_oracle.cpp and known answers.
PARITY UNPINNED (both steps call into OpenCV, which this image does not have): the KATs below are what can be known without it."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth

P = oracle_lib.P
KITTI = (718.856, 718.856, 607.1928, 185.2157)
DIST5 = (-0.28, 0.07, 0.0002, -0.0001, 0.01)            # a TUM-like lens: k1 k2 p1 p2 k3


def _oracle_undistort(oracle, cam4, dist, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty_like(xy)
    c4 = np.array(cam4, np.float32); d = np.array(list(dist) + [0.0] * (8 - len(dist)), np.float32)
    oracle.oracle_undistort_points.restype = None
    oracle.oracle_undistort_points(P(c4), P(d), len(dist), P(xy), len(xy), P(out))
    return out


def _oracle_gray(oracle, img):
    h, w, cn = img.shape
    out = np.empty((h, w), np.uint8)
    oracle.oracle_bgr2gray.restype = None
    oracle.oracle_bgr2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    oracle.oracle_bgr2gray(img.ctypes.data, w, h, img.strides[0], cn, out.ctypes.data)
    return out


def _points(n=3000, seed=0, w=1241, h=376):
    r = np.random.default_rng(seed)
    return (r.random((n, 2)) * [w, h]).astype(np.float32)


@pytest.mark.parametrize("dist", [(), DIST5[:4], DIST5, DIST5 + (0.01, -0.02, 0.003)], ids=["none", "k1k2p1p2", "k3", "rational"])
def test_host_undistort_equals_oracle_bit_for_bit(oracle, dist):
    from ucoslam_cv3_amd.orb import Camera, undistort_points_host

    xy = _points()
    got = undistort_points_host(Camera(*KITTI, dist), xy)
    np.testing.assert_array_equal(got, _oracle_undistort(oracle, KITTI, dist, xy))


def test_undistort_known_answers(oracle):
    from ucoslam_cv3_amd.orb import Camera, undistort_points_host

    xy = _points(2000, 3)
    fx, fy, cx, cy = (np.float32(v) for v in KITTI)
    # no distortion: the reference still goes through the normalised plane and back — the result is the float round trip of misc.cpp:283-291
    # computed here in numpy ((float)(double) then float multiply-add), and within one float ulp of the input, not necessarily the input
    got = undistort_points_host(Camera(*KITTI), xy)
    xn = ((xy[:, 0].astype(np.float64) - np.float64(cx)) * (1.0 / np.float64(fx))).astype(np.float32)
    yn = ((xy[:, 1].astype(np.float64) - np.float64(cy)) * (1.0 / np.float64(fy))).astype(np.float32)
    np.testing.assert_array_equal(got[:, 0], xn * fx + cx)
    np.testing.assert_array_equal(got[:, 1], yn * fy + cy)
    assert np.abs(got - xy).max() < 2e-4
    # radial-only lens: undistorting and re-applying the forward model x_d = x_u (1 + k1 r_u^2) must return the input to within what five
    # fixed-point iterations leave (contraction factor ~ 3 |k1| r^2 per iteration; r < 0.9 on this sensor)
    k1 = -0.05
    u = undistort_points_host(Camera(*KITTI, (k1,)), xy).astype(np.float64)
    xu, yu = (u[:, 0] - KITTI[2]) / KITTI[0], (u[:, 1] - KITTI[3]) / KITTI[1]
    r2 = xu * xu + yu * yu
    back = np.stack([xu * (1 + k1 * r2) * KITTI[0] + KITTI[2], yu * (1 + k1 * r2) * KITTI[1] + KITTI[3]], 1)
    assert np.abs(back - xy).max() < 2e-2
    # the principal point is a fixed point of any distortion model
    pp = undistort_points_host(Camera(*KITTI, DIST5), np.array([[KITTI[2], KITTI[3]]], np.float32))
    assert np.abs(pp - np.float32([KITTI[2], KITTI[3]])).max() < 1e-4


def test_oracle_gray_known_answers(oracle):
    # white, black, the three primaries: (3735 + 19235 + 9798) = 32768 = 2^15 -> exact white; B 29, G 150, R 77 (rounded shares of 255)
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    np.testing.assert_array_equal(_oracle_gray(oracle, px)[0], [255, 0, 29, 150, 76])
    r = np.random.default_rng(1).integers(0, 256, (7, 13, 4), dtype=np.uint8)
    ref = ((r[..., 0].astype(np.int64) * 3735 + r[..., 1].astype(np.int64) * 19235 + r[..., 2].astype(np.int64) * 9798 + 16384) >> 15).astype(np.uint8)
    np.testing.assert_array_equal(_oracle_gray(oracle, np.ascontiguousarray(r)), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("cn,pinned", [(3, False), (4, False), (3, True), (1, False)], ids=["bgr", "bgra", "bgr-pinned", "gray"])
def test_hip_extract_frame_colour_in_undistorted_out(hip_ctx, oracle, cn, pinned):
    """One call = FrameExtractor's cvtColor + detectAndCompute + undistortPoints: keypoints and descriptors equal the oracle extractor's on the
    oracle-converted gray frame (bit-exact), und_xy equals the oracle's undistortion of those keypoints (bit-exact)."""
    import torch
    from ucoslam_cv3_amd.orb import Camera, FeatParams, ORBextractor

    w, h, nf = 1241, 376, 2000
    gray0 = synth.frame(w, h, seed=11)
    if cn == 1:
        img, gray = gray0, gray0
    else:
        r = np.random.default_rng(5)
        img = np.stack([np.clip(gray0.astype(np.int32) + r.integers(-30, 31, gray0.shape), 0, 255).astype(np.uint8) for _ in range(cn)], 2)
        img = np.ascontiguousarray(img)
        gray = _oracle_gray(oracle, img)
    if pinned:
        t = torch.from_numpy(img).pin_memory()
        img = t.numpy()
    ext = ORBextractor.create(hip_ctx).setCamera(Camera(*KITTI, DIST5))
    kps, desc, und = ext.extractFrame(img, FeatParams(nf, 8, 1.2))
    rk, rd = oracle_lib.orb_extract(oracle, gray, nf, 8, 1.2, True)
    assert len(kps) == len(rk) > 0.5 * nf
    for f in ("octave", "x", "y", "size", "response", "angle"):
        np.testing.assert_array_equal(kps[f], rk[f], err_msg=f)
    np.testing.assert_array_equal(desc, rd)
    xy = np.stack([kps["x"], kps["y"]], 1)
    np.testing.assert_array_equal(und, _oracle_undistort(oracle, KITTI, DIST5, xy))
    # and the plain entry point is untouched by the camera
    k2, d2 = ext.detectAndCompute(gray, None, FeatParams(nf, 8, 1.2))
    np.testing.assert_array_equal(d2, rd)


@pytest.mark.gpu
def test_hip_extract_frame_small_capacity_does_not_write_past_a_pinned_und_array(hip_ctx):
    """a pinned und_xy of `cap` entries beside pageable keypoint / descriptor arrays — the launch runs at maxFeatures slots, so the
    positions must go through the extractor's own block: UH_ECAPACITY comes back and the bytes behind the caller's array are untouched."""
    import ctypes as C

    import torch
    from ucoslam_cv3_amd import _lib
    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, Camera, FeatParams, ORBextractor

    L = _lib.lib()
    w, h, cap = 1241, 376, 100
    img = synth.frame(w, h, seed=12)
    ext = ORBextractor.create(hip_ctx).setCamera(Camera(*KITTI, DIST5))
    fp = FeatParams(2000, 8, 1.2)
    _lib.check(L.uh_orb_set_params(ext._h, C.byref(fp)))
    kps = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    und_t = torch.full((cap * 2 + 4096,), 7.0, dtype=torch.float32).pin_memory()   # guard floats behind the cap entries
    und = und_t.numpy()
    n = C.c_int(0)
    rc = L.uh_orb_extract_frame(ext._h, img.ctypes.data_as(C.c_void_p), w, h, w, 1, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                und.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    hip_ctx.synchronize()
    assert rc == _lib.UH_ECAPACITY and n.value > cap
    assert (und[2 * cap:] == 7.0).all()
