"""Frame::create_kdtree on the device (csrc/kdbuild.hpp, uh_orb_extract_frame_dev, uh_projmatch_set_frame_dev).

The device builder must produce, node for node and leaf for leaf, the tree the host restatement builds — which
tests/test_projmatch.py / test_sweep_golden.py hold against oracle/proj_oracle.cpp, pinned to the REAL picoflann.h by
tests/golden/kdtree_golden.npz and sweep_golden.npz.  CPU part: the restated data movement of libstdc++'s std::sort (picoflann's
fallback) against std::sort itself."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth
from test_projmatch import _same_tree
from test_projmatch_oracle import _clouds


def _std_sort_perm(oracle, keys):
    keys = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(max(len(keys), 1), np.uint32)
    oracle.oracle_std_sort_perm.restype = None
    oracle.oracle_std_sort_perm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    oracle.oracle_std_sort_perm(oracle_lib.P(keys), len(keys), oracle_lib.P(perm))
    return perm[: len(keys)]


def _adversary(n):
    """McIlroy's anti-quicksort adversary played against a median-of-three quicksort: the keys it freezes drive libstdc++'s introsort
    to its depth limit (the heapsort branch) — checked: the restatement's heap branch runs on these."""
    val, state = [-1] * n, dict(nsolid=0, cand=0)

    def less(x, y):
        if val[x] < 0 and val[y] < 0:
            if x == state["cand"]:
                val[x] = state["nsolid"]
            else:
                val[y] = state["nsolid"]
            state["nsolid"] += 1
        if val[x] < 0:
            state["cand"] = x
        elif val[y] < 0:
            state["cand"] = y
        vx = n if val[x] < 0 else val[x]
        vy = n if val[y] < 0 else val[y]
        return vx < vy

    def qsort(a):   # median of (first + 1, mid, last - 1) like libstdc++, recursion instead of introsort's loop
        if len(a) <= 16:
            return
        mid = len(a) // 2
        x, y, z = a[1], a[mid], a[-1]
        m = y if (less(x, y) and less(y, z)) or (less(z, y) and less(y, x)) else (z if (less(x, z) and less(z, y)) or (less(y, z) and less(z, x)) else x)
        lo = [e for e in a if e != m and less(e, m)]
        hi = [e for e in a if e != m and not less(e, m)]
        qsort(lo)
        qsort(hi)

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(20000)
    try:
        qsort(list(range(n)))
    finally:
        sys.setrecursionlimit(old)
    return np.array([n if v < 0 else v for v in val], np.float32)


def test_restated_std_sort_equals_libstdcxx(oracle):
    from ucoslam_cv3_amd.projmatch import kdtree_sort_restated_host

    rng = np.random.default_rng(3)
    cases = []
    for n in list(range(0, 40)) + [63, 64, 65, 100, 257, 1000, 2000, 4096]:
        cases.append(rng.integers(0, 7, n).astype(np.float32))            # heavy ties
        cases.append(rng.random(n).astype(np.float32))                    # distinct
        cases.append(np.arange(n, dtype=np.float32))                      # sorted
        cases.append(np.arange(n, dtype=np.float32)[::-1].copy())         # reversed
        cases.append((np.minimum(np.arange(n), n - 1 - np.arange(n)) // 3).astype(np.float32))   # organ pipe with ties
    for n in (17, 33, 100, 500, 1500):
        cases.append(_adversary(n))
    for k in cases:
        np.testing.assert_array_equal(kdtree_sort_restated_host(k), _std_sort_perm(oracle, k))


def _more_clouds():
    rng = np.random.default_rng(21)
    out = dict(_clouds())
    for n in (12, 16, 17, 19, 20, 23, 33, 64, 65, 199, 200, 201, 399, 400, 1023, 1024, 1025, 3000, 4096):
        out[f"rand{n}"] = (rng.random((n, 2)) * [1241, 376]).astype(np.float32)
    # what an extractor leaves (19 px border): non-zero coordinates within 11 binades — the builders' exact-sum route (kdbuild.hpp sweep_levels)
    for n in (11, 21, 40, 150, 199, 200, 201, 398, 777, 1200, 1999, 2000, 2500, 4000):
        out[f"inrange{n}"] = (rng.random((n, 2)) * [1203, 338] + 19).astype(np.float32)
    # integer pixel positions of level 0 (zero-distortion camera): columns and rows repeat
    out["pixels2000"] = np.stack([rng.integers(19, 1222, 2000), rng.integers(19, 357, 2000)], 1).astype(np.float32)
    out["pixels_scaled"] = (np.stack([rng.integers(0, 200, 3000), rng.integers(0, 60, 3000)], 1).astype(np.float32) + np.float32(0.5)) * \
        (np.float32(1.2) ** rng.integers(0, 8, (3000, 1)).astype(np.float32))
    out["two_values"] = np.stack([rng.integers(0, 2, 900).astype(np.float32) * 10, rng.integers(0, 2, 900).astype(np.float32) * 7], 1)
    out["all_equal_big"] = np.full((1000, 2), 3.25, np.float32)
    out["row"] = np.stack([rng.random(1200).astype(np.float32) * 1000, np.full(1200, -4.0, np.float32)], 1)
    out["negative"] = (rng.normal(0, 300, (2500, 2))).astype(np.float32)
    out["skewed"] = (rng.random((2000, 2)) ** 6 * [1241, 376]).astype(np.float32)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [0, 256, 512])
@pytest.mark.parametrize("name", list(_more_clouds().keys()))
def test_device_kdtree_equals_host_builder(hip_ctx, name, threads):
    from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host

    xy = _more_clouds()[name]
    mine = kdtree_build_dev(hip_ctx, xy, threads)
    ref = kdtree_build_host(xy)
    assert len(mine["nodes"]) == len(ref["nodes"])
    for f in ("left", "right", "col", "divlow", "divhigh", "leaf_begin", "leaf_count"):
        np.testing.assert_array_equal(mine["nodes"][f], ref["nodes"][f], err_msg=f"{name}: {f}")
    np.testing.assert_array_equal(mine["leaf_idx"], ref["leaf_idx"])
    np.testing.assert_array_equal(mine["root_box"], ref["root_box"])
    assert mine["depth"] == ref["depth"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uniform", "ties", "small"])
def test_device_kdtree_equals_the_pinned_oracle_on_the_golden_clouds(hip_ctx, oracle, case):
    """The clouds of tests/golden/kdtree_golden.npz (hits, order and distances recorded from the real picoflann.h): the device tree is the
    oracle's tree, which test_projmatch_oracle.py holds against those recordings."""
    import os

    from ucoslam_cv3_amd.projmatch import kdtree_build_dev

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "kdtree_golden.npz"))
    xy = g[f"{case}_xy"]
    _same_tree(kdtree_build_dev(hip_ctx, xy), oracle_lib.KdOracle(oracle, "oracle_kd", xy).export())


@pytest.mark.gpu
def test_device_kdtree_fuzz_against_host_builder(hip_ctx):
    from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host

    rng = np.random.default_rng(99)
    for it in range(300):
        n = int(rng.integers(0, 4097)) if it % 3 else int(rng.integers(0, 120))
        mode = it % 5
        if mode == 0:
            xy = (rng.random((n, 2)) * [1241, 376]).astype(np.float32)
        elif mode == 1:
            xy = rng.integers(0, max(2, int(rng.integers(2, 80))), (n, 2)).astype(np.float32)
        elif mode == 2:
            xy = (rng.integers(0, 300, (n, 2)).astype(np.float32) + np.float32(0.5)) * np.float32(1.2) ** rng.integers(0, 8, (n, 1)).astype(np.float32)
        elif mode == 3:
            xy = rng.normal([600, 180], [rng.random() * 50 + 0.01, rng.random() * 5 + 0.01], (n, 2)).astype(np.float32)
        else:
            xy = np.stack([rng.random(n).astype(np.float32) * 1e-3, rng.integers(0, 3, n).astype(np.float32)], 1)
        mine = kdtree_build_dev(hip_ctx, xy, [0, 256, 512][it % 3])
        ref = kdtree_build_host(xy)
        assert mine["nodes"].tobytes() == ref["nodes"].tobytes(), (it, n, mode)
        assert mine["leaf_idx"].tobytes() == ref["leaf_idx"].tobytes(), (it, n, mode)
        assert mine["root_box"].tobytes() == ref["root_box"].tobytes() and mine["depth"] == ref["depth"], (it, n, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("dist", [(), (-0.28, 0.07, 0.0004, -0.0003, 0.01)], ids=["rectified", "distorted"])
def test_frame_stays_on_the_device_between_extractor_and_projection_matcher(hip_ctx, oracle, dist):
    """uh_orb_extract_frame_dev + uh_projmatch_set_frame_dev against the host route (extractFrame -> setFrame): same tree, same leaf
    records, same matches from both projection searches."""
    from ucoslam_cv3_amd.orb import Camera, DeviceFrame, FeatParams, ORBextractor
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher, kdtree_build_host

    img = synth.frame(1241, 376, seed=5)
    ext = ORBextractor(hip_ctx)
    fp = FeatParams(maxFeatures=2000, nOctaveLevels=8, scaleFactor=1.2)
    cam = Camera(718.856, 718.856, 607.19, 185.22, dist)
    ext.setCamera(cam)
    kps0, desc0, und0 = ext.extractFrame(img, fp)
    fr = DeviceFrame(hip_ctx)
    kps, desc, und = ext.extractFrameDev(img, fr, fp)
    np.testing.assert_array_equal(kps, kps0)
    np.testing.assert_array_equal(desc, desc0)
    np.testing.assert_array_equal(und, und0)
    t = fr.tree()
    ref = kdtree_build_host(und)
    assert t["nodes"].tobytes() == ref["nodes"].tobytes()
    np.testing.assert_array_equal(t["leaf_idx"], ref["leaf_idx"])
    np.testing.assert_array_equal(t["root_box"], ref["root_box"])
    assert t["depth"] == ref["depth"]
    np.testing.assert_array_equal(t["leaf_xy"], und[ref["leaf_idx"]])
    np.testing.assert_array_equal(t["leaf_octave"], kps["octave"][ref["leaf_idx"]])
    # the matchers on both routes
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    ukp = kps.copy()
    ukp["x"], ukp["y"] = und[:, 0], und[:, 1]
    # a map made from the frame's own features (as examples/tracker_frame.cpp does): points behind two thirds of the keypoints
    rng = np.random.default_rng(4)
    n_pts = 3000
    a_ = 0.01
    R = np.array([[np.cos(a_), 0, np.sin(a_)], [0, 1, 0], [-np.sin(a_), 0, np.cos(a_)]])
    t = np.array([0.3, -0.05, 0.1])
    pick = rng.integers(0, len(kps), n_pts)
    z = rng.uniform(4, 40, n_pts)
    uv = und[pick].astype(np.float64) + rng.normal(0, 1.0, (n_pts, 2))
    Xc = np.stack([(uv[:, 0] - cam.cx) / cam.fx * z, (uv[:, 1] - cam.cy) / cam.fy * z, z], 1)
    Xw = (Xc - t) @ R
    cc = -R.T @ t
    view = cc - Xw
    dist = np.linalg.norm(view, axis=1)
    nrm = view / dist[:, None] + rng.normal(0, 0.3, (n_pts, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lev = np.clip(kps["octave"][pick] + rng.integers(-1, 2, n_pts), 0, 7)
    maxd = dist * sf[lev] * rng.uniform(0.93, 1.07, n_pts)
    mdesc = desc[pick] ^ np.packbits(rng.random((n_pts, 256)) < 0.04, axis=1, bitorder="little")
    mp = dict(ids=np.arange(10, 10 + n_pts, dtype=np.uint32), pos3d=Xw.astype(np.float32), normal=nrm.astype(np.float32),
              min_dist=(maxd / sf[7]).astype(np.float32), max_dist=maxd.astype(np.float32), desc=np.ascontiguousarray(mdesc),
              octave=kps["octave"][pick].astype(np.int32))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    pose = np.ascontiguousarray(T.astype(np.float32).reshape(16))
    host, dev = ProjectionMatcher(hip_ctx), ProjectionMatcher(hip_ctx)
    host.setFrame(ukp, desc, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376))
    dev.setFrameDev(fr, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376))
    a = host.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    b = dev.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    for k in ("matches", "best_kp", "best_dist", "visible"):
        np.testing.assert_array_equal(a[k], b[k])
    assert (a["best_kp"] >= 0).sum() > 50
    a = host.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    b = dev.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    for k in ("matches", "best_kp", "best_dist"):
        np.testing.assert_array_equal(a[k], b[k])
    # third route: the frame resident, its tree built by the host core and uploaded into the frame object (no build launch)
    fr_h = DeviceFrame(hip_ctx).setTreeBuilder(True)
    kps_h, desc_h, und_h = ext.extractFrameDev(img, fr_h, fp)
    np.testing.assert_array_equal(kps_h, kps0)
    np.testing.assert_array_equal(und_h, und0)
    hyb = ProjectionMatcher(hip_ctx)
    hyb.setFrameDev(fr_h, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376), und_kpts=ukp)
    a = host.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    b = hyb.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    for k in ("matches", "best_kp", "best_dist", "visible"):
        np.testing.assert_array_equal(a[k], b[k])
    a = host.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    b = hyb.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    for k in ("matches", "best_kp", "best_dist"):
        np.testing.assert_array_equal(a[k], b[k])
    hyb.setFrameDev(fr_h, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376), und_kpts=ukp)   # again: the staging block is reused behind its completion word
    b = hyb.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    np.testing.assert_array_equal(a["matches"], b["matches"])
    # the extraction in two halves: the undistorted keypoints (position + octave) arrive ahead of the descriptors, the tree is built from them
    early = ext.extractFrameDevBegin(img, fr_h, fp)
    hyb.setFrameDev(fr_h, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376), und_kpts=early)
    kps_e, desc_e, und_e = ext.extractFrameDevEnd()
    np.testing.assert_array_equal(kps_e, kps0); np.testing.assert_array_equal(desc_e, desc0); np.testing.assert_array_equal(und_e, und0)
    np.testing.assert_array_equal(np.stack([early["x"], early["y"]], 1), und0)
    np.testing.assert_array_equal(early["octave"], kps0["octave"])
    b = hyb.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    np.testing.assert_array_equal(a["matches"], b["matches"])
    # a second extraction into the same frame object replaces it (and an empty image yields the empty frame)
    img2 = synth.frame(1241, 376, seed=6, shift=(3, 1))
    kps2, desc2, und2 = ext.extractFrameDev(img2, fr, fp)
    t2 = fr.tree()
    ref2 = kdtree_build_host(und2)
    assert t2["nodes"].tobytes() == ref2["nodes"].tobytes() and t2["leaf_idx"].tobytes() == ref2["leaf_idx"].tobytes()


@pytest.mark.gpu
def test_device_frame_upload_and_split_extraction_protocol(hip_ctx):
    """uh_dev_frame_upload (a frame from elsewhere) gives the tree of the host builder for both builders; uh_orb_extract_frame_dev_begin / _end:
    a second _begin before _end is refused, an empty image yields the empty frame through both halves."""
    from ucoslam_cv3_amd._lib import UcoslamHipError
    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, Camera, DeviceFrame, FeatParams, ORBextractor
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher, kdtree_build_host

    rng = np.random.default_rng(31)
    n = 1777
    kp = np.zeros(n, KEYPOINT_DTYPE)
    kp["x"], kp["y"] = rng.random(n) * 1200 + 19, rng.random(n) * 330 + 19
    kp["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ref = kdtree_build_host(np.stack([kp["x"], kp["y"]], 1))
    fr = DeviceFrame(hip_ctx).upload(kp, desc)          # tree by the build launches
    t = fr.tree()
    assert t["nodes"].tobytes() == ref["nodes"].tobytes() and t["leaf_idx"].tobytes() == ref["leaf_idx"].tobytes()
    np.testing.assert_array_equal(t["leaf_octave"], kp["octave"][ref["leaf_idx"]])
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    fr_h = DeviceFrame(hip_ctx).setTreeBuilder(True).upload(kp, desc)   # tree by the host core inside setFrameDev
    a, b = ProjectionMatcher(hip_ctx), ProjectionMatcher(hip_ctx)
    a.setFrameDev(fr, sf, 718.856, 718.856, 607.19, 185.22, (0, 0), (1241, 376))
    b.setFrameDev(fr_h, sf, 718.856, 718.856, 607.19, 185.22, (0, 0), (1241, 376), und_kpts=kp)
    fr_s, mp, pose = synth.proj_problem(0, 400, 3)
    mp["pos3d"][:200] = np.stack([(kp["x"][:200] - 607.19) / 718.856 * 10, (kp["y"][:200] - 185.22) / 718.856 * 10, np.full(200, 10.0)], 1)
    mp["desc"][:200] = desc[:200]
    eye = np.eye(4, dtype=np.float32).reshape(16)
    ra = a.matchFrameToPrevFrame(eye, mp["ids"], mp["pos3d"], np.clip(mp["octave"], 0, 7), mp["desc"], 100.0, 15.0)
    rb = b.matchFrameToPrevFrame(eye, mp["ids"], mp["pos3d"], np.clip(mp["octave"], 0, 7), mp["desc"], 100.0, 15.0)
    assert ra["matches"].tobytes() == rb["matches"].tobytes() and (ra["best_kp"] == rb["best_kp"]).all()
    # the split extraction's protocol
    ext = ORBextractor(hip_ctx)
    ext.setCamera(Camera(718.856, 718.856, 607.19, 185.22, ()))
    fp = FeatParams(maxFeatures=500, nOctaveLevels=8, scaleFactor=1.2)
    d = DeviceFrame(hip_ctx).setTreeBuilder(True)
    early = ext.extractFrameDevBegin(synth.frame(640, 480, seed=2), d, fp)
    with pytest.raises(UcoslamHipError, match="_begin without its _end"):
        ext.extractFrameDev(synth.frame(640, 480, seed=3), d, fp)
    kps, dsc, und = ext.extractFrameDevEnd()
    np.testing.assert_array_equal(np.stack([early["x"], early["y"]], 1), und)
    np.testing.assert_array_equal(early["octave"], kps["octave"])
    assert len(ext.extractFrameDevBegin(np.full((480, 640), 77, np.uint8), d, fp)) == 0
    k0, d0, u0 = ext.extractFrameDevEnd()
    assert len(k0) == 0
