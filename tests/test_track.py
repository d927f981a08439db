"""uh_track_pose (csrc/track.hpp) — the tracker's pose estimation of one frame as one call — against the same steps through the four
operators (uh_projmatch_match_prev -> look-ups -> uh_pnp_solve -> decision -> uh_projmatch_match -> union + uh_filter_ambiguous +
look-ups -> uh_pnp_solve; system.cpp:5930-6954, the host logic of examples/tracker_frame.cpp): every list, flag, count and both
poses bit for bit."""
import ctypes as C

import numpy as np
import pytest

import synth


def _scene(hip_ctx, seed, host_tree, n_prev=800, n_map=3000, pose_noise=0.0):
    from ucoslam_cv3_amd.orb import Camera, DeviceFrame, FeatParams, ORBextractor
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    rng = np.random.default_rng(seed)
    img = synth.frame(1241, 376, seed=seed)
    ext = ORBextractor(hip_ctx)
    fp = FeatParams(maxFeatures=2000, nOctaveLevels=8, scaleFactor=1.2)
    cam = Camera(718.856, 718.856, 607.19, 185.22, ())
    ext.setCamera(cam)
    fr = DeviceFrame(hip_ctx).setTreeBuilder(host_tree)
    kps, desc, und = ext.extractFrameDev(img, fr, fp)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    ukp = kps.copy()
    ukp["x"], ukp["y"] = und[:, 0], und[:, 1]
    # the map: points behind the frame's own keypoints, seen from a slightly different pose (as examples/tracker_frame.cpp builds its scenes)
    a_ = 0.01
    R = np.array([[np.cos(a_), 0, np.sin(a_)], [0, 1, 0], [-np.sin(a_), 0, np.cos(a_)]])
    t = np.array([0.3, -0.05, 0.1])
    pick = rng.integers(0, len(kps), n_map)
    z = rng.uniform(4, 40, n_map)
    uv = und[pick].astype(np.float64) + rng.normal(0, 0.7, (n_map, 2))
    Xc = np.stack([(uv[:, 0] - cam.cx) / cam.fx * z, (uv[:, 1] - cam.cy) / cam.fy * z, z], 1)
    Xw = (Xc - t) @ R
    cc = -R.T @ t
    view = cc - Xw
    dist = np.linalg.norm(view, axis=1)
    nrm = view / dist[:, None] + rng.normal(0, 0.3, (n_map, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lev = np.clip(kps["octave"][pick] + rng.integers(-1, 2, n_map), 0, 7)
    maxd = dist * sf[lev] * rng.uniform(0.93, 1.07, n_map)
    mdesc = desc[pick] ^ np.packbits(rng.random((n_map, 256)) < 0.04, axis=1, bitorder="little")
    mp = dict(ids=np.arange(10, 10 + n_map, dtype=np.uint32), pos3d=Xw.astype(np.float32), normal=nrm.astype(np.float32),
              min_dist=(maxd / sf[7]).astype(np.float32), max_dist=maxd.astype(np.float32), desc=np.ascontiguousarray(mdesc))
    weight = np.where(rng.random(n_map) < 0.2, np.float32(0.5), np.float32(1.0)).astype(np.float32)
    # the previous frame's items: some of them are map points of the local map too (same id; their own position differs slightly so
    # that the look-up's preference for the map's entry shows), the others carry ids outside it
    rows = np.sort(rng.choice(n_map, n_prev, replace=False)) if n_prev else np.zeros(0, np.int64)
    in_map = rng.random(n_prev) < 0.7
    prev = dict(ids=np.where(in_map, mp["ids"][rows], 100000 + np.arange(n_prev)).astype(np.uint32),
                pos3d=(mp["pos3d"][rows] + rng.normal(0, 0.002, (n_prev, 3))).astype(np.float32),
                octave=kps["octave"][pick[rows]].astype(np.int32), desc=np.ascontiguousarray(mp["desc"][rows]))
    order = np.argsort(prev["ids"], kind="stable")
    prev = {k: np.ascontiguousarray(v[order]) for k, v in prev.items()}
    prev_row = np.where(in_map, rows, -1).astype(np.int32)[order]
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    if pose_noise:
        T[:3, 3] += rng.normal(0, pose_noise, 3)
    pose0 = np.ascontiguousarray(T.astype(np.float32).reshape(16))
    pm = ProjectionMatcher(hip_ctx)
    pm.setFrameDev(fr, sf, cam.fx, cam.fy, cam.cx, cam.cy, (0, 0), (1241, 376), und_kpts=ukp)
    intr = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv_sf = (np.float32(1) / sf).astype(np.float32)
    return dict(pm=pm, fr=fr, ukp=ukp, und=und, prev=prev, prev_row=prev_row, mp=mp, weight=weight, pose0=pose0, intr=intr, inv_sf=inv_sf, keep=(ext, fr))


def _sequence(sc, pnp, min_inliers=30, d1=75.0, r1=15.0, d2=100.0, rt=4.0, rl=15.0):
    """The four operators one after the other with the host's list handling in between (examples/tracker_frame.cpp)."""
    from ucoslam_cv3_amd._lib import lib, np_ptr
    from ucoslam_cv3_amd.projmatch import DMATCH_DTYPE

    pm, prev, mp, ukp = sc["pm"], sc["prev"], sc["mp"], sc["ukp"]
    a = pm.matchFrameToPrevFrame(sc["pose0"], prev["ids"], prev["pos3d"], prev["octave"], prev["desc"], d1, r1)
    m1 = a["matches"]
    pid_to_i = {int(v): i for i, v in enumerate(prev["ids"])}
    it1 = np.array([pid_to_i[int(t)] for t in m1["trainIdx"]], np.int64)
    q1 = m1["queryIdx"]
    s1 = pnp.solvePnp(sc["pose0"], sc["intr"], prev["pos3d"][it1].reshape(-1, 3), np.stack([ukp["x"][q1], ukp["y"][q1]], 1).reshape(-1, 2),
                      sc["inv_sf"][ukp["octave"][q1]], np.ones(len(m1), np.float32))
    tracked = s1["ngood"] >= min_inliers
    pose_map = s1["pose"] if tracked else sc["pose0"]
    b = pm.matchFrameToMapPoints(pose_map, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], d2, rt if tracked else rl)
    m2 = b["matches"]
    union = np.concatenate([m1[s1["bad"][: len(m1)] == 0] if tracked else m1[:0], m2]).astype(DMATCH_DTYPE)
    if len(union):
        union = np.ascontiguousarray(union)
        k = lib().uh_filter_ambiguous(np_ptr(union), len(union), 0)
        assert k >= 0
        union = union[:k]
    mid_to_row = {int(v): i for i, v in enumerate(mp["ids"])}
    p3d = np.zeros((len(union), 3), np.float32)
    w = np.ones(len(union), np.float32)
    for i, tr in enumerate(union["trainIdx"]):
        row = mid_to_row.get(int(tr), -1)
        if row >= 0:
            p3d[i] = mp["pos3d"][row]; w[i] = sc["weight"][row]
        else:
            p3d[i] = prev["pos3d"][pid_to_i[int(tr)]]
    qa = union["queryIdx"]
    s2 = pnp.solvePnp(pose_map, sc["intr"], p3d, np.stack([ukp["x"][qa], ukp["y"][qa]], 1).reshape(-1, 2), sc["inv_sf"][ukp["octave"][qa]], w)
    return dict(matches_prev=m1, bad_prev=s1["bad"][: len(m1)], inliers1=s1["ngood"], iters1=s1["iters"], pose1=s1["pose"], tracked=bool(tracked), matches_map=m2,
                matches_all=union, bad_all=s2["bad"][: len(union)], inliers2=s2["ngood"], iters2=s2["iters"], pose2=s2["pose"])


def _same(f, s, what):
    for k in ("matches_prev", "matches_map", "matches_all"):
        assert f[k].tobytes() == s[k].tobytes(), (what, k, len(f[k]), len(s[k]))
    for k in ("bad_prev", "bad_all", "iters1", "iters2"):
        np.testing.assert_array_equal(f[k], s[k], err_msg=f"{what}: {k}")
    assert f["tracked"] == s["tracked"] and f["inliers1"] == s["inliers1"] and f["inliers2"] == s["inliers2"], (what, f["tracked"], f["inliers1"], f["inliers2"], s["inliers1"], s["inliers2"])
    if len(s["matches_prev"]):
        assert f["pose1"].tobytes() == np.asarray(s["pose1"], np.float32).tobytes(), what
    assert f["pose2"].tobytes() == np.asarray(s["pose2"], np.float32).tobytes(), what


@pytest.mark.gpu
@pytest.mark.parametrize("host_tree", [False, True], ids=["device_tree", "host_tree"])
def test_track_pose_equals_the_four_operators(hip_ctx, host_tree):
    from ucoslam_cv3_amd.pnp import PnPSolver

    pnp = PnPSolver(hip_ctx)
    for seed, kw in ((5, {}), (6, {}), (7, dict(n_prev=300, n_map=1200)), (8, dict(pose_noise=0.8)), (11, dict(n_prev=1500, n_map=6500))):   # (the last: working lists too long for LDS)
        sc = _scene(hip_ctx, seed, host_tree, **kw)
        s = _sequence(sc, pnp)
        f = sc["pm"].trackPose(pnp, sc["pose0"], sc["intr"], sc["inv_sf"], sc["prev"], sc["mp"], prev_map_row=sc["prev_row"], map_weight=sc["weight"])
        _same(f, s, f"seed {seed}")
        if not kw.get("pose_noise"):
            assert f["tracked"] and f["inliers2"] > 100, (seed, f["inliers1"], f["inliers2"])
        f2 = sc["pm"].trackPose(pnp, sc["pose0"], sc["intr"], sc["inv_sf"], sc["prev"], sc["mp"], prev_map_row=sc["prev_row"], map_weight=sc["weight"])   # again: buffers reused
        _same(f2, s, f"seed {seed} (second call)")


@pytest.mark.gpu
def test_track_pose_when_the_first_solve_fails_and_with_empty_sets(hip_ctx):
    from ucoslam_cv3_amd.pnp import PnPSolver

    pnp = PnPSolver(hip_ctx)
    sc = _scene(hip_ctx, 9, True)
    # a threshold nobody reaches: the first matches are dropped, the predicted pose stays, the map is searched with the wide radius
    s = _sequence(sc, pnp, min_inliers=100000)
    f = sc["pm"].trackPose(pnp, sc["pose0"], sc["intr"], sc["inv_sf"], sc["prev"], sc["mp"], prev_map_row=sc["prev_row"], map_weight=sc["weight"], min_inliers=100000)
    assert not f["tracked"]
    _same(f, s, "lost")
    # no previous-frame items at all / no map points at all
    for n_prev, n_map in ((0, 1500), (600, 0)):
        sc = _scene(hip_ctx, 10 + n_prev, False, n_prev=n_prev, n_map=max(n_map, 700))
        if n_map == 0:
            sc["mp"] = {k: v[:0] for k, v in sc["mp"].items()}
            sc["weight"] = sc["weight"][:0]
            sc["prev_row"] = np.full(len(sc["prev"]["ids"]), -1, np.int32)
        s = _sequence(sc, pnp)
        f = sc["pm"].trackPose(pnp, sc["pose0"], sc["intr"], sc["inv_sf"], sc["prev"], sc["mp"], prev_map_row=sc["prev_row"], map_weight=sc["weight"])
        _same(f, s, f"n_prev {n_prev} n_map {n_map}")


@pytest.mark.gpu
def test_track_pose_on_a_frame_without_keypoints(hip_ctx):
    """A blank image: no keypoints, hence no matches — both poses come back as the predicted one, as through the four operators."""
    from ucoslam_cv3_amd.orb import Camera, DeviceFrame, FeatParams, ORBextractor
    from ucoslam_cv3_amd.pnp import PnPSolver
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    donor = _scene(hip_ctx, 5, True, n_prev=200, n_map=500)
    ext = ORBextractor(hip_ctx)
    ext.setCamera(Camera(718.856, 718.856, 607.19, 185.22, ()))
    fr = DeviceFrame(hip_ctx).setTreeBuilder(True)
    kps, desc, und = ext.extractFrameDev(np.full((376, 1241), 90, np.uint8), fr, FeatParams(maxFeatures=2000, nOctaveLevels=8, scaleFactor=1.2))
    assert len(kps) == 0
    pm = ProjectionMatcher(hip_ctx)
    sf = (np.float32(1) / donor["inv_sf"]).astype(np.float32)
    pm.setFrameDev(fr, sf, 718.856, 718.856, 607.19, 185.22, (0, 0), (1241, 376), und_kpts=kps)
    pnp = PnPSolver(hip_ctx)
    f = pm.trackPose(pnp, donor["pose0"], donor["intr"], donor["inv_sf"], donor["prev"], donor["mp"], prev_map_row=donor["prev_row"], map_weight=donor["weight"])
    assert len(f["matches_prev"]) == 0 and len(f["matches_map"]) == 0 and len(f["matches_all"]) == 0 and not f["tracked"] and f["inliers1"] == 0 and f["inliers2"] == 0
    assert f["pose1"].tobytes() == donor["pose0"].tobytes() and f["pose2"].tobytes() == donor["pose0"].tobytes()


@pytest.mark.gpu
def test_track_pose_needs_a_device_frame(hip_ctx):
    from ucoslam_cv3_amd._lib import UcoslamHipError
    from ucoslam_cv3_amd.pnp import PnPSolver
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    sc = _scene(hip_ctx, 5, True, n_prev=50, n_map=100)
    pm = ProjectionMatcher(hip_ctx)
    sf = (np.float32(1) / sc["inv_sf"]).astype(np.float32)
    pm.setFrame(sc["ukp"], np.zeros((len(sc["ukp"]), 32), np.uint8), sf, 718.856, 718.856, 607.19, 185.22, (0, 0), (1241, 376))
    with pytest.raises(UcoslamHipError, match="device-resident frame"):
        pm.trackPose(PnPSolver(hip_ctx), sc["pose0"], sc["intr"], sc["inv_sf"], sc["prev"], sc["mp"])
