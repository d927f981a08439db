"""Projection matcher (Map::matchFrameToMapPoints, map.cpp:651-770): product vs oracle, bit-exact (indices, float distances,
visibility flags, DMatch list).  The kd-tree the product builds on the host is compared node by node with the oracle's
restatement, which tests/test_projmatch_oracle.py pins against the real picoflann."""
import numpy as np
import pytest

import oracle_lib
import synth
from test_projmatch_oracle import _clouds


def _same_tree(mine, ref):
    assert len(mine["nodes"]) == len(ref["col"])
    nd = mine["nodes"]
    inner = nd["left"] >= 0
    np.testing.assert_array_equal(nd["left"], ref["left"])
    np.testing.assert_array_equal(nd["right"], ref["right"])
    np.testing.assert_array_equal(nd["col"][inner], ref["col"][inner])
    np.testing.assert_array_equal(nd["divlow"][inner], ref["divlow"][inner])
    np.testing.assert_array_equal(nd["divhigh"][inner], ref["divhigh"][inner])
    np.testing.assert_array_equal(nd["leaf_begin"][~inner], ref["leaf_begin"][~inner])
    np.testing.assert_array_equal(nd["leaf_count"][~inner], ref["leaf_count"][~inner])
    np.testing.assert_array_equal(mine["leaf_idx"], ref["leaf_idx"])
    np.testing.assert_array_equal(mine["root_box"], ref["root_bbox"])


@pytest.mark.parametrize("scalar", [False, True], ids=["default", "no-avx512"])
@pytest.mark.parametrize("name", list(_clouds().keys()))
def test_host_kdtree_build_equals_oracle(oracle, name, scalar, monkeypatch):
    from ucoslam_cv3_amd.projmatch import kdtree_build_host

    if scalar:   # the builder's AVX-512 Hoare pass is taken where the host has it: the scalar form must stay covered too
        monkeypatch.setenv("UH_KD_NO_AVX512", "1")
    xy = _clouds()[name]
    mine = kdtree_build_host(xy)
    if len(xy) == 0:
        assert len(mine["nodes"]) == 0
        return
    ref = oracle_lib.KdOracle(oracle, "oracle_kd", xy).export()
    _same_tree(mine, ref)
    assert mine["depth"] <= 60


def test_oracle_proj_match_properties(oracle):
    """Sanity of the restated matcher itself: visible set = brute-force visibility tests in numpy; every match satisfies the
    radius / octave / distance rules; one match per keypoint after filter_ambiguous_query."""
    fr, mp, pose = synth.proj_problem(1500, 2500, 3)
    r = oracle_lib.proj_match(oracle, fr, mp, pose, 100.0, 15.0)
    assert 500 < r["visible"].sum() < 2500 and len(r["matches"]) > 200
    m = r["matches"]
    assert len(np.unique(m["queryIdx"])) == len(m)
    id2row = {int(i): k for k, i in enumerate(mp["ids"])}
    T = pose.reshape(4, 4).astype(np.float64)
    for mm in m[:200]:
        row = id2row[int(mm["trainIdx"])]
        assert r["visible"][row] == 1 and r["best_kp"][row] == mm["queryIdx"]
        X = T[:3, :3] @ mp["pos3d"][row].astype(np.float64) + T[:3, 3]
        u = np.array([fr["fx"] * X[0] / X[2] + fr["cx"], fr["fy"] * X[1] / X[2] + fr["cy"]])
        k = fr["und_kpts"][mm["queryIdx"]]
        assert np.hypot(u[0] - k["x"], u[1] - k["y"]) < 15.0 * 1.2 ** 7 * 1.6 + 1e-3
        hd = int(np.unpackbits(fr["desc"][mm["queryIdx"]] ^ mp["desc"][row]).sum())
        assert hd == mm["distance"] and hd < 100


def test_oracle_proj_match_prev_properties(oracle):
    """The restated tracker search (system.cpp:5930-6460) against an independent numpy/python restatement on a small problem:
    brute-force candidates (no kd-tree) in ascending keypoint order cannot reproduce the candidate ORDER, so the comparison is
    made on problems where the order cannot matter (full-entropy descriptors: no equal distances among candidates)."""
    fr, mp, pose = synth.proj_problem(600, 900, 11)
    minDesc, maxRepj = 75.0, 7.5
    r = oracle_lib.proj_match_prev(oracle, fr, mp, pose, minDesc, maxRepj)
    T = pose.reshape(4, 4)
    kp, sf = fr["und_kpts"], fr["scale_factors"]
    f32 = np.float32
    exp_kp = np.full(len(mp["ids"]), -1, np.int32)
    exp_d = np.full(len(mp["ids"]), np.finfo(np.float32).max, np.float32)
    for m in range(len(mp["ids"])):
        P = mp["pos3d"][m]
        z = f32(f32(f32(P[0] * T[2, 0]) + f32(P[1] * T[2, 1])) + f32(P[2] * T[2, 2])) + T[2, 3]
        if z < 0:
            continue
        x = f32(f32(f32(P[0] * T[0, 0]) + f32(P[1] * T[0, 1])) + f32(P[2] * T[0, 2])) + T[0, 3]
        y = f32(f32(f32(P[0] * T[1, 0]) + f32(P[1] * T[1, 1])) + f32(P[2] * T[1, 2])) + T[1, 3]
        iz = f32(1.0 / np.float64(z))
        px = f32(f32(f32(f32(fr["fx"]) * x) * iz) + f32(fr["cx"]))
        py = f32(f32(f32(f32(fr["fy"]) * y) * iz) + f32(fr["cy"]))
        if not (px >= fr["min_xy"][0] and py >= fr["min_xy"][1] and px < fr["max_xy"][0] and py < fr["max_xy"][1]):
            continue
        oc = int(mp["octave"][m])
        rad = np.float64(f32(f32(maxRepj) * sf[oc]))
        dx, dy = np.float64(px) - kp["x"].astype(np.float64), np.float64(py) - kp["y"].astype(np.float64)
        cand = np.nonzero((dx * dx + dy * dy < rad * rad) & (kp["octave"] == oc))[0]
        if len(cand) == 0:
            continue
        hd = np.unpackbits(fr["desc"][cand] ^ mp["desc"][m][None, :], axis=1).sum(1).astype(np.float32)
        if len(np.unique(hd)) != len(hd):
            continue   # equal distances: the outcome may depend on the tree's candidate order (covered by GPU-vs-oracle)
        # no demotion: `second` only ever takes values that lost against the best of their moment — order dependent in general,
        # but with the smallest distance FIRST or the candidates sorted the rule reduces to: second = min over the rest, provided
        # no earlier, larger best was skipped.  Evaluate the sequential rule over every order the tree could produce instead:
        # accept only if the outcome is order-independent, else skip the item (rare) — the GPU-vs-oracle test covers order.
        outcomes = set()
        for order in (np.arange(len(cand)), np.arange(len(cand))[::-1], np.argsort(hd), np.argsort(-hd)):
            best, second, bk = f32(np.float64(f32(minDesc)) + 0.01), np.finfo(np.float32).max, -1
            for j in order:
                if hd[j] < best:
                    best, bk = hd[j], int(cand[j])
                elif hd[j] < second:
                    second = hd[j]
            outcomes.add((bk, float(best)) if bk != -1 and np.float64(best) < 0.7 * np.float64(second) else (-1, 0.0))
        if len(outcomes) != 1:
            continue
        bk, best = outcomes.pop()
        assert r["best_kp"][m] == bk, f"item {m}"
        if bk >= 0:
            assert r["best_dist"][m] == np.float32(best)
            exp_kp[m] = bk
    assert (r["best_kp"] >= 0).sum() > 100
    mt = r["matches"]
    assert len(np.unique(mt["queryIdx"])) == len(mt)            # filter_ambiguous_query: one match per keypoint of the frame
    id2row = {int(i): k for k, i in enumerate(mp["ids"])}
    for mm in mt:
        row = id2row[int(mm["trainIdx"])]
        assert r["best_kp"][row] == mm["queryIdx"] and r["best_dist"][row] == mm["distance"]
        rivals = np.nonzero(r["best_kp"] == mm["queryIdx"])[0]
        assert mm["distance"] == r["best_dist"][rivals].min()


# ------------------------------------------------------------------------------------------------ GPU
CASES = [dict(n_kpts=2000, n_pts=3000, seed=0), dict(n_kpts=2000, n_pts=3000, seed=1, low_entropy=True),
         dict(n_kpts=4000, n_pts=10000, seed=2, w=640, h=480), dict(n_kpts=4000, n_pts=6000, seed=3, low_entropy=True, pose_noise=0.01),
         dict(n_kpts=7, n_pts=50, seed=4), dict(n_kpts=11, n_pts=64, seed=5), dict(n_kpts=300, n_pts=65, seed=6, n_levels=4)]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "_".join(f"{k}{v}" for k, v in c.items()))
def test_hip_projmatch_matches_oracle(hip_ctx, oracle, cfg):
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    fr, mp, pose = synth.proj_problem(**cfg)
    le = cfg.get("low_entropy", False)
    pm = ProjectionMatcher(hip_ctx)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    _same_tree(pm.debug_tree(), oracle_lib.KdOracle(oracle, "oracle_kd", np.stack([fr["und_kpts"]["x"], fr["und_kpts"]["y"]], 1)).export())
    for minDesc, maxRepj in ((100.0, 15.0), (8.0, 15.0), (50.0, 2.5)) if not le else ((8.0, 15.0), (3.0, 40.0)):
        got = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], minDesc, maxRepj)
        ref = oracle_lib.proj_match(oracle, fr, mp, pose, minDesc, maxRepj)
        np.testing.assert_array_equal(got["visible"], ref["visible"])
        np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])
        np.testing.assert_array_equal(got["best_dist"][ref["best_kp"] >= 0], ref["best_dist"][ref["best_kp"] >= 0])
        assert got["matches"].tobytes() == ref["matches"].tobytes()
    if cfg["n_kpts"] >= 2000:
        assert len(ref["matches"]) > 100


@pytest.mark.gpu
def test_hip_projmatch_big_frame_path(hip_ctx, oracle, monkeypatch):
    """Frames too large for LDS walk the tree in HBM/L2 (forced here through the test knob)."""
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    monkeypatch.setenv("UH_PROJMATCH_NO_LDS", "1")
    fr, mp, pose = synth.proj_problem(2000, 3000, 8, low_entropy=True)
    pm = ProjectionMatcher(hip_ctx)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    got = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 8.0, 15.0)
    ref = oracle_lib.proj_match(oracle, fr, mp, pose, 8.0, 15.0)
    assert got["matches"].tobytes() == ref["matches"].tobytes() and len(ref["matches"]) > 100
    np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])


@pytest.mark.gpu
def test_hip_projmatch_edge_inputs(hip_ctx, oracle):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    pm = ProjectionMatcher(hip_ctx)
    fr, mp, pose = synth.proj_problem(100, 40, 9)
    with pytest.raises(u.UcoslamHipError):      # no frame yet: loud, no fallback
        pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    # frame without keypoints: everything that projects is visible, nothing matches
    pm.setFrame(fr["und_kpts"][:0], fr["desc"][:0], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    got = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    fr0 = dict(fr, und_kpts=fr["und_kpts"][:0], desc=fr["desc"][:0])
    ref = oracle_lib.proj_match(oracle, fr0, mp, pose, 100.0, 15.0)
    assert len(got["matches"]) == 0 and (got["best_kp"] == -1).all()
    np.testing.assert_array_equal(got["visible"], ref["visible"])
    # no map points
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    e = pm.matchFrameToMapPoints(pose, mp["ids"][:0], mp["pos3d"][:0], mp["normal"][:0], mp["min_dist"][:0], mp["max_dist"][:0], mp["desc"][:0], 100.0, 15.0)
    assert len(e["matches"]) == 0
    with pytest.raises(u.UcoslamHipError):      # a non-positive radius is not a radius search in the reference
        pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 0.0)
    # default projection limits (cv::Point saturates FLT_MAX to INT_MAX)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"])
    got = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    ref = oracle_lib.proj_match(oracle, dict(fr, max_xy=(2 ** 31 - 1, 2 ** 31 - 1)), mp, pose, 100.0, 15.0)
    assert got["matches"].tobytes() == ref["matches"].tobytes()
    np.testing.assert_array_equal(got["visible"], ref["visible"])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "_".join(f"{k}{v}" for k, v in c.items()))
def test_hip_projmatch_prev_frame_matches_oracle(hip_ctx, oracle, cfg):
    """uh_projmatch_match_prev: the tracker's search against the previous frame (system.cpp:5930-6460), bit-exact vs the oracle."""
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    fr, mp, pose = synth.proj_problem(**cfg)
    le = cfg.get("low_entropy", False)
    pm = ProjectionMatcher(hip_ctx)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    total = 0
    for minDesc, maxRepj in ((75.0, 7.5), (100.0, 15.0), (20.0, 2.5)) if not le else ((8.0, 15.0), (3.0, 40.0), (0.0, 7.5)):
        got = pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], minDesc, maxRepj)
        ref = oracle_lib.proj_match_prev(oracle, fr, mp, pose, minDesc, maxRepj)
        np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])
        np.testing.assert_array_equal(got["best_dist"], ref["best_dist"])
        assert got["matches"].tobytes() == ref["matches"].tobytes()
        total += len(ref["matches"])
    if cfg["n_kpts"] >= 2000:
        assert total > 100


@pytest.mark.gpu
def test_hip_projmatch_prev_frame_edge_inputs(hip_ctx, oracle, monkeypatch):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    pm = ProjectionMatcher(hip_ctx)
    fr, mp, pose = synth.proj_problem(400, 300, 13, low_entropy=True)
    with pytest.raises(u.UcoslamHipError):      # no frame yet
        pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 7.5)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    assert len(pm.matchFrameToPrevFrame(pose, mp["ids"][:0], mp["pos3d"][:0], mp["octave"][:0], mp["desc"][:0], 75.0, 7.5)["matches"]) == 0
    bad = mp["octave"].copy()
    bad[7] = len(fr["scale_factors"])
    with pytest.raises(u.UcoslamHipError):      # scaleFactors[octave] out of range: undefined in the reference, refused here
        pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], bad, mp["desc"], 75.0, 7.5)
    with pytest.raises(u.UcoslamHipError):
        pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 0.0)
    # default image limits (cv::Point from FLT_MAX saturates to INT_MAX) and the big-frame path
    monkeypatch.setenv("UH_PROJMATCH_NO_LDS", "1")
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"])
    fr2 = dict(fr, min_xy=(0, 0), max_xy=(2 ** 31 - 1, 2 ** 31 - 1))
    got = pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 8.0, 15.0)
    ref = oracle_lib.proj_match_prev(oracle, fr2, mp, pose, 8.0, 15.0)
    assert got["matches"].tobytes() == ref["matches"].tobytes() and len(ref["matches"]) > 10
    np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])


@pytest.mark.gpu
def test_hip_projmatch_discs_wider_than_64_leaves_take_the_serial_walk(hip_ctx, oracle):
    """Round 5 spreads the walk of one point over the 64 lanes of its wave, one leaf per lane; a disc that covers more than 64 leaves (here:
    120 px x scale x 1.6 on 4000 keypoints — hundreds of keypoints per disc) falls back to the serial walk.  Both matchers, same order-
    dependent results as picoflann's recursion."""
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    fr, mp, pose = synth.proj_problem(4000, 1500, 12, low_entropy=True)
    pm = ProjectionMatcher(hip_ctx)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    got = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 3.0, 120.0)
    ref = oracle_lib.proj_match(oracle, fr, mp, pose, 3.0, 120.0)
    np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])
    np.testing.assert_array_equal(got["best_dist"][ref["best_kp"] >= 0], ref["best_dist"][ref["best_kp"] >= 0])
    assert got["matches"].tobytes() == ref["matches"].tobytes()
    got = pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 100.0, 120.0)
    ref = oracle_lib.proj_match_prev(oracle, fr, mp, pose, 100.0, 120.0)
    np.testing.assert_array_equal(got["best_kp"], ref["best_kp"])
    np.testing.assert_array_equal(got["best_dist"], ref["best_dist"])
    assert got["matches"].tobytes() == ref["matches"].tobytes()
