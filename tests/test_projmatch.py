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


@pytest.mark.parametrize("name", list(_clouds().keys()))
def test_host_kdtree_build_equals_oracle(oracle, name):
    from ucoslam_cv3_amd.projmatch import kdtree_build_host

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
