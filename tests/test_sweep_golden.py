"""The oracle AND the HIP product against tests/golden/sweep_golden.npz — outputs of the REAL g2o / xflann / picoflann (compiled from
/root/reference by oracle/Makefile, run by tests/golden/make_sweep_golden.py) at the sizes BASELINE.json names and at the local-BA window
sizes the launch-chain forms serve (17-64 free keyframes).  Inputs are regenerated from tests/synth.py seeds; the fixture holds a SHA-256
of the input bytes, so a drifted generator fails HERE, by name, and not as a parity error.  VERDICT r4 "next round" item 2.

Tolerances: indices / flags / iteration counts exact; se3 state within 1e-9 (oracle, fp64 summation order only) and 1e-6 (HIP: stated BA / PnP
tolerance)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_sweep_golden as G  # noqa: E402  (BA_CASES / KNN_CASES / PNP_CASES, sha(): the generator's own definitions)

POSE_TOL = 1e-6


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "sweep_golden.npz"))


def _ba_key(c):
    return f"ba_{c[0]}x{c[1]}_s{c[2]}_f{c[3]}"


def _ba_problem(gold, c):
    pr = synth.ba_problem(c[0], c[1], c[2], nfixed=c[3])
    np.testing.assert_array_equal(G.ba_input_sha(pr), gold[_ba_key(c) + "_in_sha"], err_msg="tests/synth.py no longer regenerates this fixture's input")
    return pr


def _check_ba(got, gold, key, tol):
    assert got["iters"].tolist() == gold[key + "_iters"].tolist()
    assert np.abs(got["state"] - gold[key + "_state"]).max() < tol
    assert int(got["bad"].sum()) == int(gold[key + "_nbad"])
    np.testing.assert_array_equal(G.sha(got["bad"]), gold[key + "_bad_sha"], err_msg="bad-association flags differ from the real g2o's")


# ------------------------------------------------------------------------------------------------ CPU: the oracle against the fixture
@pytest.mark.parametrize("case", G.BA_CASES, ids=_ba_key)
def test_oracle_ba_equals_real_g2o_at_every_window(oracle, gold, case):
    _check_ba(oracle_lib.ba_optimize(oracle, _ba_problem(gold, case), 5), gold, _ba_key(case), 1e-9)


@pytest.mark.parametrize("case", G.KNN_CASES, ids=lambda c: f"{c[0]}x{c[1]}")
def test_oracle_knn_equals_real_xflann_at_bench_sizes(oracle, gold, case):
    nq, nt, seed = case
    train, q = synth.match_set(nq, nt, seed=seed)
    for s in (0, 1):
        key = f"knn_{nq}x{nt}_nn10_s{s}"
        np.testing.assert_array_equal(G.sha(train, q), gold[key + "_in_sha"])
        idx, dist = oracle_lib.knn_search(oracle, train, q, 10, sorted_=s)
        np.testing.assert_array_equal(idx[:4], gold[key + "_first_rows"])
        np.testing.assert_array_equal(G.sha(idx), gold[key + "_idx_sha"])
        np.testing.assert_array_equal(G.sha(dist), gold[key + "_dist_sha"])


def test_oracle_hkmeans_equals_real_xflann_at_map_size(oracle, gold):
    train, q = synth.match_set(2000, 10000, seed=0)
    np.testing.assert_array_equal(G.sha(train, q), gold["hk_2000x10000_nn10_mc16_in_sha"])
    blob = oracle_lib.hkmeans_blob(oracle, train, 32, 0)
    idx, dist = oracle_lib.hkmeans_search(oracle, blob, q, 10, 16, 0)
    np.testing.assert_array_equal(G.sha(idx), gold["hk_2000x10000_nn10_mc16_idx_sha"])
    np.testing.assert_array_equal(G.sha(dist), gold["hk_2000x10000_nn10_mc16_dist_sha"])


def _kd_frame():
    fr, _, _ = synth.proj_problem(2000, 3000, 0)
    return np.stack([fr["und_kpts"]["x"], fr["und_kpts"]["y"]], 1).astype(np.float32)


def test_oracle_kdtree_equals_real_picoflann_on_a_2000_keypoint_frame(oracle, gold):
    xy = _kd_frame()
    qs, rs = G.kd_queries(xy)
    np.testing.assert_array_equal(G.sha(xy, qs, rs), gold["kd_2000_in_sha"])
    kd = oracle_lib.KdOracle(oracle, "oracle_kd", xy)
    hits, offs = [], [0]
    for (qx, qy), r in zip(qs, rs):
        i, _ = kd.radius(qx, qy, r)
        hits.append(i); offs.append(offs[-1] + len(i))
    np.testing.assert_array_equal(np.array(offs, np.int64), gold["kd_2000_off"])
    np.testing.assert_array_equal(G.sha(np.concatenate(hits).astype(np.uint32)), gold["kd_2000_idx_sha"])


def test_product_kdtree_build_equals_the_pinned_oracle_on_a_2000_keypoint_frame(oracle):
    """uh_kdtree_build_host (what uh_projmatch_set_frame uploads) node for node against the oracle tree the test above pins to picoflann."""
    from ucoslam_cv3_amd.projmatch import kdtree_build_host
    from test_projmatch import _same_tree

    xy = _kd_frame()
    _same_tree(kdtree_build_host(xy), oracle_lib.KdOracle(oracle, "oracle_kd", xy).export())


@pytest.mark.parametrize("case", G.PNP_CASES, ids=lambda c: f"n{c[0]}")
def test_oracle_pnp_equals_real_g2o_at_tracker_sizes(oracle, gold, case):
    n, seed = case
    pr = synth.pnp_problem(n, seed=seed)
    key = f"pnp_{n}_s{seed}"
    np.testing.assert_array_equal(G.pnp_input_sha(pr), gold[key + "_in_sha"])
    r = oracle_lib.pnp_solve(oracle, pr)
    assert r["ngood"] == int(gold[key + "_ngood"]) and r["iters"].tolist() == gold[key + "_iters"].tolist()
    np.testing.assert_array_equal(G.sha(r["bad"]), gold[key + "_bad_sha"])
    assert np.abs(r["state"] - gold[key + "_state"]).max() < 1e-9


# ------------------------------------------------------------------------------------------------ GPU: the HIP forms against the fixture
_FORMS = {  # window -> (expected form, environment variants that must all reproduce the real g2o)
    10: ("persist8", [{}, {"UH_BA_FORM": "legacy"}, {"UH_BA_NF": "16"}]),
    18: ("persist16", [{}, {"UH_BA_FORM": "legacy"}]),          # 16 free keyframes: the widest persistent window (two rows per lane); the launch chain
    19: ("chain", [{}, {"UH_BA_PREBUILT": "0"}, {"UH_BA_SCHUR_DENSE": "0"}]),   # 17 free: fused row-per-lane solve, dense / pair Schur form
    26: ("chain", [{}, {"UH_BA_SCHUR_DENSE": "0"}]),            # packed MFMA solve
    34: ("chain", [{}, {"UH_BA_SOLVE": "hbm"}]),                # 32 free keyframes: packed solve; forced HBM solve
    50: ("chain", [{}, {"UH_BA_SCHUR_DENSE": "0"}]),            # 48 free: wide dense kernel + HBM solve
    66: ("chain", [{}]),                                        # 64 free
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.BA_CASES, ids=_ba_key)
def test_hip_ba_equals_real_g2o_at_every_window(hip_ctx, gold, case, monkeypatch):
    from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet

    pr = _ba_problem(gold, case)
    form, envs = _FORMS[case[0]]
    for env in envs:
        for k in ("UH_BA_FORM", "UH_BA_NF", "UH_BA_SCHUR_DENSE", "UH_BA_PREBUILT", "UH_BA_SOLVE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        opt = GlobalOptimizer.create(hip_ctx)
        opt.setParams(pr, ParamSet(nIters=5))
        if not env:
            assert opt.form() == form
        opt.optimize()
        _check_ba(opt.getResults(), gold, _ba_key(case), POSE_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.KNN_CASES, ids=lambda c: f"{c[0]}x{c[1]}")
def test_hip_knn_equals_real_xflann_at_bench_sizes(hip_ctx, gold, case):
    """2000 x 10 000 runs the fused scan + heap kernel, 8000 x 10 000 (the headline's launch) the stream form; host-pointer entry too."""
    import torch
    from ucoslam_cv3_amd.knn import Index

    nq, nt, seed = case
    train, q = synth.match_set(nq, nt, seed=seed)
    index = Index(hip_ctx).build(torch.from_numpy(train).cuda())
    for s in (0, 1):
        key = f"knn_{nq}x{nt}_nn10_s{s}"
        idx, dist = index.search(torch.from_numpy(q).cuda(), 10, sorted=bool(s))
        np.testing.assert_array_equal(G.sha(idx.cpu().numpy()), gold[key + "_idx_sha"])
        np.testing.assert_array_equal(G.sha(dist.cpu().numpy()), gold[key + "_dist_sha"])
    idx, dist = Index(hip_ctx).build(train).search(q, 10, sorted=False)
    np.testing.assert_array_equal(G.sha(idx), gold[f"knn_{nq}x{nt}_nn10_s0_idx_sha"])


@pytest.mark.gpu
def test_hip_hkmeans_equals_real_xflann_at_map_size(hip_ctx, gold):
    import torch
    from ucoslam_cv3_amd.knn import Index

    train, q = synth.match_set(2000, 10000, seed=0)
    index = Index(hip_ctx).build_kmeans(train, 32, 0)
    idx, dist = index.search_kmeans(torch.from_numpy(q).cuda(), 10, 16, sorted=False)
    np.testing.assert_array_equal(G.sha(idx.cpu().numpy()), gold["hk_2000x10000_nn10_mc16_idx_sha"])
    np.testing.assert_array_equal(G.sha(dist.cpu().numpy()), gold["hk_2000x10000_nn10_mc16_dist_sha"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", G.PNP_CASES, ids=lambda c: f"n{c[0]}")
def test_hip_pnp_equals_real_g2o_at_tracker_sizes(hip_ctx, gold, case):
    from ucoslam_cv3_amd.pnp import PnPSolver

    n, seed = case
    pr = synth.pnp_problem(n, seed=seed)
    key = f"pnp_{n}_s{seed}"
    r = PnPSolver(hip_ctx).solvePnp(pr["pose"], pr["intr"], pr["p3d"], pr["kp"], pr["invsig"], pr["weight"])
    assert r["ngood"] == int(gold[key + "_ngood"]) and r["iters"].tolist() == gold[key + "_iters"].tolist()
    np.testing.assert_array_equal(G.sha(r["bad"]), gold[key + "_bad_sha"])
    assert np.abs(r["state"] - gold[key + "_state"]).max() < POSE_TOL
