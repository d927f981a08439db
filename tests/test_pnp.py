"""Pose-only optimisation (PnPSolver::solvePnp): oracle vs real g2o (CPU), HIP vs oracle (gpu)."""
import os

import numpy as np
import pytest

import oracle_lib
import synth

POSE_TOL = 1e-6      # se3 state (unit quaternion + translation, fp64), same stated tolerance as the BA stage


def test_oracle_matches_real_g2o(oracle):
    ref = oracle_lib.load_ref("g2o")
    if ref is None:
        pytest.skip("oracle/_ref/libg2o_ref.so not built (reference tree absent on this box)")
    for n, seed in [(600, 0), (150, 1), (2000, 2), (40, 3), (12, 4), (9, 5)]:
        pr = synth.pnp_problem(n, seed)
        a, b = oracle_lib.pnp_solve(oracle, pr), oracle_lib.pnp_solve_ref(ref, pr)
        assert a["iters"].tolist() == b["iters"].tolist()
        assert a["ngood"] == b["ngood"] and (a["bad"] == b["bad"]).all()
        assert np.abs(a["state"] - b["state"]).max() < 1e-10
        np.testing.assert_array_equal(a["pose"], b["pose"])


def test_oracle_golden_from_real_g2o(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pnp_golden.npz"))
    pr = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    pr["n"] = len(pr["invsig"])
    a = oracle_lib.pnp_solve(oracle, pr)
    assert a["iters"].tolist() == g["ref_iters"].tolist() and a["ngood"] == int(g["ref_ngood"])
    np.testing.assert_array_equal(a["bad"], g["ref_bad"])
    assert np.abs(a["state"] - g["ref_state"]).max() < 1e-10


def test_oracle_recovers_pose_and_outliers(oracle):
    pr = synth.pnp_problem(800, 11, outlier_frac=0.2)
    a = oracle_lib.pnp_solve(oracle, pr)
    gt = pr["pose_gt"]
    assert np.abs(a["pose"].reshape(4, 4)[:3, 3] - gt[:3, 3]).max() < 0.01
    # gross outliers are flagged, nearly all clean matches kept
    assert a["bad"][pr["outlier"]].mean() > 0.9 and a["bad"][~pr["outlier"]].mean() < 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(600, 0), (150, 1), (2000, 2), (40, 3), (12, 4), (9, 5), (3000, 6), (3001, 7), (5000, 8)], ids=lambda c: f"n{c[0]}")
def test_hip_pnp_matches_oracle(hip_ctx, oracle, cfg):
    from ucoslam_cv3_amd.pnp import PnPSolver

    pr = synth.pnp_problem(*cfg)
    sol = PnPSolver(hip_ctx)
    got = sol.solvePnp(pr["pose"], pr["intr"], pr["p3d"], pr["kp"], pr["invsig"], pr["weight"])
    ref = oracle_lib.pnp_solve(oracle, pr)
    assert got["iters"].tolist() == ref["iters"].tolist()
    assert got["ngood"] == ref["ngood"]
    np.testing.assert_array_equal(got["bad"], ref["bad"])
    assert np.abs(got["state"] - ref["state"]).max() < POSE_TOL
    assert np.abs(got["pose"] - ref["pose"]).max() < 1e-5
    again = sol.solvePnp(pr["pose"], pr["intr"], pr["p3d"], pr["kp"], pr["invsig"], pr["weight"])
    np.testing.assert_array_equal(again["state"], got["state"])          # deterministic


@pytest.mark.gpu
def test_hip_pnp_empty(hip_ctx):
    from ucoslam_cv3_amd.pnp import PnPSolver

    pose = np.eye(4, dtype=np.float32).reshape(16)
    got = PnPSolver(hip_ctx).solvePnp(pose, np.ones(4, np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32),
                                      np.zeros(0, np.float32), np.zeros(0, np.float32))
    assert got["ngood"] == 0 and (got["pose"] == pose).all()           # pnpsolver.cpp:149-150: nothing to do
