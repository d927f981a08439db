"""One small pass of the hot path on cuda:0, checked against the CPU oracle (called by __graft_entry__.smoke())."""
import numpy as np
import torch

import oracle_lib
import synth


def run():
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    torch.cuda.set_device(0)
    ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
    L = oracle_lib.load_oracle()
    # --- Hamming kNN
    train, q = synth.match_set(128, 1000, seed=7)
    index = Index(ctx).build(torch.from_numpy(train).cuda())
    idx, dist = index.search(torch.from_numpy(q).cuda(), 2, sorted=True)
    torch.cuda.synchronize()
    ri, rd = oracle_lib.knn_search(L, train, q, 2, 1)
    assert (idx.cpu().numpy() == ri).all() and (dist.cpu().numpy() == rd).all(), "kNN mismatch vs oracle"
    # ... and against rows recorded from the REAL xflann (tests/golden/knn_golden.npz): the unsorted heap order included
    import os

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "knn_golden.npz"))
    gi, gd = Index(ctx).build(torch.from_numpy(g["ties_train"]).cuda()).search(torch.from_numpy(g["ties_q"]).cuda(), 10, sorted=False)
    assert (gi.cpu().numpy() == g["ties_nn10_s0_idx"]).all() and (gd.cpu().numpy() == g["ties_nn10_s0_dist"]).all(), "kNN rows differ from the real xflann's"
    # --- ORB extractor (small frame, full pipeline) — bit-exact keypoints + descriptors
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    img = synth.frame(320, 240, seed=3)
    ext = ORBextractor.create(ctx)
    kps, desc = ext.detectAndCompute(img, None, FeatParams(600, 4, 1.2))
    rk, rd = oracle_lib.orb_extract(L, img, 600, 4, 1.2)
    assert len(kps) == len(rk) > 100, (len(kps), len(rk))
    for f in ("x", "y", "angle", "response", "octave", "size"):
        assert (kps[f] == rk[f]).all(), f"ORB keypoint field {f} differs from the oracle"
    assert (desc == rd).all(), "ORB descriptors differ from the oracle"
    # --- local BA (small), poses within the stated tolerance
    from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet

    pr = synth.ba_problem(5, 300, seed=4)
    opt = GlobalOptimizer.create(ctx)
    opt.setParams(pr, ParamSet(nIters=5))
    opt.optimize()
    got = opt.getResults()
    ref = oracle_lib.ba_optimize(L, pr, 5)
    assert got["iters"].tolist() == ref["iters"].tolist()
    assert np.abs(got["state"] - ref["state"]).max() < 1e-6, "BA pose state differs from the oracle by more than 1e-6"
    # ... and against the result of the REAL g2o on the committed problem (tests/golden/ba_golden.npz)
    gb = np.load(os.path.join(gold, "ba_golden.npz"))
    gpr = {k[3:]: gb[k] for k in gb.files if k.startswith("in_")}
    opt.setParams(gpr, ParamSet(nIters=5))
    opt.optimize()
    gg = opt.getResults()
    assert gg["iters"].tolist() == gb["ref_iters"].tolist() and np.abs(gg["state"] - gb["ref_state"]).max() < 1e-6 and (gg["bad"] == gb["ref_bad"]).all(), \
        "BA differs from the real g2o's result"
    # --- per-frame pose-only solve
    from ucoslam_cv3_amd.pnp import PnPSolver

    pp = synth.pnp_problem(200, seed=5)
    g = PnPSolver(ctx).solvePnp(pp["pose"], pp["intr"], pp["p3d"], pp["kp"], pp["invsig"], pp["weight"])
    r = oracle_lib.pnp_solve(L, pp)
    assert g["ngood"] == r["ngood"] and (g["bad"] == r["bad"]).all() and np.abs(g["state"] - r["state"]).max() < 1e-6, "PnP differs from the oracle"
    # --- projection matcher: identical matches (kd-tree order dependent)
    from ucoslam_cv3_amd.projmatch import ProjectionMatcher

    fr, mp, pose = synth.proj_problem(400, 500, seed=6)
    pm = ProjectionMatcher(ctx)
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    gm = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    rm = oracle_lib.proj_match(L, fr, mp, pose, 100.0, 15.0)
    assert gm["matches"].tobytes() == rm["matches"].tobytes() and len(rm["matches"]) > 20, "projection matcher differs from the oracle"
    gp = pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], 75.0, 15.0)
    rp = oracle_lib.proj_match_prev(L, fr, mp, pose, 75.0, 15.0)
    assert gp["matches"].tobytes() == rp["matches"].tobytes() and len(rp["matches"]) > 20, "previous-frame projection search differs from the oracle"
