"""Randomised parity sweep on the GPU: every stage against the CPU oracle on many seeds / shapes (not part of the test suite;
the summary is quoted in DESIGN.md).  Usage: python scripts/fuzz_parity.py [seconds_per_stage]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth, oracle_lib
import ucoslam_cv3_amd as u

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
L = oracle_lib.load_oracle()
rng = np.random.default_rng(12345)
report = {}


def run(name, fn):
    if only and name not in only:
        return
    t0, n, bad = time.time(), 0, []
    replay = [int(x) for x in os.environ.get("UH_FUZZ_SEEDS", "").split(",") if x]   # re-run given seeds of the selected stage (A/B of one path)
    while (n < len(replay)) if replay else (time.time() - t0 < budget):
        seed = replay[n] if replay else int(rng.integers(0, 2 ** 31))
        try:
            ok, info = fn(seed)
        except Exception as e:   # a crash is a failure too
            ok, info = False, f"exception {e!r}"
        n += 1
        if not ok:
            bad.append((seed, info))
    report[name] = (n, bad)
    print(f"{name}: {n} cases, {len(bad)} mismatches", flush=True)
    for b in bad[:40]:
        print("   ", b, flush=True)


# ---- ORB: random sizes / feature budgets / levels
from ucoslam_cv3_amd.orb import FeatParams, ORBextractor
ext = ORBextractor.create(ctx)

def orb_case(seed):
    r = np.random.default_rng(seed)
    w, h = int(r.integers(96, 900)), int(r.integers(80, 500))
    nf, nl = int(r.integers(50, 3000)), int(r.integers(1, 9))
    sf = float(r.choice([1.2, 1.1, 1.5, 2.0]))
    img = synth.frame(w, h, seed=seed % 100000)
    if r.random() < 0.2:
        img = (img.astype(np.int32) // 8 * 8).astype(np.uint8)     # plateaus -> ties in the FAST scores
    try:
        kps, desc = ext.detectAndCompute(img, None, FeatParams(nf, nl, sf))
    except u.UcoslamHipError as e:
        # geometry the reference itself rejects (cv::Exception from an out-of-image cell ROI): the oracle must reject it too
        try:
            oracle_lib.orb_extract(L, img, nf, nl, sf)
        except Exception:
            return True, None
        return False, f"product raised ({e}) but the oracle did not: {(w, h, nf, nl, sf)}"
    rk, rd = oracle_lib.orb_extract(L, img, nf, nl, sf)
    ok = len(kps) == len(rk) and all((kps[f] == rk[f]).all() for f in ("x", "y", "angle", "response", "octave", "size")) and (desc == rd).all()
    return ok, (w, h, nf, nl, sf, len(kps), len(rk))

run("orb", orb_case)

# ---- kNN (exact) and k-means index
from ucoslam_cv3_amd.knn import Index

def knn_case(seed):
    r = np.random.default_rng(seed)
    nt, nq, nn = int(r.integers(1, 6000)), int(r.integers(1, 400)), int(r.integers(1, 65))
    if r.random() < 0.3:
        train, q = synth.tie_stress_set(nq, nt, seed=seed % 1000, ndistinct=int(r.integers(1, 40)))
    else:
        train, q = synth.match_set(nq, nt, seed=seed % 1000)
    srt = bool(r.integers(0, 2))
    qpw = int(r.choice([1, 2, 4]))
    idx = Index(ctx).build(train).set_queries_per_wave(qpw)
    gi, gd = idx.search(q, nn, sorted=srt)
    ri, rd = oracle_lib.knn_search(L, train, q, nn, int(srt))
    return (gi == ri).all() and (gd == rd).all(), (nt, nq, nn, srt, qpw)

run("knn_exact", knn_case)


def knn2_case(seed, form="twophase"):   # the accept-list scan + lane replay (default only from 3000 queries on), forced for every size
    r = np.random.default_rng(seed)
    nt, nq, nn = int(r.integers(1, 6000)), int(r.integers(1, 400)), int(r.integers(1, 17))
    if r.random() < 0.3:
        train, q = synth.tie_stress_set(nq, nt, seed=seed % 1000, ndistinct=int(r.integers(1, 40)))
    else:
        train, q = synth.match_set(nq, nt, seed=seed % 1000)
    srt = bool(r.integers(0, 2))
    md = int(r.choice([-1, -1, 40, 90, 120]))
    os.environ["UH_KNN_FORM"] = form
    os.environ["UH_KNN_ACCEPT_QPW"] = str(int(r.choice([1, 2]))) if form == "twophase" else "2"
    try:
        idx = Index(ctx).build(train)
    finally:
        del os.environ["UH_KNN_FORM"], os.environ["UH_KNN_ACCEPT_QPW"]
    gi, gd = idx.search(q, nn, sorted=srt, max_dist=md)
    ri, rd = oracle_lib.knn_search(L, train, q, nn, int(srt), max_dist=md)
    return (gi == ri).all() and (gd == rd).all(), (nt, nq, nn, srt, md)

run("knn_twophase", knn2_case)


def knn3_case(seed):   # scan and replay in one launch (default from 3000 queries on, nn >= 6); every third case large enough for many workgroups
    if seed % 3:
        return knn2_case(seed, "stream")
    r = np.random.default_rng(seed)
    nt, nq, nn = int(r.integers(500, 4000)), int(r.integers(3000, 9000)), int(r.integers(1, 17))
    train, q = synth.match_set(nq, nt, seed=seed % 1000)
    srt = bool(r.integers(0, 2))
    os.environ["UH_KNN_FORM"] = "stream"
    try:
        idx = Index(ctx).build(train)
    finally:
        del os.environ["UH_KNN_FORM"]
    gi, gd = idx.search(q, nn, sorted=srt)
    ri, rd = oracle_lib.knn_search(L, train, q, nn, int(srt))
    return (gi == ri).all() and (gd == rd).all(), (nt, nq, nn, srt)

run("knn_stream", knn3_case)

def km_case(seed):
    r = np.random.default_rng(seed)
    nt, nq = int(r.integers(1, 5000)), int(r.integers(1, 300))
    k = int(r.choice([32, 32, 8, 16, 64, 3]))
    nn, mc = int(r.integers(3, 33)), int(r.choice([16, 16, 1, 5, 64, 300]))
    if r.random() < 0.3:
        train = np.zeros((nt, 32), np.uint8); train[:, :2] = r.integers(0, 256, (nt, 2)); q = train[r.integers(0, nt, nq)].copy()
    else:
        train, q = synth.match_set(nq, nt, seed=seed % 1000)
    mi = int(r.choice([0, 0, 0, 1, 11, -1]))
    blob = oracle_lib.hkmeans_blob(L, train, k, mi)
    idx = Index(ctx)
    if isinstance(blob, int):
        try:
            idx.build_kmeans(train, k, mi)
            return False, "expected an error"
        except u.UcoslamHipError:
            return True, None
    idx.build_kmeans(train, k, mi)
    srt = bool(r.integers(0, 2))
    gi, gd = idx.search_kmeans(q, nn, mc, srt)
    ri, rd = oracle_lib.hkmeans_search(L, blob, q, nn, mc, int(srt))
    return idx.kmeans_blob().tobytes() == blob.tobytes() and (gi == ri).all() and (gd == rd).all(), (nt, nq, k, mi, nn, mc, srt)

run("kmeans_index", km_case)

# ---- BoW frame matcher (GPU distances + order-dependent bookkeeping, shared host tail) vs the pure-Python restatement
import py_oracle_matcher as pyo
from ucoslam_cv3_amd import matcher as MM
bowm = MM.FrameMatcherBoW(ctx)

def bowm_case(seed):
    r = np.random.default_rng(seed)
    nq, nt = int(r.integers(1, 300)), int(r.integers(1, 400))
    le = bool(r.random() < 0.5)
    train, q = synth.match_set(nq, nt, seed=seed % 100000)
    if le:
        train[:, 1:] = 0; q[:, 1:] = 0
    def fr(n, desc):
        return dict(desc=desc, ids=np.where(r.random(n) < 0.5, r.integers(0, 1000, n), 0xFFFFFFFF).astype(np.uint32), nonmaxima=r.random(n) < 0.05,
                    octave=r.integers(0, 8, n).astype(np.int32), angle=(r.random(n) * 360).astype(np.float32), pt=(r.random((n, 2)) * 600).astype(np.float32),
                    scaleFactors=(1.2 ** np.arange(8)).astype(np.float32))
    tf, qf = fr(nt, train), fr(nq, q)
    nn = int(r.integers(1, 40))
    for f, n in ((tf, nt), (qf, nq)):
        bv = {}
        for i, nd in enumerate(r.integers(0, nn, n).tolist()):
            bv.setdefault(int(nd) * 3, []).append(i)
        f["bowvector_level"] = bv
    tm, qm = int(r.integers(0, 3)), int(r.integers(0, 3))
    md, ratio, co, mod = float(r.choice([3.0, 6.0] if le else [60.0, 100.0, 3e38])), float(r.choice([0.6, 0.8, 0.9])), bool(r.random() < 0.5), int(r.integers(0, 4))
    F = (r.normal(0, 1, 9) * [1e-6, 1e-5, 1e-3, 1e-5, 1e-6, 1e-3, 1e-3, 1e-3, 1]).astype(np.float32) if r.random() < 0.4 else None
    bowm.setParams(tf, tm, md, ratio, co, mod)
    g = bowm.matchEpipolar(qf, qm, F)
    o = pyo.bow_match(qf, tf, MM.is_used(qf, qm), MM.is_used(tf, tm), md, ratio, co, mod, F12=F)
    return [(int(m["queryIdx"]), int(m["trainIdx"]), float(m["distance"])) for m in g] == [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in o], (nq, nt, le, tm, qm, md, ratio, co, mod, F is not None)

run("bow_matcher", bowm_case)

# ---- projection matcher
from ucoslam_cv3_amd.projmatch import ProjectionMatcher
pm = ProjectionMatcher(ctx)

def pm_case(seed):
    r = np.random.default_rng(seed)
    nk, npt = int(r.integers(0, 4500)), int(r.integers(1, 6000))
    le = bool(r.random() < 0.4)
    fr, mp, pose = synth.proj_problem(nk, npt, seed % 100000, low_entropy=le, n_levels=int(r.integers(2, 9)), pose_noise=float(r.choice([0.0, 0.002, 0.02])))
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    md, mr = (8.0 if le else float(r.choice([100.0, 50.0, 30.0]))), float(r.choice([15.0, 2.5, 40.0]))
    g = pm.matchFrameToMapPoints(pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], md, mr)
    o = oracle_lib.proj_match(L, fr, mp, pose, md, mr)
    return g["matches"].tobytes() == o["matches"].tobytes() and (g["visible"] == o["visible"]).all() and (g["best_kp"] == o["best_kp"]).all(), (nk, npt, le, md, mr)

run("projmatch", pm_case)

def pmprev_case(seed):   # the tracker's search against the previous frame (uh_projmatch_match_prev)
    r = np.random.default_rng(seed)
    nk, npt = int(r.integers(0, 4500)), int(r.integers(1, 6000))
    le = bool(r.random() < 0.4)
    fr, mp, pose = synth.proj_problem(nk, npt, seed % 100000, low_entropy=le, n_levels=int(r.integers(2, 9)), pose_noise=float(r.choice([0.0, 0.002, 0.02])))
    pm.setFrame(fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    md, mr = (float(r.choice([0.0, 3.0, 8.0])) if le else float(r.choice([100.0, 75.0, 30.0]))), float(r.choice([7.5, 15.0, 2.5, 40.0]))
    g = pm.matchFrameToPrevFrame(pose, mp["ids"], mp["pos3d"], mp["octave"], mp["desc"], md, mr)
    o = oracle_lib.proj_match_prev(L, fr, mp, pose, md, mr)
    return g["matches"].tobytes() == o["matches"].tobytes() and (g["best_kp"] == o["best_kp"]).all() and (g["best_dist"] == o["best_dist"]).all(), (nk, npt, le, md, mr)

run("projmatch_prev", pmprev_case)

# ---- Frame::create_kdtree on the device (csrc/kdbuild.hpp) against the host builder (pinned to the real picoflann by the goldens)
from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host

def kd_case(seed):
    r = np.random.default_rng(seed)
    n = int(r.integers(0, 4097)) if r.random() < 0.7 else int(r.integers(0, 200))
    mode = int(r.integers(0, 8))
    if mode == 0:
        xy = (r.random((n, 2)) * [1241, 376]).astype(np.float32)
    elif mode == 6:   # inside an extractor's 19 px border: the builders' exact-sum route
        xy = (r.random((n, 2)) * [1203, 338] + 19).astype(np.float32)
    elif mode == 7:   # the same on integer pixels (ties in both coordinates)
        xy = np.stack([r.integers(19, 1222, n), r.integers(19, 357, n)], 1).astype(np.float32)
    elif mode == 1:
        xy = r.integers(0, max(2, int(r.integers(2, 200))), (n, 2)).astype(np.float32)
    elif mode == 2:   # extractor-like: pixel centres of a level scaled back to level 0
        xy = (r.integers(0, 400, (n, 2)).astype(np.float32) + np.float32(0.5)) * np.float32(1.2) ** r.integers(0, 8, (n, 1)).astype(np.float32)
    elif mode == 3:
        xy = r.normal([600, 180], [r.random() * 80 + 0.01, r.random() * 8 + 0.01], (n, 2)).astype(np.float32)
    elif mode == 4:
        xy = np.stack([r.random(n).astype(np.float32) * 1e-3, r.integers(0, 3, n).astype(np.float32)], 1)
    else:
        xy = (r.normal(0, 1, (n, 2)) * [3e4, 1e-2]).astype(np.float32)
    a, b = kdtree_build_dev(ctx, xy, int(r.choice([0, 256, 512]))), kdtree_build_host(xy)
    ok = a["nodes"].tobytes() == b["nodes"].tobytes() and a["leaf_idx"].tobytes() == b["leaf_idx"].tobytes() and a["root_box"].tobytes() == b["root_box"].tobytes() and a["depth"] == b["depth"]
    return ok, None if ok else (n, mode)

run("kdtree_device", kd_case)

# ---- uh_track_pose (csrc/track.hpp): the tracker's pose estimation as one call against the four operators in sequence (bit for bit)
from ucoslam_cv3_amd.orb import DeviceFrame
from ucoslam_cv3_amd.pnp import PnPSolver as _TrkPnP
from ucoslam_cv3_amd.projmatch import DMATCH_DTYPE as _DM
_trk_pnp = _TrkPnP(ctx)
_trk_frames = [DeviceFrame(ctx), DeviceFrame(ctx).setTreeBuilder(True)]

def track_case(seed):
    from ucoslam_cv3_amd._lib import lib, np_ptr
    r = np.random.default_rng(seed)
    nk, npt = int(r.integers(12, 4000)), int(r.integers(40, 5000))
    le = bool(r.random() < 0.2)
    fr, mp, pose = synth.proj_problem(nk, npt, seed % 100000, low_entropy=le, n_levels=int(r.integers(2, 9)), pose_noise=float(r.choice([0.0, 0.002, 0.02, 0.5])))
    dfr = _trk_frames[int(r.integers(0, 2))]
    dfr.upload(fr["und_kpts"], fr["desc"])
    pm.setFrameDev(dfr, fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"], und_kpts=fr["und_kpts"])
    # previous-frame items: a subset of the map points (some with ids outside the local map and a slightly different position)
    n_prev = int(r.integers(0, min(npt, 1500)))
    rows = np.sort(r.choice(npt, n_prev, replace=False)) if n_prev else np.zeros(0, np.int64)
    in_map = r.random(n_prev) < 0.7
    prev = dict(ids=np.where(in_map, mp["ids"][rows], 10 ** 6 + np.arange(n_prev)).astype(np.uint32), pos3d=(mp["pos3d"][rows] + r.normal(0, 0.002, (n_prev, 3))).astype(np.float32),
                octave=mp["octave"][rows].astype(np.int32), desc=np.ascontiguousarray(mp["desc"][rows]))
    prev_row = np.where(in_map, rows, -1).astype(np.int32)
    weight = np.where(r.random(npt) < 0.2, np.float32(0.5), np.float32(1.0)).astype(np.float32)
    intr = np.array([fr["fx"], fr["fy"], fr["cx"], fr["cy"]], np.float32)
    inv_sf = (np.float32(1) / fr["scale_factors"]).astype(np.float32)
    d1, r1 = (float(r.choice([3.0, 8.0])) if le else float(r.choice([100.0, 75.0]))), float(r.choice([7.5, 15.0, 40.0]))
    d2, rt, rl = (8.0 if le else float(r.choice([100.0, 50.0]))), float(r.choice([4.0, 2.5])), float(r.choice([15.0, 40.0]))
    mi = int(r.choice([30, 30, 5, 100000]))
    ukp = fr["und_kpts"]
    # the four operators
    a = pm.matchFrameToPrevFrame(pose, prev["ids"], prev["pos3d"], prev["octave"], prev["desc"], d1, r1)
    m1 = a["matches"]
    pid = {int(v): i for i, v in enumerate(prev["ids"])}
    it1 = np.array([pid[int(t)] for t in m1["trainIdx"]], np.int64)
    q1 = m1["queryIdx"]
    s1 = _trk_pnp.solvePnp(pose, intr, prev["pos3d"][it1].reshape(-1, 3), np.stack([ukp["x"][q1], ukp["y"][q1]], 1).reshape(-1, 2), inv_sf[ukp["octave"][q1]], np.ones(len(m1), np.float32))
    tracked = s1["ngood"] >= mi
    pose_map = s1["pose"] if tracked else pose
    b = pm.matchFrameToMapPoints(pose_map, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], d2, rt if tracked else rl)
    m2 = b["matches"]
    un = np.ascontiguousarray(np.concatenate([m1[s1["bad"][: len(m1)] == 0] if tracked else m1[:0], m2]).astype(_DM))
    if len(un):
        un = un[: lib().uh_filter_ambiguous(np_ptr(un), len(un), 0)]
    mrow = {int(v): i for i, v in enumerate(mp["ids"])}
    p3d = np.zeros((len(un), 3), np.float32); w = np.ones(len(un), np.float32)
    for i, tr in enumerate(un["trainIdx"]):
        row = mrow.get(int(tr), -1)
        if row >= 0: p3d[i] = mp["pos3d"][row]; w[i] = weight[row]
        else: p3d[i] = prev["pos3d"][pid[int(tr)]]
    qa = un["queryIdx"]
    s2 = _trk_pnp.solvePnp(pose_map, intr, p3d, np.stack([ukp["x"][qa], ukp["y"][qa]], 1).reshape(-1, 2), inv_sf[ukp["octave"][qa]], w)
    # the one call
    f = pm.trackPose(_trk_pnp, pose, intr, inv_sf, prev, mp, prev_map_row=prev_row, map_weight=weight, prev_min_desc_dist=d1, prev_max_repj_dist=r1, map_min_desc_dist=d2,
                     map_radius_tracked=rt, map_radius_lost=rl, min_inliers=mi)
    ok = (f["matches_prev"].tobytes() == m1.tobytes() and f["matches_map"].tobytes() == m2.tobytes() and f["matches_all"].tobytes() == un.tobytes()
          and (f["bad_prev"] == s1["bad"][: len(m1)]).all() and (f["bad_all"] == s2["bad"][: len(un)]).all() and f["tracked"] == bool(tracked)
          and f["inliers1"] == s1["ngood"] and f["inliers2"] == s2["ngood"] and (f["iters1"] == s1["iters"]).all() and (f["iters2"] == s2["iters"]).all()
          and (len(m1) == 0 or f["pose1"].tobytes() == np.asarray(s1["pose"], np.float32).tobytes()) and f["pose2"].tobytes() == np.asarray(s2["pose"], np.float32).tobytes())
    return ok, None if ok else (nk, npt, n_prev, le, mi, len(m1), len(m2), len(un))

run("track_pose", track_case)

# ---- BA and PnP (tolerance 1e-6 on the se3 state, identical iteration counts / flags)
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
from ucoslam_cv3_amd.pnp import PnPSolver

_ba_stream = GlobalOptimizer.create(ctx)   # ONE object takes every other problem: a keyframe stream (staging blocks regrown, table cells reused)

def ba_case2(seed):
    r = np.random.default_rng(seed)
    K, P, nfix = int(r.integers(3, 37)), int(r.integers(40, 1500)), int(r.integers(1, 3))   # 1..35 free keyframes: both persistent instantiations, the launch chain with the fused, the packed (dense Schur form: 17-32) and the HBM solve
    nit = int(r.choice([5, 10]))
    hard = r.random() < 0.35   # a third of the cases: rejected trials, lambda factors other than 1/3, passes that end early (the speculative trial's drop path)
    if hard:
        pr = synth.ba_problem(K, min(P, 400), seed % 100000, nfixed=min(nfix, K - 1), outlier_frac=0.3, pix_noise=3.0,
                              pose_noise=float(r.choice([0.1, 0.3, 0.6, 1.0])), point_noise=float(r.choice([0.5, 2.0, 5.0, 10.0])))
    else:
        pr = synth.ba_problem(K, P, seed % 100000, nfixed=min(nfix, K - 1), outlier_frac=float(r.choice([0.0, 0.02, 0.1])), pose_noise=float(r.choice([0.005, 0.01, 0.03])))
    if r.random() < 0.25: os.environ["UH_BA_NF"] = "16"   # the 16-lane instantiation also on windows of up to 8 free keyframes
    else: os.environ.pop("UH_BA_NF", None)
    opt = _ba_stream if r.random() < 0.5 else GlobalOptimizer.create(ctx)
    staged = r.random() < 0.3
    if staged:
        dims = opt.fillStaging(pr)
        opt.setParamsStaged(*dims, ParamSet(nIters=nit))
    else:
        opt.setParams(pr, ParamSet(nIters=nit))
    form = opt.form()
    opt.optimize()
    g = opt.getResults()
    os.environ.pop("UH_BA_NF", None)
    o = oracle_lib.ba_optimize(L, pr, nit)
    err = float(np.abs(g["state"] - o["state"]).max())
    ok = g["iters"].tolist() == o["iters"].tolist() and err < 1e-6 and (g["bad"] == o["bad"]).mean() > 0.999
    if not ok and hard and _g2o is not None:
        # Ill-conditioned on purpose: where the oracle and the REAL g2o (oracle/_ref, same algorithm, another summation order) disagree with
        # each other by more than the tolerance, the case measures round-off amplification, not an implementation.  Counted separately.
        ref = oracle_lib.ba_optimize_ref(_g2o, pr, nit)
        spread = float(np.abs(ref["state"] - o["state"]).max())
        if spread > 1e-7 or ref["iters"].tolist() != o["iters"].tolist():
            _chaotic.append((seed, err, spread))
            # the GPU has to land on ONE of the two trajectories: the oracle's (within the amplified tolerance) or the real g2o's —
            # iteration counts, state and bad flags; "the two references disagree" alone waives nothing
            tol = 1e3 * max(spread, 1e-9)
            near_o = g["iters"].tolist() == o["iters"].tolist() and err <= tol
            err_r = float(np.abs(g["state"] - ref["state"]).max())
            near_r = g["iters"].tolist() == ref["iters"].tolist() and err_r <= tol and (g["bad"] == ref["bad"]).mean() > 0.99
            ok = near_o or near_r
            if ok and not near_o:
                _waived.append(seed)
    return ok, (K, P, nfix, nit, form, staged, "hard" if hard else "", g["iters"].tolist(), o["iters"].tolist(), err)

_g2o = oracle_lib.load_ref("g2o")
_chaotic = []
_waived = []   # hard cases accepted because the GPU follows the real g2o's trajectory where the oracle's differs
run("ba", ba_case2)
if "ba" in report and len(_waived) > max(3, report["ba"][0] // 50):   # (a handful is round-off amplification; a fifth of the hard cases is a bug)
    print(f"ba: FAIL — {len(_waived)} hard cases follow the real g2o but not the oracle (seeds {_waived[:10]}): more than round-off amplification explains", flush=True)
    report["ba"][1].append(("waived", _waived[:20]))
if _chaotic:
    print(f"ba: {len(_chaotic)} hard cases on which the oracle and the real g2o disagree beyond 1e-7 themselves (round-off amplification; worst GPU-oracle {max(c[1] for c in _chaotic):.1e}, worst g2o-oracle {max(c[2] for c in _chaotic):.1e})", flush=True)

def ba_wide_case(seed):   # more than 64 free keyframes: the wide form (sparse pair lists, blocked dense LDL^T in HBM)
    r = np.random.default_rng(seed)
    K, P, nfix = int(r.integers(67, 120)), int(r.integers(200, 1200)), int(r.integers(1, 3))
    nit = int(r.choice([5, 10]))
    pr = synth.ba_problem(K, P, seed % 100000, nfixed=nfix, outlier_frac=float(r.choice([0.0, 0.02, 0.1])), pose_noise=float(r.choice([0.005, 0.01, 0.03])))
    opt = GlobalOptimizer.create(ctx)
    opt.setParams(pr, ParamSet(nIters=nit))
    opt.optimize()
    g = opt.getResults()
    o = oracle_lib.ba_optimize(L, pr, nit)
    err = float(np.abs(g["state"] - o["state"]).max())
    ok = g["iters"].tolist() == o["iters"].tolist() and err < 1e-6 and (g["bad"] == o["bad"]).mean() > 0.999
    return ok, (K, P, nfix, nit, g["iters"].tolist(), o["iters"].tolist(), err)

run("ba_wide", ba_wide_case)
pnp = PnPSolver(ctx)

def pnp_case(seed):
    r = np.random.default_rng(seed)
    n = int(r.integers(8, 3500))
    pr = synth.pnp_problem(n, seed % 100000, outlier_frac=float(r.choice([0.0, 0.15, 0.4])), pose_noise=float(r.choice([0.01, 0.03, 0.08])))
    g = pnp.solvePnp(pr["pose"], pr["intr"], pr["p3d"], pr["kp"], pr["invsig"], pr["weight"])
    o = oracle_lib.pnp_solve(L, pr)
    err = float(np.abs(g["state"] - o["state"]).max())
    return g["ngood"] == o["ngood"] and (g["bad"] == o["bad"]).all() and g["iters"].tolist() == o["iters"].tolist() and err < 1e-6, (n, g["ngood"], o["ngood"], err)

run("pnp", pnp_case)
tot = sum(len(b) for _, b in report.values())
print("FUZZ SUMMARY:", {k: (n, len(b)) for k, (n, b) in report.items()}, "TOTAL MISMATCHES", tot)
