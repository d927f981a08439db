"""Timing of the projection matcher (Map::matchFrameToMapPoints) on the tracking-sized problem: 2000 keypoints, 3000 map points."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.projmatch import ProjectionMatcher, kdtree_build_host
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
for nk, npts in ((2000, 3000), (4000, 10000)):
    fr, mp, pose = synth.proj_problem(nk, npts, 0)
    pm = ProjectionMatcher(ctx)
    args = (fr["und_kpts"], fr["desc"], fr["scale_factors"], fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"], fr["max_xy"])
    margs = (pose, mp["ids"], mp["pos3d"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], 100.0, 15.0)
    for _ in range(3): pm.setFrame(*args); pm.matchFrameToMapPoints(*margs)
    N = 20
    t = time.perf_counter()
    for _ in range(N): pm.setFrame(*args)
    t_set = (time.perf_counter() - t) / N
    xy = np.stack([fr["und_kpts"]["x"], fr["und_kpts"]["y"]], 1)
    t = time.perf_counter()
    for _ in range(N): kdtree_build_host(xy)
    t_tree = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for _ in range(N): r = pm.matchFrameToMapPoints(*margs)
    t_match = (time.perf_counter() - t) / N
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(N): pm.matchFrameToMapPoints(*margs)
    rep = ctx.prof_report(); ctx.prof_enable(False)
    L = oracle_lib.load_oracle()
    t = time.perf_counter()
    for _ in range(3): ro = oracle_lib.proj_match(L, fr, mp, pose, 100.0, 15.0)
    t_cpu = (time.perf_counter() - t) / 3
    assert r["matches"].tobytes() == ro["matches"].tobytes()
    print(f"kpts={nk} pts={npts}: set_frame {t_set*1e3:.3f} ms (host kd build {t_tree*1e3:.3f} ms incl. python) | match call {t_match*1e3:.3f} ms | kernel {rep} | "
          f"matches {len(r['matches'])} | CPU oracle (tree build + match) {t_cpu*1e3:.3f} ms")
