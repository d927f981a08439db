"""Timing of the hierarchical k-means index (FrameMatcher_Flann's index): host build + GPU search, 2000 queries x 2000/10000 rows."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.knn import Index, kmeans_build_host
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
L = oracle_lib.load_oracle()
R_ = oracle_lib.load_ref("xflann")
for nt in (2000, 10000):
    train, q = synth.match_set(2000, nt, seed=1)
    idx = Index(ctx)
    for _ in range(3): idx.build_kmeans(train, 32, 0)
    t = time.perf_counter()
    for _ in range(10): idx.build_kmeans(train, 32, 0)
    t_build = (time.perf_counter() - t) / 10
    qd = torch.from_numpy(q).cuda()
    for _ in range(3): idx.search_kmeans(qd, 10, 16, False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): idx.search_kmeans(qd, 10, 16, False)
    torch.cuda.synchronize()
    t_search = (time.perf_counter() - t) / 50
    ex = Index(ctx).build(torch.from_numpy(train).cuda())
    for _ in range(3): ex.search(qd, 10)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): ex.search(qd, 10)
    torch.cuda.synchronize()
    t_exact = (time.perf_counter() - t) / 50
    t = time.perf_counter()
    if R_ is not None:
        oracle_lib.ref_hkmeans_search(R_, train, q, 10, 32, 0, 16, 0)
    t_cpu = time.perf_counter() - t
    gi, gd = idx.search_kmeans(qd, 10, 16, False)
    ei, ed = ex.search(qd, 10, sorted=True)
    torch.cuda.synchronize()
    best_km = gd.cpu().numpy().astype(np.int64); best_km[gi.cpu().numpy() < 0] = 10**6
    recall = float((best_km.min(1) == ed.cpu().numpy()[:, 0]).mean())
    print(f"nt={nt}: build (host tree + device distances) {t_build*1e3:.3f} ms | GPU search (nn=10, maxChecks=16) {t_search*1e3:.3f} ms | exact GPU scan {t_exact*1e3:.3f} ms | "
          f"real xflann CPU build+search {t_cpu*1e3:.1f} ms | 1-nn recall of the approximate search {recall:.3f}")
