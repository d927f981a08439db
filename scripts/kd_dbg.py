import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(21)
sizes = [int(a) for a in sys.argv[1:]] or [5, 11, 12, 20, 40, 64, 128, 500, 1024, 2000, 4096]
for n in sizes:
    xy = (rng.random((n, 2)) * [1241, 376]).astype(np.float32)
    for th in (256, 512):
        print("run", n, th, flush=True)
        a = kdtree_build_dev(ctx, xy, th); b = kdtree_build_host(xy)
        ok = a["nodes"].tobytes() == b["nodes"].tobytes() and a["leaf_idx"].tobytes() == b["leaf_idx"].tobytes()
        print(n, th, ok, flush=True)
