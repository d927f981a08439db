import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(21)
for n in (64, 128, 1024, 2048, 3000, 3500, 3900, 4000, 4032, 4033, 4090, 4095, 4096):
    xy = (rng.random((n, 2)) * [1241, 376]).astype(np.float32)
    for th in (1024, 512, 256):
        a = kdtree_build_dev(ctx, xy, th); b = kdtree_build_host(xy)
        ok = a["nodes"].tobytes() == b["nodes"].tobytes() and a["leaf_idx"].tobytes() == b["leaf_idx"].tobytes()
        msg = ""
        if not ok:
            na, nb = len(a["nodes"]), len(b["nodes"])
            k = min(na, nb)
            d = [i for i in range(k) if a["nodes"][i] != b["nodes"][i]]
            ld = np.nonzero(a["leaf_idx"] != b["leaf_idx"])[0]
            msg = f"nodes {na} vs {nb}, first node diff {d[:3]} {a['nodes'][d[0]] if d else ''} {b['nodes'][d[0]] if d else ''} leafdiff {ld[:5]} depth {a['depth']} {b['depth']}"
        print(n, th, ok, msg)
