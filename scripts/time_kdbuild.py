"""Phase stamps of the device kd-tree build (UH_KD_CLK=1 prints them) and HIP-event timing of the launch for a few point counts."""
import os
import sys

os.environ.setdefault("UH_KD_CLK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.projmatch import kdtree_build_dev, kdtree_build_host

ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
for n in (2000, 4000, 500):
    xy = (rng.random((n, 2)) * [1203, 338] + 19).astype(np.float32)   # as an extractor leaves them (19 px border): the exact-sum route
    for threads in (0, 512, 256):   # 0: the build over three launches
        for rep in range(3):
            t = kdtree_build_dev(ctx, xy, threads)
        assert t["nodes"].tobytes() == kdtree_build_host(xy)["nodes"].tobytes()
