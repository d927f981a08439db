"""How many independent sessions (frame stream + map + local BA each) one GPU carries: the resident-form step of bench.py's
stages.sessions_on_one_gpu for 1..5 sessions, with the local BAs of sessions beyond the first `npersist` in the launch-chain form
(UH_BA_FORM=legacy while their problem is set) instead of waiting for admission of a third persistent launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd._lib import check, dev_ptr
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
from ucoslam_cv3_amd.knn import Index
from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

W, H, F, NT, NQ, NN = 1241, 376, 4, 10000, 2000, 10
dev = torch.device("cuda", 0)
L = u.lib()
frames = torch.from_numpy(np.stack([synth.frame(W, H, seed=f, shift=(2 * f, f)) for f in range(F)])).to(dev)
map_desc = torch.from_numpy(synth.match_set(1, NT, seed=50)[0]).to(dev)
fp = FeatParams(2000, 8, 1.2)


class Session:
    def __init__(self, i, chain):
        self.ctx_ba, self.ctx_t = u.Context(0, private=True), u.Context(0, private=True)
        if chain:
            os.environ["UH_BA_FORM"] = "legacy"
        self.ba = GlobalOptimizer.create(self.ctx_ba).wantChi2(False)
        self.ba.setParams(synth.ba_problem(10, 3000, seed=100 * i), ParamSet(nIters=5))
        self.ba.optimize()
        os.environ.pop("UH_BA_FORM", None)
        self.ext = ORBextractor.create(self.ctx_t)
        self.idx = Index(self.ctx_t).build(map_desc)
        self.out = self.ext.extract_batch(frames, fp)
        self.knn_idx = torch.empty((F, NQ, NN), dtype=torch.int32, device=dev)
        self.knn_dist = torch.empty_like(self.knn_idx)


def run(nsess, npersist):
    ss = [Session(i, chain=i >= npersist) for i in range(nsess)]

    def step():
        for s in ss: s.ba.optimize_async()
        for s in ss: s.ext.extract_batch(frames, fp, s.out)
        for s in ss: check(L.uh_knn_search_dev(s.idx._h, dev_ptr(s.out[1]), F * NQ, NN, dev_ptr(s.knn_idx), dev_ptr(s.knn_dist), 0, -1))
        for s in ss: s.ba.wait()

    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(15): step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 15 * 1e3)
    ms = float(np.median(ts))
    print(f"sessions {nsess} (persistent {min(nsess, npersist)}, chain {max(0, nsess - npersist)}): step {ms:.4f} ms  {1e3 * nsess * F / ms:.0f} frames/s  forms {[s.ba.form() for s in ss]}", flush=True)


for nsess, npersist in ((1, 2), (2, 2), (3, 2), (4, 2), (3, 1), (4, 1), (2, 0), (4, 0), (3, 3)):
    run(nsess, npersist)
