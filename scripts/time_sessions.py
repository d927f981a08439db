"""Aggregate tracking throughput of S independent sessions (frame stream + map + local BA each) on ONE GPU: every session has its own
tracking context (private stream) and its own BA context; one step = every session extracts 4 frames, matches them and runs one local BA."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd._lib import check, dev_ptr
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
from ucoslam_cv3_amd.knn import Index
from ucoslam_cv3_amd.orb import FeatParams, ORBextractor
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
F, NQ, NN = 4, 2000, 10
L = u.lib()
fp = FeatParams(2000, 8, 1.2)
frames = torch.from_numpy(np.stack([synth.frame(1241, 376, seed=f, shift=(2 * f, f)) for f in range(F)])).to(dev)
map_desc = torch.from_numpy(synth.match_set(1, 10000, seed=50)[0]).to(dev)

class Session:
    def __init__(self, i):
        self.ct, self.cb = u.Context(0, private=True), u.Context(0, private=True)
        self.ext = ORBextractor.create(self.ct)
        self.out = self.ext.extract_batch(frames, fp)
        self.index = Index(self.ct).build(map_desc)
        self.ba = GlobalOptimizer.create(self.cb)
        self.ba.setParams(synth.ba_problem(10, 3000, seed=i), ParamSet(nIters=5))
        self.ki = torch.empty((F, NQ, NN), dtype=torch.int32, device=dev); self.kd = torch.empty_like(self.ki)
    def issue(self):
        self.ba.optimize_async()
    def track(self):
        self.ext.extract_batch(frames, fp, self.out)
        check(L.uh_knn_search_dev(self.index._h, dev_ptr(self.out[1]), F * NQ, NN, dev_ptr(self.ki), dev_ptr(self.kd), 0, -1))

for S in (1, 2, 3, 4, 6, 8):
    ss = [Session(i) for i in range(S)]
    def step():
        for s in ss: s.issue()
        for s in ss: s.track()
        for s in ss: s.ba.wait()
    for _ in range(5): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 30
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{S} sessions: {1e3*dt:.3f} ms per step, {S*F/dt:.0f} frames/s aggregate", flush=True)
    del ss
