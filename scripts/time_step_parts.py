"""The bench step with parts removed: local BA alone, + ORB only, + matcher only (1 / 2 / 4 queries per wave), + both."""
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
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
cb = u.Context(0, private=True)
fp = FeatParams(2000, 8, 1.2)
frames = torch.from_numpy(np.stack([synth.frame(1241, 376, seed=f, shift=(2 * f, f)) for f in range(F)])).to(dev)
ext = ORBextractor.create(ctx)
out = ext.extract_batch(frames, fp)
index = Index(ctx).build(torch.from_numpy(synth.match_set(1, 10000, seed=50)[0]).to(dev))
ba = GlobalOptimizer.create(cb)
ba.setParams(synth.ba_problem(10, 3000, seed=0), ParamSet(nIters=5))
ki = torch.empty((F, NQ, NN), dtype=torch.int32, device=dev); kd = torch.empty_like(ki)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n
def orb(): ext.extract_batch(frames, fp, out)
def knn(): check(L.uh_knn_search_dev(index._h, dev_ptr(out[1]), F * NQ, NN, dev_ptr(ki), dev_ptr(kd), 0, -1))
def step(do_orb, do_knn):
    def f():
        ba.optimize_async()
        if do_orb: orb()
        if do_knn: knn()
        ba.wait()
    return f
print(f"BA alone {timed(step(False, False)):.3f} ms | ORB alone {timed(orb):.3f} ms")
print(f"BA + ORB {timed(step(True, False)):.3f} ms")
for q in (1, 2, 4):
    index.set_queries_per_wave(q)
    print(f"qpw {q}: matcher alone {timed(knn):.3f} ms | BA + matcher {timed(step(False, True)):.3f} ms | BA + ORB + matcher {timed(step(True, True)):.3f} ms")
