"""Wall/HIP timing of the BA optimize() on the BASELINE local-BA problem (10 KF x 3000 pts)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
pr = synth.ba_problem(10, 3000, 0)
opt = GlobalOptimizer.create(ctx)
opt.setParams(pr, ParamSet(nIters=5))
for _ in range(3): opt.optimize()
torch.cuda.synchronize()
t = time.perf_counter()
N = 20
for _ in range(N): opt.optimize()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
r = opt.getResults()
print(f"ba optimize: {dt*1e3:.3f} ms per call (E={pr['E']}), iters={r['iters']}")
