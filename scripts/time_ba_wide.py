"""Global-BA-sized problems through the wide form (more than 64 free keyframes): time per optimize() next to the real g2o
(oracle/_ref) or the oracle port on one host core."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx = u.Context(0, private=True)
L = oracle_lib.load_oracle()
g2o = oracle_lib.load_ref("g2o")
for K, P in ((100, 5000), (300, 20000), (600, 40000)):
    pr = synth.ba_problem(K, P, seed=1, nfixed=2)
    opt = GlobalOptimizer.create(ctx)
    t = time.perf_counter(); opt.setParams(pr, ParamSet(nIters=10)); t_set = time.perf_counter() - t
    opt.optimize()
    t = time.perf_counter(); opt.optimize(); dt = time.perf_counter() - t
    r = opt.getResults()
    line = f"K={K} P={P} E={pr['E']}: setParams {1e3*t_set:.1f} ms, optimize {1e3*dt:.1f} ms, iters {r['iters'].tolist()}"
    if K <= 300:
        t = time.perf_counter()
        ref = (oracle_lib.ba_optimize_ref(g2o, pr, 10) if g2o is not None else oracle_lib.ba_optimize(L, pr, 10))
        tc = time.perf_counter() - t
        line += f" | {'real g2o' if g2o is not None else 'oracle port'} {1e3*tc:.0f} ms, iters {ref['iters'].tolist()}, max |state diff| {np.abs(r['state'] - ref['state']).max():.2e}"
    print(line, flush=True)

# per-kernel split of one optimize() of the middle problem (HIP events around every launch)
pr = synth.ba_problem(300, 20000, seed=1, nfixed=2)
opt = GlobalOptimizer.create(ctx)
opt.setParams(pr, ParamSet(nIters=10))
opt.optimize()
ctx.prof_enable(True); ctx.prof_reset()
opt.optimize()
rep = ctx.prof_report(); ctx.prof_enable(False)
tot = sum(v[1] for v in rep.values())
for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {k.split('::')[-1][:40]:40s} calls {v[0]:5d}  total {v[1]:8.2f} ms  ({100*v[1]/tot:4.1f} %)  avg {1e3*v[1]/v[0]:8.1f} us")
