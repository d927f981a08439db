"""Per-kernel HIP-event averages of one optimize() of the launch chain (17+ free keyframes).  usage: python scripts/ba_chain_kernels.py [nfree ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx = u.Context(0, private=True)
for nfree in [int(a) for a in sys.argv[1:]] or [17]:
    pr = synth.ba_problem(nfree + 2, 3000, seed=nfree, nfixed=2)
    opt = GlobalOptimizer.create(ctx)
    opt.setParams(pr, ParamSet(nIters=5))
    for _ in range(3): opt.optimize()
    ctx.prof_enable(True); ctx.prof_reset()
    N = 5
    for _ in range(N): opt.optimize()
    torch.cuda.synchronize()
    rep = dict(ctx.prof_report()); ctx.prof_enable(False)
    tot = sum(v[1] for v in rep.values())
    print(f"nfree {nfree} form {opt.form()}: {tot / N:.3f} ms of kernels per optimize")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k.split('::')[-1][:40]:40s} {v[0] / N:6.1f} launches  {1e3 * v[1] / max(v[0], 1):8.1f} us each  {v[1] / N:7.3f} ms")
