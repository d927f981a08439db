import os, sys, time
R='/root/repo'
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx = u.Context(0, private=True)
probs=[synth.ba_problem(10,3000,s) for s in range(3)]+[synth.ba_problem(8,1500,7), synth.ba_problem(10,5000,9)]
for lw in sys.argv[1:]:
    if lw!='0': os.environ['UH_BA_LW']=lw
    else: os.environ.pop('UH_BA_LW',None)
    out=[]
    for pr in probs:
        opt=GlobalOptimizer.create(ctx); opt.setParams(pr, ParamSet(nIters=5))
        for _ in range(3): opt.optimize()
        ts=[]
        for _ in range(15):
            t=time.perf_counter(); opt.optimize(); ts.append(time.perf_counter()-t)
        out.append(round(1e3*float(np.median(ts)),4))
    print('LW',lw,out,flush=True)
