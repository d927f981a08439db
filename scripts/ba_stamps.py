import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np, synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
pr = synth.ba_problem(10, 3000, 0)
opt = GlobalOptimizer.create(ctx); opt.setParams(pr, ParamSet(nIters=5)); opt.optimize(); opt.optimize()
st = np.zeros(16, np.uint64)
u.lib().uh_ba_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
u.lib().uh_ba_debug_stamps(opt._h, st.ctypes.data_as(C.c_void_p))
d = np.diff(st[:6].astype(np.int64))
print("solve kernel cycle deltas: assemble %d factor %d (sync) %d scale %d subst %d pose %d" % (d[0], d[1], 0, d[2], d[3], d[4]))
