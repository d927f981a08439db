import sys, os
R='/root/repo'; sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import torch, numpy as np, synth, ctypes as C
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
ctx=u.Context(0, private=True)
L=u.lib(); L.uh_ba_debug_clocks.argtypes=[C.c_void_p,C.c_void_p]; L.uh_ba_debug_clocks.restype=C.c_int
for nfree in ([int(a) for a in sys.argv[1:]] or [17, 21]):
    pr=synth.ba_problem(nfree+2,3000,nfree); opt=GlobalOptimizer.create(ctx); opt.setParams(pr, ParamSet(nIters=5))
    for _ in range(3): opt.optimize()
    clk=np.zeros(64,dtype=np.int64); L.uh_ba_debug_clocks(opt._h, clk.ctypes.data)
    us=lambda a,b:(clk[b]-clk[a])/100.0
    print(nfree, f"schur block0 {us(4,5):.1f} (dense: panel {us(4,6):.1f} product {us(6,7):.1f} write {us(7,5):.1f}; with -DUH_BA_DENSE_CLK: entry -> prologue done {us(15,4):.1f}, last product block ends {us(15,16):.1f}, last camera block {us(15,17):.1f} after entry) | solve: assemble {us(10,11):.1f} factor {us(11,12):.1f} subst {us(12,13):.1f} update {us(13,14):.1f} | backsub landmarks {us(20,21):.1f} | schur end -> solve start {us(5,10):.1f}")
