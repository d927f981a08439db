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

if opt.form().startswith("persist"):   # the persistent kernel's phase clocks are read by scripts/ba_ab.py (other indices)
    sys.exit(0)
# phase clocks of the last executed LM step of the LAUNCH CHAIN (10 ns ticks), see UH_BA_CLK in csrc/ba.hip (UH_BA_FORM=legacy)
import ctypes as C
L = u.lib()
L.uh_ba_debug_clocks.argtypes = [C.c_void_p, C.c_void_p]
L.uh_ba_debug_clocks.restype = C.c_int
clk = np.zeros(64, dtype=np.int64)
L.uh_ba_debug_clocks(opt._h, clk.ctypes.data)
us = lambda a, b: (clk[b] - clk[a]) / 100.0
print(f"lin block0 {us(0,1):.2f} us | schur block0 {us(4,5):.2f} | solve: assemble {us(10,11):.2f} factor {us(11,12):.2f} subst {us(12,13):.2f} update {us(13,14):.2f} "
      f"| backsub block0 {us(20,21):.2f} | decide {us(24,25):.2f}")
print(f"assemble: issue {us(10,26):.2f} state {us(26,27):.2f} sums+stores {us(27,28):.2f} barrier {us(28,11):.2f}")
print(f"schur block0 end -> solve start (same step, next launch) {us(5,10):.2f} us")
if clk[32 + 12] and clk[32 + 11]:
    print(f"shader clock during solve factor phase: {(clk[32+12]-clk[32+11]) / us(11,12):.0f} cycles/us")
