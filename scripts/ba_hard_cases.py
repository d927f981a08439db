"""Local BAs that Levenberg-Marquardt does NOT sail through: large pose / landmark / pixel noise, 30 % outliers — trials are rejected (up to
the ten-in-a-row that terminates a pass), accepted with lambda factors other than 1/3, passes end early on the chi2 criterion.  Every
form against the CPU oracle; for the persistent form the number of speculative trials kept / dropped (debug clocks 58 / 59).
usage: python scripts/ba_hard_cases.py [n_cases]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import ctypes as C
import numpy as np, torch
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet

torch.cuda.set_device(0)
ctx = u.Context(0, private=True)
O = oracle_lib.load_oracle()
L = u.lib()
L.uh_ba_debug_clocks.argtypes = [C.c_void_p, C.c_void_p]; L.uh_ba_debug_clocks.restype = C.c_int
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
worst = 0.0
for seed in range(n):
    pr = synth.ba_hard_problem(seed)
    ref = oracle_lib.ba_optimize(O, pr, 5)
    line = f"seed {seed:3d} K{pr['K']} P{pr['P']} E{pr['E']} oracle iters {ref['iters'].tolist()}"
    for form, env in (("spec", {}), ("nospec", {"UH_BA_SPEC": "0"}), ("chain", {"UH_BA_FORM": "legacy"})):
        for k in ("UH_BA_SPEC", "UH_BA_FORM"): os.environ.pop(k, None)
        os.environ.update(env)
        opt = GlobalOptimizer.create(ctx)
        opt.setParams(pr, ParamSet(nIters=5)); opt.optimize(); got = opt.getResults()
        clk = np.zeros(64, dtype=np.int64); L.uh_ba_debug_clocks(opt._h, clk.ctypes.data)
        err = float(np.abs(got["state"] - ref["state"]).max())
        same = got["iters"].tolist() == ref["iters"].tolist()
        badeq = float((got["bad"] == ref["bad"]).mean())
        worst = max(worst, err)
        line += f" | {form}: iters {'==' if same else got['iters'].tolist()} err {err:.1e} bad {badeq:.4f}" + (f" kept/dropped {clk[58]}/{clk[59]}" if form == "spec" else "")
        opt.close()
    print(line, flush=True)
print("worst state error", worst)
