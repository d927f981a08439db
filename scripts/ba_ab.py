"""A/B of the two local-BA forms on the GPU: persistent one-launch kernel (default) vs legacy two-launches-per-trial
(UH_BA_FORM=legacy), both against the CPU oracle; timing of optimize() and the persistent kernel's phase clocks.
usage: python scripts/ba_ab.py [K P seed nfix]..."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
import ctypes as C

torch.cuda.set_device(0)
ctx = u.Context(0, torch.cuda.current_stream().cuda_stream)
O = oracle_lib.load_oracle()
L = u.lib()
L.uh_ba_debug_clocks.argtypes = [C.c_void_p, C.c_void_p]; L.uh_ba_debug_clocks.restype = C.c_int
cfgs = [(10, 3000, 0, 2), (6, 400, 1, 1), (4, 150, 2, 2), (10, 800, 3, 2), (9, 5000, 4, 1), (5, 33, 5, 2)]
if len(sys.argv) > 4:
    a = list(map(int, sys.argv[1:])); cfgs = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
for K, P, seed, nfix in cfgs:
    pr = synth.ba_problem(K, P, seed, nfixed=nfix)
    ref = oracle_lib.ba_optimize(O, pr, 5)
    out = {}
    for form in ("persist", "legacy"):
        os.environ["UH_BA_FORM"] = "legacy" if form == "legacy" else "persist"
        opt = GlobalOptimizer.create(ctx)
        opt.setParams(pr, ParamSet(nIters=5))
        try:
            opt.optimize()
        except Exception as e:
            print(form, "FAILED:", e); continue
        got = opt.getResults()
        for _ in range(3): opt.optimize()
        torch.cuda.synchronize(); t = time.perf_counter(); N = 20
        for _ in range(N): opt.optimize()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
        again = opt.getResults()
        det = np.array_equal(again["state"], got["state"])
        err = np.abs(got["state"] - ref["state"]).max()
        print(f"K{K} P{pr['P']} E{pr['E']} nfree{K-nfix} {form:13s}: {dt*1e3:7.3f} ms  iters {got['iters'].tolist()} (oracle {ref['iters'].tolist()})  |state-oracle| {err:.2e}  "
              f"chi2 err {np.abs(got['chi2']-ref['chi2']).max():.2e}  bad== {(got['bad']==ref['bad']).mean():.5f}  deterministic {det}")
        out[form] = got
        if form.startswith("persist"):
            clk = np.zeros(64, dtype=np.int64); L.uh_ba_debug_clocks(opt._h, clk.ctypes.data)
            us = lambda a, b: (clk[b] - clk[a]) / 100.0
            spec = os.environ.get("UH_BA_SPEC", "1") != "0"   # (a speculative trial has no errors / C / decision phases of its own: clocks 47, 48 are stale)
            tailp = f"| speculative trials kept / dropped {clk[58]} / {clk[59]} " if spec else f"| errors {us(46,47):.2f} | C {us(47,48):.2f} | decide {us(48,49):.2f} "
            print(f"    pass 2, 4th loop iteration: lin + phase1 {us(40,41):.2f} | A+slices+B {us(41,42):.2f} (wait A {us(41,50):.2f}, slice {us(50,51):.2f}, wait B {us(51,42):.2f}) | assemble (+ decision) {us(42,43):.2f} (reduced vector fetched + scattered {us(42,57):.2f}, barrier {us(57,62):.2f}, decision {us(62,43):.2f}) | factor {us(43,44):.2f} "
                  f"| backsolve {us(44,45):.2f} | pose+backsub {us(45,46):.2f} {tailp}| total {us(40,49):.2f} us")
            print(f"    kernel: setup {us(0,1):.2f} | pass 1: begin+opening {us(2,3):.2f}, trials {us(3,5):.2f} | pass 2: relabel+opening {us(5,6):.2f}, trials {us(6,8):.2f} | total to results {us(0,8):.2f} us")
            print(f"    results: stores issued {us(8,11):.2f} | acknowledged + barrier {us(11,12):.2f} | count of all workgroups {us(12,9):.2f}")
            print(f"    results hand-over (workgroup 0): result block + count {us(8,9):.2f} | copy of its slice to the pinned host block {us(9,10):.2f} us")
            print(f"    phase1: edges {us(40,52):.2f} | butterfly+chol+Y {us(52,53):.2f} | block sums+camsum {us(53,54):.2f} | product+stores {us(54,55):.2f} | tail {us(55,41):.2f} (wait for the other waves' product {us(55,56):.2f}, fold + barrier {us(56,60):.2f}, camera-sum stores {us(60,61):.2f}, product stores {us(61,41):.2f})")
    if "persist" in out and "legacy" in out:
        print(f"    persist vs legacy |state| {np.abs(out['persist']['state']-out['legacy']['state']).max():.2e}")
