#!/usr/bin/env python
"""Pose-only solve (uh_pnp_solve, host in / host out): wall time per call and the kernel's own clock stamps, by match count."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.pnp import PnPSolver

ctx = u.Context(0, private=True)
sol = PnPSolver(ctx)
for n in (100, 300, 600, 800, 1300, 1500, 3000, 4000):
    pr = synth.pnp_problem(n, seed=3)
    args = (pr["pose"], pr["intr"], pr["p3d"], pr["kp"], pr["invsig"], pr["weight"])
    for _ in range(5):
        r = sol.solvePnp(*args)
    t = time.perf_counter()
    for _ in range(50):
        r = sol.solvePnp(*args)
    wall = (time.perf_counter() - t) / 50 * 1e6
    sol.debug_clocks(True)
    sol.solvePnp(*args)
    c = sol.debug_clocks(True)
    if os.environ.get("PNP_TRACE"):
        sol.debug_clocks(True); sol.solvePnp(*args); c = sol.debug_clocks(True)
        tr = c[8:8 + int(c[4]) + 2]
        print("   trials (round.it:qmax+/-):", " ".join(f"{(v >> 16) & 255}.{(v >> 8) & 255}:{v & 255}{'+' if v >> 24 else '-'}" for v in tr if v))
    sol.debug_clocks(False)
    tot = c[3] - c[0]
    print("   wave 0 cycles: prepare(solve+update) %d  barrierA %d  matches+butterfly %d  barrierB %d  totals+decision %d  ladder passes %d" % tuple(c[16:22]))
    print(f"n={n:5d} wall {wall:7.1f} us  iters {r['iters'].tolist()} passes {c[4]}  clk: stage {c[1]-c[0]} rounds {c[2]-c[1]} post {c[3]-c[2]} total {tot}  per pass {(c[2]-c[1])/max(c[4],1):.0f}")
