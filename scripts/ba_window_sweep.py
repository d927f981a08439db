"""Local-BA window sweep: the three phases of the plugin protocol (setParams / optimize / getResults) against the number of FREE
keyframes, 3000 landmarks, every form the problem size selects (persistent one-launch / legacy launch chain), checked against the
CPU oracle (state within 1e-6, identical iteration counts).  The reference's window is whatever the covisibility neighbourhood
yields (mapmanager.cpp:11085-11413), not a constant: there must be no cliff between neighbouring sizes.
usage: python scripts/ba_window_sweep.py [P] [nfree ...]      (UH_SWEEP_NO_ORACLE=1 skips the CPU check for the large windows)"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import synth, oracle_lib
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet

torch.cuda.set_device(0)
ctx = u.Context(0, private=True)
O = None if os.environ.get("UH_SWEEP_NO_ORACLE") else oracle_lib.load_oracle()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sizes = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 6, 8, 9, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 64]
rows = []
for nfree in sizes:
    K = nfree + 2
    pr = synth.ba_problem(K, P, seed=nfree, nfixed=2)
    opt = GlobalOptimizer.create(ctx)
    ps = ParamSet(nIters=5)
    opt.setParams(pr, ps)
    opt.optimize()
    got = opt.getResults()
    row = {"nfree": nfree, "K": K, "P": pr["P"], "E": pr["E"], "iters": got["iters"].tolist(), "form": opt.form() if hasattr(opt, "form") else "?"}
    if O is not None:
        ref = oracle_lib.ba_optimize(O, pr, 5)
        row["state_err"] = float(np.abs(got["state"] - ref["state"]).max())
        row["iters_equal"] = bool((got["iters"] == ref["iters"]).all())
        row["bad_equal"] = float((got["bad"] == ref["bad"]).mean())
    N = 12
    ts, to, tg = [], [], []
    for _ in range(N):
        t0 = time.perf_counter(); opt.setParams(pr, ps)
        t1 = time.perf_counter(); opt.optimize()
        t2 = time.perf_counter(); opt.getResults()
        t3 = time.perf_counter()
        ts.append(t1 - t0); to.append(t2 - t1); tg.append(t3 - t2)
    med = lambda v: 1e3 * float(np.median(v))
    row.update(set_ms=round(med(ts), 4), opt_ms=round(med(to), 4), get_ms=round(med(tg), 4))
    rows.append(row)
    print(json.dumps(row), flush=True)
    opt.close()
print("nfree  E      form      set_ms  opt_ms  get_ms  ratio_to_prev")
prev = None
for r in rows:
    print(f"{r['nfree']:5d} {r['E']:6d} {r['form']:9s} {r['set_ms']:7.3f} {r['opt_ms']:7.3f} {r['get_ms']:7.3f}  {'' if prev is None else round(r['opt_ms'] / prev, 2)}")
    prev = r["opt_ms"]
