"""The persistent local BA beside OTHER PROCESSES on the same GPU ("the time-outs + chain fallback are tested only through
UH_BA_FAIL_RESIDENCY=1, never under a real second tenant").  The measured process optimises the bench problem (10 keyframes x 3000 landmarks, the
real g2o's result in tests/golden/sweep_golden.npz) over and over while the tenants run in processes of their own:
  gemm     one process multiplying 8192^3 fp32 matrices back to back (every CU busy with long-running workgroups of another queue)
  ba1      one more process running this same persistent BA in a loop (two spin-waiting kernels, 94 + 94 workgroups: both fit)
  ba2      two more (3 x 94 = 282 workgroups > 256 CUs: somebody's workgroups are not all resident at once)
  ba2gemm  both kinds
For every optimisation: state / iteration counts / bad flags against the real g2o, the form it ran in (a residency time-out sends this and the
next 64 problems to the launch chain), the wall time.  usage: python scripts/ba_second_tenant.py [seconds per scenario]   (run under `timeout`)"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np

SECS = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 8.0


def tenant(kind, ready, stop):
    import torch
    torch.cuda.set_device(0)
    if kind == "gemm":
        a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
        torch.matmul(a, b); torch.cuda.synchronize()
        open(ready, "w").write("1")
        while not os.path.exists(stop):
            for _ in range(8): c = torch.matmul(a, b)
            torch.cuda.synchronize()
    else:
        import synth, ucoslam_cv3_amd as u
        from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
        ctx = u.Context(0, private=True)
        pr = synth.ba_problem(10, 3000, 1)
        opt = GlobalOptimizer.create(ctx)
        opt.setParams(pr, ParamSet(nIters=5)); opt.optimize()
        open(ready, "w").write("1")
        n = 0
        while not os.path.exists(stop):
            opt.setParams(pr, ParamSet(nIters=5)); opt.optimize(); n += 1
        print(f"   tenant ba: {n} optimisations, last form {opt.form()}", flush=True)


if len(sys.argv) > 2 and sys.argv[1] == "--tenant":
    tenant(sys.argv[2], sys.argv[3], sys.argv[4])
    sys.exit(0)

import torch
import synth, make_sweep_golden as G
import ucoslam_cv3_amd as u
from ucoslam_cv3_amd.ba import GlobalOptimizer, ParamSet
torch.cuda.set_device(0)
gold = np.load(os.path.join(R, "tests", "golden", "sweep_golden.npz"))
case = G.BA_CASES[0]
key = f"ba_{case[0]}x{case[1]}_s{case[2]}_f{case[3]}"
pr = synth.ba_problem(case[0], case[1], case[2], nfixed=case[3])
ctx = u.Context(0, private=True)


def measure(label, kinds):
    tag = f"/tmp/uh_tenant_{os.getpid()}_{label}"
    stop = tag + ".stop"
    procs, readies = [], []
    for i, k in enumerate(kinds):
        rd = f"{tag}.{i}.ready"
        readies.append(rd)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--tenant", k, rd, stop]))
    t0 = time.time()
    while not all(os.path.exists(r) for r in readies):
        if time.time() - t0 > 240 or any(p.poll() is not None for p in procs):
            print(f"{label}: a tenant did not start"); break
        time.sleep(0.2)
    opt = GlobalOptimizer.create(ctx)
    lat, forms, bad, errs = [], {}, 0, 0
    t0 = time.time()
    while time.time() - t0 < SECS:
        try:
            t = time.perf_counter()
            opt.setParams(pr, ParamSet(nIters=5))
            opt.optimize()
            r = opt.getResults()
            lat.append(1e3 * (time.perf_counter() - t))
            f = opt.form(); forms[f] = forms.get(f, 0) + 1
            ok = r["iters"].tolist() == gold[key + "_iters"].tolist() and np.abs(r["state"] - gold[key + "_state"]).max() < 1e-6 and int(r["bad"].sum()) == int(gold[key + "_nbad"])
            bad += 0 if ok else 1
        except Exception as e:   # an error return is reported, never swallowed
            errs += 1
            if errs <= 3: print(f"   {label}: {e!r}"[:300])
    open(stop, "w").write("1")
    for p in procs:
        try: p.wait(timeout=60)
        except Exception: p.kill()
    for f in readies + [stop]:
        if os.path.exists(f): os.remove(f)
    a = np.sort(np.array(lat)) if lat else np.zeros(1)
    print(f"{label:8s}: {len(lat)} optimisations, {bad} differ from the real g2o, {errs} error returns; forms {forms}; "
          f"setParams+optimize+getResults ms: p50 {a[len(a) // 2]:.3f} p90 {a[len(a) * 9 // 10]:.3f} p99 {a[len(a) * 99 // 100]:.3f} max {a[-1]:.3f}", flush=True)


measure("alone", [])
measure("gemm", ["gemm"])
measure("ba1", ["ba"])
measure("ba2", ["ba", "ba"])
measure("ba2gemm", ["ba", "ba", "gemm"])
