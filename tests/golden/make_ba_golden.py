"""Generates tests/golden/ba_golden.npz by running the REAL reference g2o (oracle/_ref/libg2o_ref.so, compiled from
/root/reference/3rdparty/g2o by oracle/Makefile with the driver oracle/ref_drivers/g2o_ref.cpp) on a seeded problem.
Run in the build container only:  python tests/golden/make_ba_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

ref = oracle_lib.load_ref("g2o")
assert ref is not None
pr = synth.ba_problem(8, 600, seed=77, nfixed=2)
out = oracle_lib.ba_optimize_ref(ref, pr, 5)
save = {f"in_{k}": pr[k] for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w")}
save.update(ref_state=out["state"], ref_poses=out["poses"], ref_points=out["points"], ref_bad=out["bad"], ref_iters=out["iters"],
            ref_chi2=out["chi2"])
np.savez_compressed(os.path.join(HERE, "ba_golden.npz"), **save)
print("wrote ba_golden.npz", pr["K"], pr["P"], pr["E"])
