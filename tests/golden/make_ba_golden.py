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
# The first fixture's information scalars are full doubles 1 / 1.2^octave (synth.ba_problem(double_weights=True)): not float-exact, unlike
# the reference's (double)(float) values — on purpose, it is the test of the 24-byte observation records.  The generator reproduces the
# committed file bit for bit (checked below against the file it is about to replace).
pr = synth.ba_problem(8, 600, seed=77, nfixed=2, double_weights=True)
out = oracle_lib.ba_optimize_ref(ref, pr, 5)
save = {f"in_{k}": pr[k] for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w")}
save.update(ref_state=out["state"], ref_poses=out["poses"], ref_points=out["points"], ref_bad=out["bad"], ref_iters=out["iters"],
            ref_chi2=out["chi2"])
path = os.path.join(HERE, "ba_golden.npz")
if os.path.exists(path):
    old = np.load(path)
    diff = [k for k in save if k not in old.files or not np.array_equal(old[k], save[k])]
    print("ba_golden.npz:", "regenerated bit for bit" if not diff else f"DIFFERS from the committed file in {diff}")
np.savez_compressed(path, **save)
print("wrote ba_golden.npz", pr["K"], pr["P"], pr["E"])

# Second fixture: problems Levenberg-Marquardt does NOT sail through (synth.ba_hard_problem: rejected trials, lambda factors other than
# 1/3, passes ended early) — the real g2o's answers for the branches the first fixture never takes.  Seeds chosen among those on which
# the outcome is well conditioned (the restated oracle agrees with g2o to 1e-9): 2 (a pass that ends on the chi2 criterion after 6
# iterations), 3 (lambda factors other than 1/3), 7 and 13 (rejections, second pass cut to 2 iterations), 8 (ten rejections in a row).
hard = {}
for seed in (2, 3, 7, 8, 13):
    pr = synth.ba_hard_problem(seed)
    out = oracle_lib.ba_optimize_ref(ref, pr, 5)
    for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w"):
        hard[f"s{seed}_in_{k}"] = pr[k]
    hard.update({f"s{seed}_ref_state": out["state"], f"s{seed}_ref_bad": out["bad"], f"s{seed}_ref_iters": out["iters"], f"s{seed}_ref_chi2": out["chi2"],
                 f"s{seed}_ref_points": out["points"], f"s{seed}_ref_poses": out["poses"]})
    print("hard seed", seed, pr["K"], pr["P"], pr["E"], out["iters"])
hard["seeds"] = np.array([2, 3, 7, 8, 13])
np.savez_compressed(os.path.join(HERE, "ba_hard_golden.npz"), **hard)
