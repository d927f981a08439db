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
# The committed ba_golden.npz was generated when synth.ba_problem still produced full-double information weights (1 / 1.2^octave); the
# generator has since moved to the reference's (double)(float) weights.  The old fixture is kept on purpose: its weights are not
# float-exact, which makes it the test of the 24-byte observation records.  Pass --first to regenerate it with today's weights.
if "--first" in sys.argv:
    np.savez_compressed(os.path.join(HERE, "ba_golden.npz"), **save)
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
