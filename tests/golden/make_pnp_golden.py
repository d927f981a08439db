"""Generates tests/golden/pnp_golden.npz from the REAL reference g2o (oracle/_ref/libg2o_ref.so, driver
oracle/ref_drivers/g2o_ref.cpp::g2o_ref_pnp_solve).  Build container only:  python tests/golden/make_pnp_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

ref = oracle_lib.load_ref("g2o")
assert ref is not None
pr = synth.pnp_problem(500, seed=99)
out = oracle_lib.pnp_solve_ref(ref, pr)
save = {f"in_{k}": pr[k] for k in ("pose", "intr", "p3d", "kp", "invsig", "weight")}
save.update(ref_state=out["state"], ref_pose=out["pose"], ref_bad=out["bad"], ref_iters=out["iters"], ref_ngood=np.int32(out["ngood"]))
np.savez_compressed(os.path.join(HERE, "pnp_golden.npz"), **save)
print("wrote pnp_golden.npz", out["ngood"], out["iters"])
