"""Generates tests/golden/knn_golden.npz by running the REAL reference xflann (oracle/_ref/libxflann_ref.so,
compiled from /root/reference/3rdparty/xflann by oracle/Makefile) on seeded inputs.  Run in the build
container only:  python tests/golden/make_knn_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

ref = oracle_lib.load_ref("xflann")
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
P = oracle_lib.P
out = {}
cases = {
    "rand": synth.match_set(96, 777, seed=11),
    "ties": synth.tie_stress_set(64, 500, seed=12),
    "desc": synth.descending_set(4, 300, seed=13),
    "tiny": synth.match_set(5, 3, seed=14),
}
for name, (train, q) in cases.items():
    out[f"{name}_train"] = train
    out[f"{name}_q"] = q
    for nn in (1, 2, 10):
        for s in (0, 1):
            idx = np.empty((len(q), nn), np.int32)
            dist = np.empty((len(q), nn), np.int32)
            rc = ref.xflann_ref_linear_search(P(train), len(train), P(q), len(q), nn, s, 1, P(idx), P(dist))
            assert rc == 0
            out[f"{name}_nn{nn}_s{s}_idx"] = idx
            out[f"{name}_nn{nn}_s{s}_dist"] = dist
np.savez_compressed(os.path.join(HERE, "knn_golden.npz"), **out)
print("wrote knn_golden.npz", sum(v.nbytes for v in out.values()), "bytes raw")
