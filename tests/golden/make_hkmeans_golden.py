"""Generates tests/golden/hkmeans_golden.npz by running the REAL reference xflann (oracle/_ref/libxflann_ref.so, compiled from
/root/reference/3rdparty/xflann by oracle/Makefile): the serialised hierarchical k-means index (xflann::Index::toStream of
build(features, HKMeansParams(k, 0))) and the rows of search(KnnSearchParams(maxChecks, sorted)) on seeded inputs.
Run in the build container only:  python tests/golden/make_hkmeans_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

ref = oracle_lib.load_ref("xflann")
assert ref is not None and hasattr(ref, "xflann_ref_hkmeans_stream"), "build oracle/_ref first (make -C oracle ref)"
rng = np.random.default_rng(21)
low = np.zeros((700, 32), np.uint8)
low[:, :2] = rng.integers(0, 256, (700, 2))
cases = {"rand": synth.match_set(64, 900, seed=31), "low_entropy": (low, low[rng.integers(0, 700, 48)].copy()),
         "tiny": (rng.integers(0, 256, (5, 32), dtype=np.uint8), rng.integers(0, 256, (6, 32), dtype=np.uint8)),
         "k_plus_1": (rng.integers(0, 256, (33, 32), dtype=np.uint8), rng.integers(0, 256, (8, 32), dtype=np.uint8))}
out = {}
for name, (train, q) in cases.items():
    out[f"{name}_train"] = train
    out[f"{name}_q"] = q
    for k in (32, 8):
        stream = oracle_lib.ref_hkmeans_stream(ref, train, k, 0)
        blob = stream[64:]
        if name in ("tiny", "k_plus_1"):
            out[f"{name}_k{k}_blob"] = blob.copy()                     # small: keep the bytes themselves
        out[f"{name}_k{k}_blob_sha256"] = np.frombuffer(hashlib.sha256(blob.tobytes()).digest(), np.uint8).copy()
        out[f"{name}_k{k}_blob_size"] = np.array([len(blob)], np.int64)
        for nn, mc, s in ((10, 16, 0), (10, 16, 1), (5, 1, 0), (3, 40, 0), (2, 3, 0)):
            i, d = oracle_lib.ref_hkmeans_search(ref, train, q, nn, k, 0, mc, s)
            out[f"{name}_k{k}_nn{nn}_mc{mc}_s{s}_idx"] = i
            out[f"{name}_k{k}_nn{nn}_mc{mc}_s{s}_dist"] = d
np.savez_compressed(os.path.join(HERE, "hkmeans_golden.npz"), **out)
print("wrote hkmeans_golden.npz", sum(v.nbytes for v in out.values()), "bytes raw")
