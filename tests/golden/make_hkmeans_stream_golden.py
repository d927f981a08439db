"""Generates tests/golden/hkmeans_stream_golden.npz: the COMPLETE byte stream xflann::Index::toStream writes (8-byte signature, 8-byte
implementation hash, KMeansIndex signature + params + block data) for small seeded train sets, produced by the REAL xflann
(oracle/_ref/libxflann_ref.so, compiled from /root/reference/3rdparty/xflann by oracle/Makefile), plus rows the real library
returns for queries against the index it built.  Run in the build container only:  python tests/golden/make_hkmeans_stream_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

ref = oracle_lib.load_ref("xflann")
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
rng = np.random.default_rng(77)
out = {}
for name, n in (("s40", 40), ("s300", 300)):
    train = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (24, 32), dtype=np.uint8)
    out[f"{name}_train"], out[f"{name}_q"] = train, q
    for k in (32, 8):
        out[f"{name}_k{k}_stream"] = oracle_lib.ref_hkmeans_stream(ref, train, k, 0).copy()
        i, d = oracle_lib.ref_hkmeans_search(ref, train, q, 10, k, 0, 16, 0)
        out[f"{name}_k{k}_idx"], out[f"{name}_k{k}_dist"] = i, d
np.savez_compressed(os.path.join(HERE, "hkmeans_stream_golden.npz"), **out)
print("wrote hkmeans_stream_golden.npz", sum(v.nbytes for v in out.values()), "bytes raw")
