"""Generates tests/golden/kdtree_golden.npz by running the REAL reference kd-tree (src/basictypes/picoflann.h compiled into
oracle/_ref/libpicoflann_ref.so): radiusSearch(sorted=false) hits IN ORDER with their squared distances, on seeded clouds.
Run in the build container only:  python tests/golden/make_kdtree_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

ref = oracle_lib.load_ref("picoflann")
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
rng = np.random.default_rng(77)
clouds = {"uniform": (rng.random((600, 2)) * [1241, 376]).astype(np.float32),
          "ties": (rng.integers(0, 50, (500, 2)).astype(np.float32) * np.float32(1.2) ** rng.integers(0, 3, (500, 1)).astype(np.float32)).astype(np.float32),
          "small": (rng.random((11, 2)) * 100).astype(np.float32)}
out = {}
for name, xy in clouds.items():
    kd = oracle_lib.KdOracle(ref, "picoflann_ref", xy)
    qs, rs, offs, idxs, sqds = [], [], [0], [], []
    for t in range(120):
        q = xy[rng.integers(len(xy))] if t % 3 == 0 else (xy.min(0) - 20 + rng.random(2) * (xy.max(0) - xy.min(0) + 40)).astype(np.float32)
        r = np.float32([3.0, 15.0, 41.5, 90.0][t % 4])
        i, d = kd.radius(q[0], q[1], r)
        qs.append(q); rs.append(r); idxs.append(i); sqds.append(d); offs.append(offs[-1] + len(i))
    out[f"{name}_xy"] = xy
    out[f"{name}_q"] = np.array(qs, np.float32)
    out[f"{name}_r"] = np.array(rs, np.float32)
    out[f"{name}_off"] = np.array(offs, np.int64)
    out[f"{name}_idx"] = np.concatenate(idxs).astype(np.uint32)
    out[f"{name}_sqd"] = np.concatenate(sqds).astype(np.float64)
np.savez_compressed(os.path.join(HERE, "kdtree_golden.npz"), **out)
print("wrote kdtree_golden.npz", sum(v.nbytes for v in out.values()), "bytes raw")
