"""Pins the bag-of-words stage to the REAL fbow — to be run ONCE where OpenCV's development files exist AND the reference tree is
present (this build container has no OpenCV, so fbow cannot be compiled here and the file this writes is absent from the repo).

    make -C oracle ref-fbow            # compiles /root/reference/3rdparty/fbow/fbow/fbow.cpp + oracle/ref_drivers/fbow_ref.cpp
    python tests/golden/make_fbow_golden.py

Writes tests/golden/fbow_golden.npz: a synthetic vocabulary in the reference's binary stream format (tests/synth.vocabulary, the
real `orb.fbow` is not in the reference tree), 700 descriptors, and what fbow::Vocabulary::transform(desc, 3, fBow&, fBow2&) and
fBow::score returned for them.  tests/test_opencv_golden.py::test_oracle_fbow_equals_real_fbow compares the oracle with it."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, ROOT)
import synth  # noqa: E402


def main():
    so = os.path.join(ROOT, "oracle", "_ref", "libfbow_ref.so")
    if not os.path.exists(so):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref-fbow"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(so):
            print("could not build oracle/_ref/libfbow_ref.so (needs OpenCV development files and /root/reference):\n" + r.stdout[-2000:] + r.stderr[-2000:])
            return 2
    from ucoslam_cv3_amd.bow import write_vocabulary_stream

    L = C.CDLL(so)
    params, blob, meta = synth.vocabulary(k=10, depth=4, seed=3, aligment=8)
    stream = write_vocabulary_stream(params, blob)
    rng = np.random.default_rng(2)
    desc = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    level, cap = 3, 4096
    bag_ids, bag_w = np.zeros(cap, np.uint32), np.zeros(cap, np.float32)
    node_ids, node_ptr, node_feats = np.zeros(cap, np.uint32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.uint32)
    nn, nf, score = C.c_int(0), C.c_int(0), C.c_double(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    L.fbow_ref_transform.restype = C.c_int
    nb = L.fbow_ref_transform(stream, C.c_size_t(len(stream)), P(desc), 700, 32, level, P(bag_ids), P(bag_w), cap, P(node_ids), P(node_ptr), cap,
                              P(node_feats), cap, C.byref(nn), C.byref(nf), C.byref(score))
    assert 0 <= nb <= cap and nn.value <= cap and nf.value <= cap, (nb, nn.value, nf.value)
    np.savez_compressed(os.path.join(HERE, "fbow_golden.npz"), params=np.frombuffer(params, np.uint8), blob=np.frombuffer(blob, np.uint8), desc=desc,
                        level=np.array(level), bag_ids=bag_ids[:nb], bag_weights=bag_w[:nb], node_ids=node_ids[:nn.value], node_ptr=node_ptr[:nn.value + 1],
                        node_feats=node_feats[:nf.value], self_score=np.array(score.value))
    print("wrote tests/golden/fbow_golden.npz:", nb, "words,", nn.value, "nodes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
