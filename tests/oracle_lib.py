"""Loader for the TEST-ONLY CPU oracle (oracle/liboracle.so) and the real-reference builds (oracle/_ref/)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
VP, I, SZ = C.c_void_p, C.c_int, C.c_size_t


def P(a):
    return a.ctypes.data_as(VP)


_oracle = None


def load_oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    L = C.CDLL(path)
    L.oracle_knn_search.restype = I
    L.oracle_knn_search.argtypes = [VP, I, SZ, VP, I, SZ, I, I, I, I, I, VP, VP]
    _oracle = L
    return L


def load_ref(name):
    """oracle/_ref/lib<name>_ref.so or None when it was never built (reference tree absent)."""
    path = os.path.join(ORACLE_DIR, "_ref", f"lib{name}_ref.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


def knn_search(L, train, queries, nn, sorted_=0, max_dist=-1, t_begin=0, t_end=-1):
    import numpy as np

    nt, nq = len(train), len(queries)
    idx = np.empty((nq, nn), np.int32)
    dist = np.empty((nq, nn), np.int32)
    rc = L.oracle_knn_search(P(train), nt, train.strides[0] if nt else 32, P(queries), nq,
                             queries.strides[0] if nq else 32, nn, int(sorted_), max_dist, t_begin, t_end, P(idx), P(dist))
    assert rc == 0
    return idx, dist
