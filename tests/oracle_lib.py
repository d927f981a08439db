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


# ------------------------------------------------------------------------------------------------ ORB
import numpy as _np

KEYPOINT_DTYPE = _np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                            ("octave", "<i4"), ("class_id", "<i4")])


def orb_extract(L, img, maxFeatures=2000, nlevels=8, scaleFactor=1.2, blur=True, nonmaxima=False):
    cap = max(maxFeatures, 1)
    kps = _np.zeros(cap, KEYPOINT_DTYPE)
    desc = _np.zeros((cap, 32), _np.uint8)
    fn = L.oracle_orb_extract_nonmaxima if nonmaxima else L.oracle_orb_extract
    if fn.argtypes is None:   # set once: bench.py calls this from several threads, and re-assigning argtypes races with a call in flight
        fn.restype = I
        fn.argtypes = [VP, I, I, SZ, I, I, C.c_float, I, VP, VP, I]
    img = _np.ascontiguousarray(img)
    n = fn(P(img), img.shape[1], img.shape[0], img.strides[0], maxFeatures, nlevels, scaleFactor, int(blur),
           P(kps), P(desc), cap)
    assert n >= 0, n
    return kps[:n].copy(), desc[:n].copy()


def orb_pyramid_level(L, img, level, nlevels=8, scaleFactor=1.2, blur=True):
    img = _np.ascontiguousarray(img)
    w, h = C.c_int(0), C.c_int(0)
    L.oracle_orb_pyramid_level.argtypes = [VP, I, I, SZ, I, C.c_float, I, I, VP, C.POINTER(I), C.POINTER(I)]
    assert L.oracle_orb_pyramid_level(P(img), img.shape[1], img.shape[0], img.strides[0], nlevels, scaleFactor, int(blur), level,
                                      None, C.byref(w), C.byref(h)) == 0
    out = _np.empty((h.value, w.value), _np.uint8)
    L.oracle_orb_pyramid_level(P(img), img.shape[1], img.shape[0], img.strides[0], nlevels, scaleFactor, int(blur), level,
                               P(out), C.byref(w), C.byref(h))
    return out


def fast_score_map(L, img):
    img = _np.ascontiguousarray(img)
    out = _np.empty_like(img)
    L.oracle_fast_score_map.argtypes = [VP, I, I, I, VP]
    L.oracle_fast_score_map(P(img), img.shape[1], img.shape[0], img.strides[0], P(out))
    return out


def fast_detect(L, img, threshold):
    img = _np.ascontiguousarray(img)
    cap = img.size // 4 + 16
    xys = _np.empty((cap, 3), _np.int32)
    L.oracle_fast_detect.argtypes = [VP, I, I, I, I, VP, I]
    n = L.oracle_fast_detect(P(img), img.shape[1], img.shape[0], img.strides[0], threshold, P(xys), cap)
    return xys[:n].copy()


# ------------------------------------------------------------------------------------------------ BA
def _ba_call(fn, pr, nIters, with_stop):
    K, Pn, E = pr["K"], pr["P"], pr["E"]
    out = dict(poses=_np.zeros((K, 16), _np.float32), points=_np.zeros((Pn, 3), _np.float32), chi2=_np.zeros(E, _np.float64),
               bad=_np.zeros(E, _np.uint8), iters=_np.zeros(2, _np.int32), state=_np.zeros((K, 7), _np.float64))
    args = [K, Pn, E, P(pr["poses"]), P(pr["fixed"]), P(pr["intr"]), P(pr["points"]), P(pr["obs_pt"]), P(pr["obs_kf"]),
            P(pr["obs_uv"]), P(pr["obs_w"]), nIters]
    if with_stop:
        args.append(None)
    args += [P(out["poses"]), P(out["points"]), P(out["chi2"]), P(out["bad"]), P(out["iters"]), P(out["state"])]
    fn.restype = I
    rc = fn(*args)
    assert rc == 0
    return out


def ba_optimize(L, pr, nIters=5):
    return _ba_call(L.oracle_ba_optimize, pr, nIters, True)


def ba_optimize_ref(R, pr, nIters=5):
    return _ba_call(R.g2o_ref_ba_optimize, pr, nIters, False)


# ------------------------------------------------------------------------------------------------ PnP
def _pnp_call(fn, pr):
    n = pr["n"]
    out = dict(pose=_np.zeros(16, _np.float32), bad=_np.zeros(max(n, 1), _np.uint8), iters=_np.zeros(4, _np.int32), state=_np.zeros(7, _np.float64))
    fn.restype = I
    out["ngood"] = fn(P(pr["pose"]), P(pr["intr"]), n, P(pr["p3d"]), P(pr["kp"]), P(pr["invsig"]), P(pr["weight"]), P(out["pose"]), P(out["bad"]),
                      P(out["iters"]), P(out["state"]))
    out["bad"] = out["bad"][:n]
    return out


def pnp_solve(L, pr):
    return _pnp_call(L.oracle_pnp_solve, pr)


def pnp_solve_ref(R, pr):
    return _pnp_call(R.g2o_ref_pnp_solve, pr)


# ------------------------------------------------------------------------------------------------ projection matcher (a27)
class KdOracle:
    """oracle_kd_* (restated picoflann) or picoflann_ref_* (the real header) behind one interface."""

    def __init__(self, lib, prefix, xy):
        import numpy as np

        self.L, self.pre = lib, prefix
        self.xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        b = getattr(lib, prefix + "_build")
        b.restype = VP
        b.argtypes = [VP, I]
        r = getattr(lib, prefix + "_radius")
        r.restype = I
        r.argtypes = [VP, C.c_float, C.c_float, C.c_double, VP, VP, I]
        getattr(lib, prefix + "_free").argtypes = [VP]
        self.h = b(P(self.xy), len(self.xy))

    def radius(self, qx, qy, radius):
        import numpy as np

        cap = max(len(self.xy), 1)
        idx = np.empty(cap, np.uint32)
        sqd = np.empty(cap, np.float64)
        n = getattr(self.L, self.pre + "_radius")(self.h, float(qx), float(qy), float(radius), P(idx), P(sqd), cap)
        return idx[:n].copy(), sqd[:n].copy()

    def export(self):
        import numpy as np

        n = len(self.xy)
        m = 2 * n + 2
        a = dict(col=np.zeros(m, np.int32), divlow=np.zeros(m, np.float32), divhigh=np.zeros(m, np.float32), left=np.zeros(m, np.int32),
                 right=np.zeros(m, np.int32), leaf_begin=np.zeros(m, np.int32), leaf_count=np.zeros(m, np.int32),
                 leaf_idx=np.zeros(max(n, 1), np.uint32), root_bbox=np.zeros(4, np.float64))
        f = self.L.oracle_kd_export
        f.restype = I
        f.argtypes = [VP] * 10
        nn = f(self.h, *[P(a[k]) for k in ("col", "divlow", "divhigh", "left", "right", "leaf_begin", "leaf_count", "leaf_idx", "root_bbox")])
        for k in ("col", "divlow", "divhigh", "left", "right", "leaf_begin", "leaf_count"):
            a[k] = a[k][:nn]
        a["leaf_idx"] = a["leaf_idx"][:n]
        return a

    def __del__(self):
        try:
            getattr(self.L, self.pre + "_free")(self.h)
        except Exception:
            pass


def proj_match(L, fr, mp, pose, minDescDist, maxRepjDist):
    """oracle_proj_match on the dicts made by synth.proj_problem; returns dict(best_kp, best_dist, visible, matches[n,4 int32 view])."""
    import numpy as np

    n = len(mp["ids"])
    best_kp = np.empty(n, np.int32)
    best_d = np.empty(n, np.float32)
    vis = np.empty(n, np.uint8)
    mout = np.zeros((max(n, 1), 4), np.int32)
    f = L.oracle_proj_match
    f.restype = I
    F = C.c_float
    f.argtypes = [VP, I, VP, VP, I, F, F, F, F, I, I, I, I, VP, I, VP, VP, VP, VP, VP, VP, F, F, VP, VP, VP, VP]
    pose = np.ascontiguousarray(pose, np.float32)
    k = f(P(fr["und_kpts"]), len(fr["und_kpts"]), P(fr["desc"]), P(fr["scale_factors"]), len(fr["scale_factors"]),
          fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"][0], fr["min_xy"][1], fr["max_xy"][0], fr["max_xy"][1], P(pose), n,
          P(mp["ids"]), P(mp["pos3d"]), P(mp["normal"]), P(mp["min_dist"]), P(mp["max_dist"]), P(mp["desc"]), minDescDist, maxRepjDist,
          P(best_kp), P(best_d), P(vis), P(mout))
    dm = np.zeros(k, dtype=np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]))
    dm[:] = mout[:k].copy().view(dm.dtype).reshape(-1)
    return dict(best_kp=best_kp, best_dist=best_d, visible=vis, matches=dm)


def proj_match_prev(L, fr, mp, pose, minDescDist, maxRepjDist):
    """oracle_proj_match_prev (tracker's search against the previous frame): mp needs ids, pos3d, octave, desc."""
    import numpy as np

    n = len(mp["ids"])
    best_kp = np.empty(n, np.int32)
    best_d = np.empty(n, np.float32)
    mout = np.zeros((max(n, 1), 4), np.int32)
    f = L.oracle_proj_match_prev
    f.restype = I
    F = C.c_float
    f.argtypes = [VP, I, VP, VP, I, F, F, F, F, I, I, I, I, VP, I, VP, VP, VP, VP, F, F, VP, VP, VP]
    pose = np.ascontiguousarray(pose, np.float32)
    k = f(P(fr["und_kpts"]), len(fr["und_kpts"]), P(fr["desc"]), P(fr["scale_factors"]), len(fr["scale_factors"]),
          fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["min_xy"][0], fr["min_xy"][1], fr["max_xy"][0], fr["max_xy"][1], P(pose), n,
          P(mp["ids"]), P(mp["pos3d"]), P(np.ascontiguousarray(mp["octave"], np.int32)), P(mp["desc"]), minDescDist, maxRepjDist,
          P(best_kp), P(best_d), P(mout))
    assert k >= 0, k
    dm = np.zeros(k, dtype=np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]))
    dm[:] = mout[:k].copy().view(dm.dtype).reshape(-1)
    return dict(best_kp=best_kp, best_dist=best_d, matches=dm)


# ------------------------------------------------------------------------------------------------ hierarchical k-means (a14)
def hkmeans_blob(L, train, k=32, max_iters=0):
    import numpy as np

    f = L.oracle_hkmeans_blob
    f.restype = C.c_long
    f.argtypes = [VP, I, I, I, VP, C.c_long]
    n = f(P(train), len(train), k, max_iters, None, 0)
    if n < 0:
        return int(n)
    out = np.zeros(n, np.uint8)
    f(P(train), len(train), k, max_iters, P(out), n)
    return out


def hkmeans_search(L, blob, queries, nn, max_checks, sorted_=0):
    import numpy as np

    nq = len(queries)
    idx = np.empty((nq, nn), np.int32)
    dist = np.empty((nq, nn), np.int32)
    f = L.oracle_hkmeans_search
    f.restype = I
    f.argtypes = [VP, VP, I, I, I, I, VP, VP]
    rc = f(P(blob), P(queries), nq, nn, max_checks, int(sorted_), P(idx), P(dist))
    assert rc == 0
    return idx, dist


def ref_hkmeans_stream(R, train, k=32, max_iters=0):
    import numpy as np

    f = R.xflann_ref_hkmeans_stream
    f.restype = C.c_long
    f.argtypes = [VP, I, I, I, VP, C.c_long]
    n = f(P(train), len(train), k, max_iters, None, 0)
    assert n > 0
    out = np.zeros(n, np.uint8)
    f(P(train), len(train), k, max_iters, P(out), n)
    return out


def ref_hkmeans_search(R, train, queries, nn, k, max_iters, max_checks, sorted_=0):
    import numpy as np

    nq = len(queries)
    idx = np.empty((nq, nn), np.int32)
    dist = np.empty((nq, nn), np.int32)
    f = R.xflann_ref_hkmeans_search
    f.restype = I
    f.argtypes = [VP, I, VP, I, I, I, I, I, I, VP, VP]
    rc = f(P(train), len(train), P(queries), nq, nn, k, max_iters, max_checks, int(sorted_), P(idx), P(dist))
    assert rc == 0
    return idx, dist
