"""Host-side mirror of the reference BA plugin on top of the C ABI.

Reference surface: `GlobalOptimizer` (src/optimization/globaloptimizer.h:28-68): `create(type)`,
`setParams(map, ParamSet)`, `optimize(bool* stopASAP)`, `getResults(map)`, `getBadAssociations()`, `getName()`.
The map is passed already flattened (dict of arrays, see include/ucoslam_hip.h uh_ba_problem).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import I, VP, check, lib, np_ptr


class _Problem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("poses_f2g", VP), ("fixed", VP),
                ("intr", VP), ("points", VP), ("obs_point", VP), ("obs_frame", VP), ("obs_uv", VP), ("obs_inv_sigma", VP)]


class ParamSet(C.Structure):
    """The subset of GlobalOptimizer::ParamSet that the monocular path uses (globaloptimizer.h:31-47)."""
    _fields_ = [("n_iters", C.c_int32), ("huber_delta", C.c_double), ("chi2_threshold", C.c_double),
                ("min_chi2_between_iter", C.c_float)]

    def __init__(self, nIters=5, huber_delta=0.0, chi2_threshold=0.0, min_chi2_between_iter=1.0):
        super().__init__(nIters, huber_delta, chi2_threshold, min_chi2_between_iter)


def _declare(L, sig):
    sig("uh_ba_create", I, VP, C.POINTER(VP))
    sig("uh_ba_destroy", None, VP)
    sig("uh_ba_set_problem", I, VP, C.POINTER(_Problem), C.POINTER(ParamSet))
    sig("uh_ba_optimize", I, VP, VP)
    sig("uh_ba_optimize_async", I, VP, VP)
    sig("uh_ba_wait", I, VP)
    sig("uh_ba_stop_flag", VP, VP)
    sig("uh_ba_get_results", I, VP, VP, VP, VP, VP, VP)
    sig("uh_ba_get_pose_state", I, VP, VP)


_lib._EXTRA_DECLS.append(_declare)


class GlobalOptimizer:
    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_ba_create(ctx.handle, C.byref(self._h)))
        self._dims = None
        self._bad = None

    @staticmethod
    def create(ctx: _lib.Context, type: str = "hip") -> "GlobalOptimizer":
        if type not in ("", "hip"):                       # globaloptimizer.cpp:27-33: unknown type throws
            raise RuntimeError("GlobalOptimizer::create: invalid optimizer type " + type)
        return GlobalOptimizer(ctx)

    def getName(self) -> str:
        return "hip"

    def setParams(self, problem: dict, params: ParamSet | None = None):
        a = {k: np.ascontiguousarray(problem[k]) for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w")}
        assert a["poses"].dtype == np.float32 and a["intr"].dtype == np.float32 and a["points"].dtype == np.float32
        assert a["obs_pt"].dtype == np.int32 and a["obs_kf"].dtype == np.int32 and a["obs_uv"].dtype == np.float32
        assert a["obs_w"].dtype == np.float64 and a["fixed"].dtype == np.uint8
        K, P, E = len(a["fixed"]), len(a["points"]), len(a["obs_pt"])
        pr = _Problem(K, P, E, np_ptr(a["poses"]), np_ptr(a["fixed"]), np_ptr(a["intr"]), np_ptr(a["points"]), np_ptr(a["obs_pt"]),
                      np_ptr(a["obs_kf"]), np_ptr(a["obs_uv"]), np_ptr(a["obs_w"]))
        check(lib().uh_ba_set_problem(self._h, C.byref(pr), C.byref(params) if params is not None else None))
        self._dims = (K, P, E)
        self._obs = (a["obs_pt"].copy(), a["obs_kf"].copy())

    def optimize(self, stop_asap: np.ndarray | None = None):
        check(lib().uh_ba_optimize(self._h, np_ptr(stop_asap) if stop_asap is not None else None))

    def optimize_async(self, stop_asap: np.ndarray | None = None):
        """Starts optimize() on the object's worker thread (the reference's mapper thread); wait() returns when it is done."""
        self._stop_keep = stop_asap
        check(lib().uh_ba_optimize_async(self._h, np_ptr(stop_asap) if stop_asap is not None else None))

    def wait(self):
        check(lib().uh_ba_wait(self._h))

    def getResults(self):
        K, P, E = self._dims
        out = dict(poses=np.zeros((K, 16), np.float32), points=np.zeros((P, 3), np.float32), chi2=np.zeros(E, np.float64),
                   bad=np.zeros(E, np.uint8), iters=np.zeros(2, np.int32), state=np.zeros((K, 7), np.float64))
        check(lib().uh_ba_get_results(self._h, np_ptr(out["poses"]), np_ptr(out["points"]), np_ptr(out["chi2"]), np_ptr(out["bad"]),
                                      np_ptr(out["iters"])))
        check(lib().uh_ba_get_pose_state(self._h, np_ptr(out["state"])))
        self._bad = [(int(p), int(f)) for p, f, b in zip(self._obs[0], self._obs[1], out["bad"]) if b]
        return out

    def getBadAssociations(self):
        """vector<pair<point id, frame id>> (globaloptimizer.h:60)."""
        return self._bad or []

    def close(self):
        if self._h:
            lib().uh_ba_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
