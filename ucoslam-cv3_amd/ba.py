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


class _Staging(C.Structure):
    _fields_ = [("poses_f2g", VP), ("fixed", VP), ("intr", VP), ("points", VP), ("obs", VP),
                ("cap_frames", C.c_int32), ("cap_points", C.c_int32), ("cap_obs", C.c_int32)]


class _ResultsView(C.Structure):
    _fields_ = [("poses", VP), ("points", VP), ("chi2", VP), ("bad", VP), ("pose_state", VP), ("iters", C.c_int32 * 2),
                ("n_frames", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32)]


OBS_DTYPE = np.dtype([("point", np.int32), ("frame", np.int32), ("u", np.float32), ("v", np.float32), ("inv_sigma", np.float64)])   # uh_ba_obs, 24 B


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
    sig("uh_ba_map_staging", I, VP, I, I, I, C.POINTER(_Staging))
    sig("uh_ba_set_problem_staged", I, VP, I, I, I, C.POINTER(ParamSet))
    sig("uh_ba_results_view_get", I, VP, C.POINTER(_ResultsView))
    sig("uh_ba_solve_async", I, VP, C.POINTER(_Problem), I, I, I, C.POINTER(ParamSet), VP)
    sig("uh_ba_form", I, VP, C.POINTER(C.c_int))
    sig("uh_ba_want_chi2", I, VP, I)


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

    def _problem_struct(self, problem: dict):
        a = {k: np.ascontiguousarray(problem[k]) for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w")}
        assert a["poses"].dtype == np.float32 and a["intr"].dtype == np.float32 and a["points"].dtype == np.float32
        assert a["obs_pt"].dtype == np.int32 and a["obs_kf"].dtype == np.int32 and a["obs_uv"].dtype == np.float32
        assert a["obs_w"].dtype == np.float64 and a["fixed"].dtype == np.uint8
        K, P, E = len(a["fixed"]), len(a["points"]), len(a["obs_pt"])
        pr = _Problem(K, P, E, np_ptr(a["poses"]), np_ptr(a["fixed"]), np_ptr(a["intr"]), np_ptr(a["points"]), np_ptr(a["obs_pt"]),
                      np_ptr(a["obs_kf"]), np_ptr(a["obs_uv"]), np_ptr(a["obs_w"]))
        return pr, a, (K, P, E)

    def setParams(self, problem, params: ParamSet | None = None):
        pr, a, dims = problem if isinstance(problem, tuple) else self._problem_struct(problem)
        check(lib().uh_ba_set_problem(self._h, C.byref(pr), C.byref(params) if params is not None else None))
        self._dims = dims
        self._obs = (a["obs_pt"], a["obs_kf"])
        self._bad = None

    # ---- the staged form: the flattened problem is written straight into the optimiser's pinned staging block
    def mapStaging(self, K: int, P: int, max_obs: int) -> dict:
        """numpy views of the staging block (uh_ba_map_staging): poses [K,16] f32, fixed [K] u8, intr [K,4] f32, points [P,3] f32,
        obs [max_obs] records (OBS_DTYPE).  Valid until the next mapStaging / setParams on this object."""
        st = _Staging()
        check(lib().uh_ba_map_staging(self._h, K, P, max_obs, C.byref(st)))

        def view(ptr, dtype, shape):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * n).from_address(ptr), dtype=dtype).reshape(shape)

        return dict(poses=view(st.poses_f2g, np.float32, (K, 16)), fixed=view(st.fixed, np.uint8, (K,)), intr=view(st.intr, np.float32, (K, 4)),
                    points=view(st.points, np.float32, (P, 3)), obs=view(st.obs, OBS_DTYPE, (max_obs,)))

    def fillStaging(self, problem: dict) -> tuple:
        """Copies a flattened problem dict into the staging block (what flatten_for_ba does in C++ while it walks the map)."""
        K, P, E = len(problem["fixed"]), len(problem["points"]), len(problem["obs_pt"])
        m = self.mapStaging(K, P, E)
        m["poses"][:] = problem["poses"].reshape(K, 16); m["fixed"][:] = problem["fixed"]; m["intr"][:] = problem["intr"]
        m["points"][:] = problem["points"]
        o = m["obs"]
        o["point"] = problem["obs_pt"]; o["frame"] = problem["obs_kf"]; o["u"] = problem["obs_uv"][:, 0]; o["v"] = problem["obs_uv"][:, 1]
        o["inv_sigma"] = problem["obs_w"]
        self._obs = (np.array(problem["obs_pt"], np.int32), np.array(problem["obs_kf"], np.int32))
        return K, P, E

    def setParamsStaged(self, K: int, P: int, E: int, params: ParamSet | None = None):
        check(lib().uh_ba_set_problem_staged(self._h, K, P, E, C.byref(params) if params is not None else None))
        self._dims = (K, P, E)
        self._bad = None

    def solve_async(self, problem: dict | None, params: ParamSet | None = None, dims=None, stop_asap: np.ndarray | None = None):
        """setParams + optimize on the object's worker thread (the reference's mapper thread runs exactly these two calls back to back,
        mapmanager.cpp:11388-11405); `problem=None`: the staging block with `dims = (K, P, E)`.  wait() returns when both are done."""
        self._stop_keep = stop_asap
        self._params_keep = params
        if problem is not None:
            pr, a, dims = problem if isinstance(problem, tuple) else self._problem_struct(problem)   # (a tuple: prepareProblem()'s result)
            self._keep = (pr, a)
            self._obs = (a["obs_pt"], a["obs_kf"])
            check(lib().uh_ba_solve_async(self._h, C.byref(pr), 0, 0, 0, C.byref(params) if params is not None else None,
                                          np_ptr(stop_asap) if stop_asap is not None else None))
        else:
            check(lib().uh_ba_solve_async(self._h, None, dims[0], dims[1], dims[2], C.byref(params) if params is not None else None,
                                          np_ptr(stop_asap) if stop_asap is not None else None))
        self._dims = tuple(dims)
        self._bad = None

    def prepareProblem(self, problem: dict):
        """The ctypes view of a flattened problem, built once (solve_async / setParams accept it in place of the dict)."""
        return self._problem_struct(problem)

    def wantChi2(self, on: bool):
        """The per-observation chi2 is an extra of this ABI; switching it off (before setParams) shortens the kernel's result hand-over."""
        check(lib().uh_ba_want_chi2(self._h, int(on)))
        return self

    def form(self) -> str:
        """'persist<NF>' (one persistent launch, NF lanes per landmark), 'chain' (launch chain) or 'wide' (global BA)."""
        lanes = C.c_int(0)
        f = lib().uh_ba_form(self._h, C.byref(lanes))
        if f < 0:
            check(f)
        return {0: "chain", 1: f"persist{lanes.value}", 2: "wide"}[f]

    def resultsView(self) -> dict:
        """getResults in place (uh_ba_results_view_get): numpy views of the pinned result block, valid until the next setParams."""
        v = _ResultsView()
        check(lib().uh_ba_results_view_get(self._h, C.byref(v)))
        K, P, E = v.n_frames, v.n_points, v.n_obs

        def view(ptr, dtype, shape):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * max(n, 1)).from_address(ptr), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

        return dict(poses=view(v.poses, np.float32, (K, 16)), points=view(v.points, np.float32, (P, 3)), chi2=view(v.chi2, np.float64, (E,)),
                    bad=view(v.bad, np.uint8, (E,)), state=view(v.pose_state, np.float64, (K, 7)), iters=np.array(list(v.iters), np.int32))

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
        self._bad_flags = out["bad"]
        self._bad = None
        return out

    def getBadAssociations(self):
        """vector<pair<point id, frame id>> (globaloptimizer.h:60)."""
        if self._bad is None and getattr(self, "_bad_flags", None) is not None:
            sel = np.flatnonzero(self._bad_flags)
            self._bad = [(int(p), int(f)) for p, f in zip(self._obs[0][sel], self._obs[1][sel])]
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
