"""Host-side mirror of ucoslam::PnPSolver::solvePnp (monocular matches) on top of the C ABI.

Reference: src/optimization/pnpsolver.h:30-38 / pnpsolver.cpp:116-409: `solvePnp(frame, map, matches, pose)` refines `pose`
in place, marks outlier matches (DMatch::imgIdx = -1, inliers = 1) and returns the number of inliers.
Here the frame/map lookups are already done: the caller passes, per match, the map point, the undistorted keypoint,
1/scaleFactor[octave] and the stability weight (1, or 0.5 for MapPoint::isStable() == false).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import I, VP, check, lib, np_ptr


def _declare(L, sig):
    sig("uh_pnp_create", I, VP, C.POINTER(VP))
    sig("uh_pnp_destroy", None, VP)
    sig("uh_pnp_solve", I, VP, VP, VP, I, VP, VP, VP, VP, VP, VP, VP, VP)
    sig("uh_pnp_solve_dev", I, VP, VP, VP, I, VP, VP, VP, VP, VP, VP, VP, VP, VP)
    sig("uh_pnp_debug_clocks", I, VP, I, VP)


_lib._EXTRA_DECLS.append(_declare)


class PnPSolver:
    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self._h = VP()
        check(lib().uh_pnp_create(ctx.handle, C.byref(self._h)))

    def solvePnp(self, pose_f2g, intr, p3d, kp, inv_sigma, weight):
        """Returns dict(pose [16] float32, bad [n] uint8, iters [4], state [7] fp64, ngood)."""
        a = [np.ascontiguousarray(x, np.float32) for x in (pose_f2g, intr, p3d, kp, inv_sigma, weight)]
        n = len(a[4])
        out = dict(pose=np.zeros(16, np.float32), bad=np.zeros(max(n, 1), np.uint8), iters=np.zeros(4, np.int32), state=np.zeros(7, np.float64))
        rc = lib().uh_pnp_solve(self._h, np_ptr(a[0]), np_ptr(a[1]), n, np_ptr(a[2]), np_ptr(a[3]), np_ptr(a[4]), np_ptr(a[5]), np_ptr(out["pose"]),
                                np_ptr(out["bad"]), np_ptr(out["iters"]), np_ptr(out["state"]))
        if rc < 0:
            check(rc)
        out["ngood"] = rc
        out["bad"] = out["bad"][:n]
        return out

    def debug_clocks(self, on=True):
        """Shader-clock stamps of the last solve (measurement hook): [entry, staged, rounds done, posted, passes, ...]."""
        out = np.zeros(512, np.int64)
        check(lib().uh_pnp_debug_clocks(self._h, int(on), np_ptr(out)))
        return out

    def close(self):
        if self._h:
            lib().uh_pnp_destroy(self._h)
            self._h = VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
