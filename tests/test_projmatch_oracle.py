"""Pins the oracle's restated kd-tree (oracle/proj_oracle.cpp) against the REAL reference kd-tree
(src/basictypes/picoflann.h compiled into oracle/_ref/libpicoflann_ref.so): same hits, same ORDER, same squared distances —
the order is observable through Map::matchFrameToMapPoints' best / second-best bookkeeping (map.cpp:727-741)."""
import numpy as np
import pytest

import oracle_lib


def _clouds():
    rng = np.random.default_rng(7)
    out = {}
    out["empty"] = np.zeros((0, 2), np.float32)
    for n in (1, 5, 10, 11, 21, 100, 777):
        out[f"uniform{n}"] = (rng.random((n, 2)) * [1241, 376]).astype(np.float32)
    out["image2000"] = (rng.random((2000, 2)) * [1241, 376]).astype(np.float32)
    out["image4000"] = (rng.random((4000, 2)) * [640, 480]).astype(np.float32)
    # integer pixel coordinates x level scale: many equal coordinates -> planeSplit ties and the std::sort fallback
    g = rng.integers(0, 60, (1500, 2)).astype(np.float32) * np.float32(1.2) ** rng.integers(0, 3, (1500, 1)).astype(np.float32)
    out["ties"] = g.astype(np.float32)
    out["all_equal"] = np.full((64, 2), 17.5, np.float32)
    out["column"] = np.stack([np.full(300, 33.0, np.float32), rng.random(300).astype(np.float32) * 400], 1)
    c = rng.normal([600, 180], [3, 3], (500, 2)).astype(np.float32)   # one tight cluster + sparse background
    out["cluster"] = np.concatenate([c, (rng.random((200, 2)) * [1241, 376]).astype(np.float32)])
    return out


@pytest.mark.parametrize("name", list(_clouds().keys()))
def test_restated_kdtree_matches_real_picoflann(oracle, name):
    ref = oracle_lib.load_ref("picoflann")
    if ref is None:
        pytest.skip("oracle/_ref/libpicoflann_ref.so not built (reference tree absent)")
    xy = _clouds()[name]
    if len(xy) == 0:
        pytest.skip("the reference indexes an empty node vector for an empty cloud; the restatement returns no hits")
    mine = oracle_lib.KdOracle(oracle, "oracle_kd", xy)
    real = oracle_lib.KdOracle(ref, "picoflann_ref", xy)
    rng = np.random.default_rng(11)
    lo, hi = xy.min(0) - 30, xy.max(0) + 30
    nhits = 0
    for t in range(400):
        if t % 3 == 0 and len(xy):   # queries sitting exactly on a data point / on split values
            q = xy[rng.integers(len(xy))]
        else:
            q = (lo + rng.random(2) * (hi - lo)).astype(np.float32)
        r = [3.0, 15.0, 15.0 * 1.2 ** 3 * 1.6, 80.0, 1e4][t % 5]
        i0, d0 = mine.radius(q[0], q[1], np.float32(r))
        i1, d1 = real.radius(q[0], q[1], np.float32(r))
        assert i0.tolist() == i1.tolist(), (name, t)
        assert d0.tolist() == d1.tolist()
        nhits += len(i0)
        # the hit SET is the brute-force disc (strict <, float difference, double square)
        dx = (np.float32(q[0]) - xy[:, 0]).astype(np.float64)
        dy = (np.float32(q[1]) - xy[:, 1]).astype(np.float64)
        inside = np.nonzero(dx * dx + dy * dy < float(np.float32(r)) ** 2)[0]
        assert sorted(i0.tolist()) == inside.tolist()
    assert nhits > 0


GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "kdtree_golden.npz")


@pytest.mark.parametrize("case", ["uniform", "ties", "small"])
def test_restated_kdtree_matches_committed_golden(oracle, case):
    """Radius-search hits, ORDER and squared distances recorded from the real picoflann.h (tests/golden/make_kdtree_golden.py):
    pins the restatement where /root/reference and oracle/_ref do not exist."""
    g = np.load(GOLD)
    xy = g[f"{case}_xy"]
    kd = oracle_lib.KdOracle(oracle, "oracle_kd", xy)
    off = g[f"{case}_off"]
    for t, (q, r) in enumerate(zip(g[f"{case}_q"], g[f"{case}_r"])):
        i, d = kd.radius(q[0], q[1], r)
        assert i.tolist() == g[f"{case}_idx"][off[t]:off[t + 1]].tolist(), (case, t)
        assert d.tolist() == g[f"{case}_sqd"][off[t]:off[t + 1]].tolist()
    assert off[-1] > 100
