"""FrameMatcher filter chain: product host code (uh_match_filter) vs an independent Python restatement + known answers.
The filter is host logic, so these run without a GPU; the end-to-end matcher (kNN on the GPU + filter) is a gpu test."""
import numpy as np
import pytest

import oracle_lib
import py_oracle_matcher as pyo
import synth


def _frame(n, rng, desc=None):
    return dict(desc=desc if desc is not None else rng.integers(0, 256, (n, 32), dtype=np.uint8),
                ids=np.where(rng.random(n) < 0.5, rng.integers(0, 1000, n), 0xFFFFFFFF).astype(np.uint32),
                nonmaxima=rng.random(n) < 0.05, octave=rng.integers(0, 8, n).astype(np.int32),
                angle=(rng.random(n) * 360).astype(np.float32), pt=(rng.random((n, 2)) * 600).astype(np.float32),
                scaleFactors=(1.2 ** np.arange(8)).astype(np.float32))


def _as_tuples(ms):
    return [(int(m["queryIdx"]), int(m["trainIdx"]), float(m["distance"])) for m in ms]


def test_filter_matches_independent_restatement(oracle):
    from ucoslam_cv3_amd.matcher import match_filter

    rng = np.random.default_rng(0)
    for trial in range(15):
        nq, nt, nn = int(rng.integers(1, 120)), int(rng.integers(1, 300)), int(rng.choice([1, 2, 10]))
        train, q = synth.match_set(nq, nt, seed=trial)
        tf, qf = _frame(nt, rng, train), _frame(nq, rng, q)
        if trial % 3 == 0:      # correlated angles/octaves so that the orientation histogram has structure
            src = rng.integers(0, nt, nq)
            qf["angle"] = ((tf["angle"][src] + 17 + rng.normal(0, 3, nq)) % 360).astype(np.float32)
            qf["octave"] = tf["octave"][src]
        idx, dist = oracle_lib.knn_search(oracle, train, q, nn, 0)
        for mdd, ratio, co, mod in [(100.0, 0.6, True, 3), (60.0, 0.8, False, 1), (np.inf, 0.8, True, 1)]:
            got = match_filter(idx, dist, qf, tf, None, None, mdd, ratio, co, mod)
            ref = pyo.match_filter(idx, dist, qf, tf, None, None, min(mdd, np.finfo(np.float32).max), ratio, co, mod)
            assert _as_tuples(got) == [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in ref]


def test_filter_known_answers():
    from ucoslam_cv3_amd.matcher import match_filter

    z = np.zeros
    q = dict(octave=z(3, np.int32), angle=z(3, np.float32), pt=z((3, 2), np.float32), scaleFactors=np.ones(8, np.float32))
    t = dict(octave=z(4, np.int32), angle=z(4, np.float32), pt=z((4, 2), np.float32))
    # row order matters: a better candidate seen AFTER the current best does not demote it into second-best (Appendix B)
    idx = np.array([[0, 1], [1, 0], [2, 3]], np.int32)
    dist = np.array([[30, 40], [40, 30], [90, 95]], np.int32)
    got = match_filter(idx, dist, q, t, min_desc_dist=100.0, nn_match_ratio=0.6, check_orientation=False)
    # query 0: best 30 (t0), runner-up 40 same octave: 30 > 40*0.6=24 -> rejected by the ratio test
    # query 1: sees 40 first (best), then 30 improves best; bestDist2 never set -> accepted with t0 at 30
    # query 2: 90 then 95: 90 > 95*0.6 -> rejected
    assert _as_tuples(got) == [(1, 0, 30.0)]
    # ambiguous train: two queries on the same train index keep the smaller distance, first wins ties
    idx = np.array([[0], [0], [0]], np.int32)
    dist = np.array([[50], [20], [20]], np.int32)
    got = match_filter(idx, dist, q, t, min_desc_dist=100.0, check_orientation=False)
    assert _as_tuples(got) == [(1, 0, 20.0)]
    # unfilled slot (-1, 0) of a short index row is ignored
    idx = np.array([[-1, 2]], np.int32)
    dist = np.array([[0, 10]], np.int32)
    got = match_filter(idx, dist, dict(q, octave=z(1, np.int32), angle=z(1, np.float32), pt=z((1, 2), np.float32)), t,
                       min_desc_dist=100.0, check_orientation=False)
    assert _as_tuples(got) == [(0, 2, 10.0)]


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True], ids=["hkmeans32_maxchecks16", "exact_scan"])
def test_hip_frame_matcher_end_to_end(hip_ctx, oracle, exact):
    """Index search on the GPU (unsorted heap rows) + filter == oracle search + independent filter, with the ASSIGNED/UNASSIGNED
    modes.  Default = the reference's own index (HKMeansParams(32,0), maxChecks=16); exact=True = brute-force scan."""
    from ucoslam_cv3_amd import matcher as M

    rng = np.random.default_rng(5)
    train, q = synth.match_set(400, 1500, seed=9)
    tf, qf = _frame(1500, rng, train), _frame(400, rng, q)
    fm = M.FrameMatcher(hip_ctx, exact=exact)
    for tmode, qmode in [(M.MODE_ALL, M.MODE_ALL), (M.MODE_ASSIGNED, M.MODE_ALL), (M.MODE_ALL, M.MODE_UNASSIGNED)]:
        fm.setParams(tf, tmode, 100.0, 0.6, True, 3)
        got = fm.match(qf, qmode)
        map_t, tdesc = M.manage_mode(tmode, tf)
        map_q, qdesc = M.manage_mode(qmode, qf)
        if exact:
            idx, dist = oracle_lib.knn_search(oracle, tdesc, qdesc, 10, 0)
        else:
            idx, dist = oracle_lib.hkmeans_search(oracle, oracle_lib.hkmeans_blob(oracle, tdesc, 32, 0), qdesc, 10, 16, 0)
        ref = pyo.match_filter(idx, dist, qf, tf, map_q, map_t, 100.0, 0.6, True, 3)
        assert _as_tuples(got) == [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in ref]
        assert len(got) > 10
