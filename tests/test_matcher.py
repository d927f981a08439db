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


def _bow_frames(nq, nt, seed, n_nodes=120, low_entropy=False):
    """Two frames whose features share vocabulary nodes: each query feature is a noisy copy of a train feature and mostly lands in
    the same node (as a real vocabulary would put it), some land elsewhere; node lists in feature order like fBow2's push_back."""
    rng = np.random.default_rng(seed)
    train, q = synth.match_set(nq, nt, seed=seed)
    if low_entropy:                                   # many equal distances: the "last not better" rule becomes order sensitive
        train[:, 2:] = 0
        q[:, 2:] = 0
    tf, qf = _frame(nt, rng, train), _frame(nq, rng, q)
    src = rng.integers(0, nt, nq)
    qf["octave"] = np.clip(tf["octave"][src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    qf["angle"] = ((tf["angle"][src] + 25 + rng.normal(0, 4, nq)) % 360).astype(np.float32)
    t_node = rng.integers(0, n_nodes, nt) * 7 + 3      # sparse, unsorted-looking node ids
    q_node = np.where(rng.random(nq) < 0.8, t_node[src], rng.integers(0, n_nodes, nq) * 7 + 3)
    for f, nodes in ((tf, t_node), (qf, q_node)):
        bv = {}
        for i, nd in enumerate(nodes.tolist()):
            bv.setdefault(int(nd), []).append(i)
        f["bowvector_level"] = bv
    return qf, tf


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(400, 1500, 1, False), (2000, 2000, 2, False), (700, 900, 3, True), (1, 1, 4, False), (50, 3000, 5, True)],
                         ids=lambda c: f"q{c[0]}_t{c[1]}{'_ties' if c[3] else ''}")
def test_hip_bow_matcher_matches_independent_restatement(hip_ctx, cfg):
    """FrameMatcher (TYPE_BOW): GPU distances + bookkeeping and the shared host tail vs the pure-Python restatement, all modes,
    with and without the epipolar gate."""
    from ucoslam_cv3_amd import matcher as M

    nq, nt, seed, le = cfg
    qf, tf = _bow_frames(nq, nt, seed, low_entropy=le)
    fm = M.FrameMatcherBoW(hip_ctx)
    rng = np.random.default_rng(seed)
    F12 = (rng.normal(0, 1, 9) * [1e-6, 1e-5, 1e-3, 1e-5, 1e-6, 1e-3, 1e-3, 1e-3, 1]).astype(np.float32)
    total = 0
    for tmode, qmode, params, F in [(M.MODE_ALL, M.MODE_ALL, (100.0 if not le else 6.0, 0.6, True, 3), None),
                                    (M.MODE_ASSIGNED, M.MODE_ALL, (60.0 if not le else 3.0, 0.8, False, 1), None),
                                    (M.MODE_ALL, M.MODE_UNASSIGNED, (np.inf, 0.8, True, 1), None),
                                    (M.MODE_ALL, M.MODE_ALL, (100.0 if not le else 6.0, 0.9, True, 2), F12)]:
        fm.setParams(tf, tmode, *params)
        got = fm.matchEpipolar(qf, qmode, F)
        ref = pyo.bow_match(qf, tf, M.is_used(qf, qmode), M.is_used(tf, tmode), min(params[0], np.finfo(np.float32).max), *params[1:], F12=F)
        assert _as_tuples(got) == [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in ref]
        total += len(got)
    if nq >= 400:
        assert total > 30


@pytest.mark.gpu
def test_hip_bow_matcher_known_answers_and_errors(hip_ctx):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd import matcher as M

    z = np.zeros
    d = np.zeros((4, 32), np.uint8)
    d[1, 0] = 0b111          # distance 3 from descriptor 0
    d[2, 0] = 0b1            # distance 1
    d[3, 0] = 0b11           # distance 2
    base = dict(ids=np.full(4, 0xFFFFFFFF, np.uint32), nonmaxima=z(4, bool), octave=z(4, np.int32), angle=z(4, np.float32),
                pt=z((4, 2), np.float32), scaleFactors=np.ones(8, np.float32))
    q = dict(base, desc=d[:1].copy(), ids=base["ids"][:1], nonmaxima=z(1, bool), octave=z(1, np.int32), angle=z(1, np.float32),
             pt=z((1, 2), np.float32), bowvector_level={5: [0]})
    fm = M.FrameMatcherBoW(hip_ctx)
    # candidates at distances 3, 1, 2 in this order: best ends at 1; "second" is the LAST not-better one (2), not the smallest
    # (3 was overwritten): 1 > 2 * 0.4 -> rejected; with the list order 2, 1, 3 the last not-better is 3: 1 > 3 * 0.4 is false -> accepted
    t = dict(base, desc=d.copy(), bowvector_level={5: [1, 2, 3]})
    fm.setParams(t, M.MODE_ALL, 100.0, 0.4, False, 1)
    assert _as_tuples(fm.match(q)) == []
    t["bowvector_level"] = {5: [3, 2, 1]}
    assert _as_tuples(fm.match(q)) == [(0, 2, 1.0)]
    # no common node -> nothing; a node with an empty list is refused
    t["bowvector_level"] = {6: [1, 2, 3]}
    assert _as_tuples(fm.match(q)) == []
    t["bowvector_level"] = {5: []}
    with pytest.raises(u.UcoslamHipError):
        fm.match(q)


@pytest.mark.gpu
def test_hip_bow_matcher_end_to_end_with_vocabulary(hip_ctx):
    """The chain the reference runs (keyframedatabase.cpp:319 -> FrameMatcher_BoW): Vocabulary::transform(desc, 3) on the GPU gives
    each frame's fBow2, the BoW matcher consumes them; compared with the Python restatement fed the same maps."""
    from ucoslam_cv3_amd import matcher as M
    from ucoslam_cv3_amd.bow import Vocabulary, write_vocabulary_stream

    params, blob, meta = synth.vocabulary(k=10, depth=4, seed=3, aligment=8)
    voc = Vocabulary(hip_ctx).fromStream(write_vocabulary_stream(params, blob))
    rng = np.random.default_rng(12)
    nt, nq = 1500, 1200
    train = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    src = rng.integers(0, nt, nq)
    q = train[src] ^ np.packbits(rng.random((nq, 256)) < 0.03, axis=1, bitorder="little")
    tf, qf = _frame(nt, rng, train), _frame(nq, rng, q)
    qf["octave"] = tf["octave"][src]
    qf["angle"] = ((tf["angle"][src] + 40 + rng.normal(0, 3, nq)) % 360).astype(np.float32)
    for f in (tf, qf):
        f["bowvector_level"] = voc.transform(f["desc"], 3)[1]
    fm = M.FrameMatcherBoW(hip_ctx)
    fm.setParams(tf, M.MODE_ALL, 100.0, 0.6, True, 1)
    got = fm.match(qf, M.MODE_ALL)
    ref = pyo.bow_match(qf, tf, M.is_used(qf, M.MODE_ALL), M.is_used(tf, M.MODE_ALL), 100.0, 0.6, True, 1)
    assert _as_tuples(got) == [(m["queryIdx"], m["trainIdx"], m["distance"]) for m in ref]
    right = sum(1 for m in got if src[m["queryIdx"]] == m["trainIdx"])
    assert len(got) > 100 and right > 0.8 * len(got)
