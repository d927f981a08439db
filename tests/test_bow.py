"""fbow stage: oracle KATs (CPU) and HIP vs oracle (gpu)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth

P = oracle_lib.P


def _oracle_transform(L, params, blob, desc, level):
    n = len(desc)
    pb = np.frombuffer(params, np.uint8)
    bb = np.frombuffer(blob, np.uint8)
    word, weight = np.empty(n, np.uint32), np.empty(n, np.float32)
    node, valid = np.empty(n, np.uint32), np.empty(n, np.uint8)
    L.oracle_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_int] + [C.c_void_p] * 4
    assert L.oracle_bow_transform(P(pb), P(bb), P(desc), n, desc.strides[0], level, P(word), P(weight), P(node), P(valid)) == 0
    return word, weight, node, valid


def test_params_sizes_match_setparams_formulas(oracle):
    params, blob, meta = synth.vocabulary(k=10, depth=3, aligment=8)
    out = np.zeros(120, np.uint8)
    oracle.oracle_bow_make_params(8, 10, 32, meta["nblocks"], P(out))
    assert out.tobytes() == params                      # writer and the restated setParams (fbow.cpp:10-49) agree
    assert len(blob) == meta["block_size"] * meta["nblocks"]
    params32, _, meta32 = synth.vocabulary(k=9, depth=2, aligment=32)
    out = np.zeros(120, np.uint8)
    oracle.oracle_bow_make_params(32, 9, 32, meta32["nblocks"], P(out))
    assert out.tobytes() == params32


def test_oracle_descent_known_answers(oracle):
    """Hand-checkable: a query equal to a leaf's own descriptor chain must land in that leaf; the node id packs the child
    indices 4 bits per level (k=10 -> ceil(log2 10) = 4)."""
    import struct

    params, blob, meta = synth.vocabulary(k=10, depth=3, seed=3)
    f = struct.unpack("<50s2xII4xQQQQQiiI4x", params)
    bs, fo, co, dwp = f[4], f[5], f[6], f[3]
    b = np.frombuffer(blob, np.uint8)
    # follow children (2, 5, 7) by hand
    path, block = (2, 5, 7), 0
    for lvl, c in enumerate(path):
        info = block * bs + co + c * 8
        idc, w = struct.unpack("<If", b[info:info + 8].tobytes())
        if lvl < 2:
            assert not idc & 0x80000000
            last_desc = b[block * bs + fo + c * dwp: block * bs + fo + c * dwp + 32]
            block = idc
        else:
            assert idc & 0x80000000
            leaf_desc = b[block * bs + fo + c * dwp: block * bs + fo + c * dwp + 32].copy()
            leaf_word, leaf_w = idc & 0x7FFFFFFF, w
    word, weight, node, valid = _oracle_transform(oracle, params, blob, leaf_desc[None, :].copy(), 2)
    # the leaf's descriptor is a noisy copy of its ancestors', so the greedy descent follows the same path
    assert word[0] == leaf_word and weight[0] == np.float32(leaf_w)
    assert valid[0] == 1 and node[0] == (2 << 4) | 5
    word, weight, node, valid = _oracle_transform(oracle, params, blob, leaf_desc[None, :].copy(), 7)   # level beyond the leaf
    assert valid[0] == 1 and node[0] == (2 << 4) | 5                                                    # stored at the leaf's level


def test_oracle_score(oracle):
    oracle.oracle_bow_score.restype = C.c_double
    a_ids, a_w = np.array([1, 4, 9], np.uint32), np.array([0.6, 0.0, 0.8], np.float32)
    s = oracle.oracle_bow_score(P(a_ids), P(a_w), 3, P(a_ids), P(a_w), 3)
    assert abs(s - (1.0 - np.sqrt(1.0 - float(np.float32(0.6) * np.float32(0.6) + np.float32(0.8) * np.float32(0.8))))) < 1e-12 or s == 1.0
    b_ids, b_w = np.array([2, 5], np.uint32), np.array([1.0, 1.0], np.float32)
    assert oracle.oracle_bow_score(P(a_ids), P(a_w), 3, P(b_ids), P(b_w), 2) == 0.0      # disjoint supports


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(10, 3, 8), (10, 4, 8), (9, 2, 32), (3, 5, 8), (1, 3, 8)], ids=lambda c: f"k{c[0]}_L{c[1]}_al{c[2]}")
def test_hip_bow_transform_bit_exact(hip_ctx, oracle, cfg):
    from ucoslam_cv3_amd.bow import Vocabulary, fBow, write_vocabulary_stream

    k, depth, al = cfg
    params, blob, meta = synth.vocabulary(k=k, depth=depth, seed=11, aligment=al)
    voc = Vocabulary(hip_ctx).fromStream(write_vocabulary_stream(params, blob))
    assert voc.getK() == k and voc.getDescSize() == 32 and voc.getDescType() == 0
    rng = np.random.default_rng(2)
    # queries: noisy copies of random node descriptors (deep descents) + pure noise
    b = np.frombuffer(blob, np.uint8)
    desc = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    import struct
    f = struct.unpack("<50s2xII4xQQQQQiiI4x", params)
    for i in range(0, 500):
        blk, c = int(rng.integers(0, meta["nblocks"])), int(rng.integers(0, max(k - 2, 1)))
        src = b[blk * f[4] + f[5] + c * f[3]: blk * f[4] + f[5] + c * f[3] + 32]
        desc[i] = src ^ np.packbits(rng.random(256) < 0.05, bitorder="little")
    for level in (0, 1, 3, 9):
        word, weight, node, valid = voc._descend(desc, level)
        rw, rwt, rn, rv = _oracle_transform(oracle, params, blob, desc, level)
        np.testing.assert_array_equal(word, rw)
        np.testing.assert_array_equal(weight, rwt)
        np.testing.assert_array_equal(valid, rv)
        np.testing.assert_array_equal(node[valid == 1], rn[rv == 1])
    # API-level: maps, normalised transform, score
    r1, r2 = voc.transform(desc, 3)
    assert sum(len(v) for v in r2.values()) == int(rv.sum()) if level == 3 else True
    assert all(v == sorted(v) for v in r2.values())          # feature indices in feature order
    bag = voc.transform(desc)
    norm = sum(float(v) ** 2 for v in bag.values())
    assert abs(norm - 1.0) < 1e-5
    assert fBow.score(bag, bag) > 0.999
    other = voc.transform(desc[::-1][:300].copy())
    s = fBow.score(bag, other)
    assert 0.0 <= s <= 1.0


@pytest.mark.gpu
def test_hip_bow_errors(hip_ctx):
    from ucoslam_cv3_amd.bow import Vocabulary, write_vocabulary_stream

    params, blob, _ = synth.vocabulary()
    voc = Vocabulary(hip_ctx)
    import ucoslam_cv3_amd as u
    with pytest.raises(u.UcoslamHipError):                      # bad signature (fbow.cpp:184)
        voc.fromStream(b"\0" * 8 + params + blob)
    voc.fromStream(write_vocabulary_stream(params, blob))
    with pytest.raises(RuntimeError):                           # No input data (fbow.cpp:52)
        voc.transform(np.zeros((0, 32), np.uint8), 3)
    with pytest.raises(u.UcoslamHipError):                      # descriptor size mismatch (fbow.cpp:54)
        voc.transform(np.zeros((4, 61), np.uint8), 3)


def _py_reloc(db, query, covis, sorted_, min_score, excluded):
    """Independent restatement of KPFrameDataBase::relocalizationCandidates (keyframedatabase.cpp:195-275) on plain dicts:
    db = {frame id: fBow}, fBow = {word: float32}; the inverted word -> frames index is rebuilt like add() does."""
    word_frames = {}
    for f, bow in db.items():
        for w in bow:
            word_frames.setdefault(w, set()).add(f)
    nobs, max_common = {}, 0
    for w in sorted(query):
        for f in sorted(word_frames.get(w, ())):
            if f in excluded:
                continue
            nobs[f] = nobs.get(f, 0) + 1
            max_common = max(max_common, nobs[f])
    if not nobs:
        return [], []
    min_common = int(np.float32(max_common) * np.float32(0.8))
    frame_score = {}
    for f in sorted(nobs):
        if nobs[f] > min_common:
            s, a, b = 0.0, query, db[f]
            for w in sorted(set(a) & set(b)):
                s += float(np.float32(a[w]) * np.float32(b[w]))
            s = 1.0 if s >= 1 else 1.0 - np.sqrt(1.0 - s)
            if s > min_score:
                frame_score[f] = s
    scored = [(f, nobs[f], frame_score[f]) for f in sorted(frame_score)]
    if len(frame_score) == 0:
        return scored, []
    if len(frame_score) == 1:
        return scored, [next(iter(frame_score))]
    acc, best = [], float(np.float32(min_score))
    for f in sorted(frame_score):
        a = frame_score[f]
        for nb, _w in covis(f)[:10]:
            if nb in frame_score:
                a += frame_score[nb]
        acc.append((f, a))
        best = max(best, a)
    keep = float(np.float32(0.75)) * best
    acc = [fa for fa in acc if not fa[1] < keep]
    if sorted_:
        acc = sorted(acc, key=lambda fa: -fa[1])
    return scored, [f for f, _ in acc]


@pytest.mark.gpu
def test_hip_keyframe_database_relocalization_candidates(hip_ctx):
    """uh_bowdb_*: bags of words produced by the GPU vocabulary descent, common-word counts and fBow::score of every keyframe on
    the GPU, cuts on the host; compared with the dict restatement (scores to the last bit: same products, same order)."""
    from ucoslam_cv3_amd.bow import KPFrameDataBase, Vocabulary, write_vocabulary_stream

    params, blob, meta = synth.vocabulary(k=10, depth=3, seed=5, aligment=8)
    voc = Vocabulary(hip_ctx).fromStream(write_vocabulary_stream(params, blob))
    rng = np.random.default_rng(3)
    scene = rng.integers(0, 256, (4000, 32), dtype=np.uint8)          # a pool of "world" descriptors
    def frame_desc(center, n=500):
        idx = (center + rng.integers(0, 900, n)) % len(scene)         # keyframes see overlapping windows of the pool
        return scene[idx] ^ np.packbits(rng.random((n, 256)) < 0.02, axis=1, bitorder="little")
    db_py, db = {}, KPFrameDataBase(hip_ctx)
    for k in range(60):
        bow = voc.transform(frame_desc(60 * k), 3)[0]
        fid = 3 * k + 1
        db_py[fid] = dict(bow)
        db.add(fid, bow)
    assert db.size() == 60
    nbrs = {f: [(g, 100 - abs(f - g)) for g in sorted(db_py, key=lambda g: abs(f - g)) if g != f][:14] for f in db_py}
    covis = lambda f: nbrs[f]
    for trial, (center, min_score, excluded, srt) in enumerate([(600, 0.0, (), True), (1500, 0.01, (76, 79), True), (3000, 0.0, (), False), (123, 0.3, (), True)]):
        q = voc.transform(frame_desc(center, 450), 3)[0]
        got_scored = db.scoredFrames(q, min_score, excluded)
        ref_scored, ref_cand = _py_reloc(db_py, dict(q), covis, srt, min_score, set(excluded))
        assert [(f, n) for f, n, _ in got_scored] == [(f, n) for f, n, _ in ref_scored]
        assert [s for _, _, s in got_scored] == [s for _, _, s in ref_scored]
        assert db.relocalizationCandidates(q, covis, srt, min_score, excluded) == ref_cand
        if trial == 0:
            assert len(ref_cand) >= 1 and abs(ref_cand[0] - (3 * 10 + 1)) <= 9       # the keyframes around window 600 win
    # del: the frame disappears from the answers; deleting twice is an error
    import ucoslam_cv3_amd as u
    q = voc.transform(frame_desc(600, 450), 3)[0]
    top = db.relocalizationCandidates(q, covis, True, 0.0)[0]
    db.delete(top)
    del db_py[top]
    assert top not in [f for f, _, _ in db.scoredFrames(q)]
    assert db.relocalizationCandidates(q, covis, True, 0.0) == _py_reloc(db_py, dict(q), covis, True, 0.0, set())[1]
    with pytest.raises(u.UcoslamHipError):
        db.delete(top)
