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
