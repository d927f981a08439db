"""The unpinned stages, hardened as far as this image allows: oracle/orb_oracle.cpp and oracle/bow_oracle.cpp
against tests/py_oracle_orb.py — a second restatement written independently in vectorised numpy — bit for bit, on the synthetic
frames bench.py and the GPU parity tests use."""
import os
import re

import numpy as np
import pytest

import oracle_lib
import py_oracle_orb as po
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = oracle_lib.C
I, VP = oracle_lib.I, oracle_lib.VP
P = oracle_lib.P


def _pattern():
    txt = open(os.path.join(ROOT, "include", "ucoslam_hip_orb_pattern.inc")).read()
    vals = [int(v) for v in re.findall(r"-?\d+", txt.split("UH_ORB_PATTERN_VALUES", 1)[1])]
    assert len(vals) >= 1024
    return np.array(vals[:1024], np.int8).reshape(256, 4)


FRAMES = [(1241, 376, 3), (640, 480, 5), (173, 131, 7)]


@pytest.mark.parametrize("w,h,seed", FRAMES)
def test_blur_and_pyramid_agree(oracle, w, h, seed):
    img = synth.frame(w, h, seed=seed)
    out = np.empty_like(img)
    oracle.oracle_orb_blur.argtypes = [VP, I, I, oracle_lib.SZ, VP]
    oracle.oracle_orb_blur(P(img), w, h, img.strides[0], P(out))
    np.testing.assert_array_equal(po.gaussian_blur7(img), out)
    nl = 8 if min(w, h) > 200 else 4
    for blur in (True, False):
        lv = po.pyramid(img, nl, 1.2, blur)
        for l in range(nl):
            np.testing.assert_array_equal(lv[l], oracle_lib.orb_pyramid_level(oracle, img, l, nl, 1.2, blur), err_msg=f"level {l} blur {blur}")


def test_resize_and_taps_agree(oracle):
    rng = np.random.default_rng(0)
    oracle.oracle_orb_resize_cubic.argtypes = [VP, I, I, VP, I, I]
    oracle.oracle_orb_cubic_taps.argtypes = [I, I, VP, VP]
    for (sw, sh, dw, dh) in [(1241, 376, 1034, 313), (640, 480, 533, 400), (97, 61, 81, 51), (50, 40, 25, 20), (33, 47, 33, 47), (20, 20, 31, 17)]:
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        out = np.empty((dh, dw), np.uint8)
        oracle.oracle_orb_resize_cubic(P(src), sw, sh, P(out), dw, dh)
        np.testing.assert_array_equal(po.resize_cubic(src, dw, dh), out, err_msg=str((sw, sh, dw, dh)))
        ofs, coef = np.empty(dw, np.int32), np.empty(4 * dw, np.int16)
        oracle.oracle_orb_cubic_taps(sw, dw, P(ofs), P(coef))
        s, q = po.cubic_taps(sw, dw)
        np.testing.assert_array_equal(s, ofs)
        np.testing.assert_array_equal(q.reshape(-1), coef)


def test_level_plan_agrees(oracle):
    oracle.oracle_orb_level_plan.argtypes = [I, I, I, I, C.c_float, VP, VP, VP, VP]
    for (w, h, nf, nl, sf) in [(1241, 376, 2000, 8, 1.2), (640, 480, 4000, 8, 1.2), (320, 240, 500, 3, 1.5), (800, 600, 1234, 5, 1.1)]:
        lw, lh, nfeat, scales = np.empty(nl, np.int32), np.empty(nl, np.int32), np.empty(nl, np.int32), np.empty(nl, np.float32)
        oracle.oracle_orb_level_plan(w, h, nf, nl, sf, P(lw), P(lh), P(nfeat), P(scales))
        scale, nfe, sizes = po.level_plan(w, h, nf, nl, sf)
        assert nfe == nfeat.tolist() and [s[0] for s in sizes] == lw.tolist() and [s[1] for s in sizes] == lh.tolist()
        np.testing.assert_array_equal(scale, scales)


@pytest.mark.parametrize("w,h,seed", FRAMES)
def test_fast_strength_and_detection_agree(oracle, w, h, seed):
    img = synth.frame(w, h, seed=seed)
    for l, im in enumerate(po.pyramid(img, 4, 1.2)):
        im = np.ascontiguousarray(im)
        np.testing.assert_array_equal(po.fast_strength(im), oracle_lib.fast_score_map(oracle, im), err_msg=f"level {l}")
    rng = np.random.default_rng(seed)
    im = np.ascontiguousarray(po.pyramid(img, 2, 1.2)[1])
    for _ in range(12):      # cv::FAST on cell-sized sub-images, both thresholds the extractor uses
        cw, ch = int(rng.integers(8, 120)), int(rng.integers(8, 60))
        x0, y0 = int(rng.integers(0, im.shape[1] - cw)), int(rng.integers(0, im.shape[0] - ch))
        sub = np.ascontiguousarray(im[y0:y0 + ch, x0:x0 + cw])
        for thr in (20, 7):
            np.testing.assert_array_equal(po.fast_detect(sub, thr), oracle_lib.fast_detect(oracle, sub, thr), err_msg=f"{cw}x{ch}@{x0},{y0} thr {thr}")


def test_fast_atan2_and_umax_agree(oracle):
    oracle.oracle_fast_atan2.restype = C.c_float
    oracle.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(1)
    y = np.r_[rng.integers(-60000, 60000, 4000), [0, 0, 5, -5, 1, 0]].astype(np.float32)
    x = np.r_[rng.integers(-60000, 60000, 4000), [0, 7, 0, 0, 1, -3]].astype(np.float32)
    ref = np.array([oracle.oracle_fast_atan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    np.testing.assert_array_equal(po.fast_atan2(y, x), ref)
    um = np.empty(16, np.int32)
    oracle.oracle_orb_umax.argtypes = [VP]
    oracle.oracle_orb_umax(P(um))
    assert po.umax_table().tolist() == um.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


@pytest.mark.parametrize("w,h,nf,seed", [(1241, 376, 2000, 11), (640, 480, 2000, 12), (400, 300, 700, 13)])
def test_full_extraction_cross_checked_by_the_second_restatement(oracle, w, h, nf, seed):
    """Every keypoint the C++ oracle extracts: (i) is a cv::FAST keypoint of a cell per the numpy restatement (threshold 20, or 7 where
    a cell had <= 3), with the same response; (ii) carries the numpy intensity-centroid angle and rotated-BRIEF descriptor bit for
    bit; (iii) maps back from image to level coordinates exactly; per level the count respects precalculateParams' budget."""
    nl, sf = 8, 1.2
    img = synth.frame(w, h, seed=seed)
    kps, desc = oracle_lib.orb_extract(oracle, img, nf, nl, sf)
    assert len(kps) > 0.5 * nf
    scale, nfeat, sizes = po.level_plan(w, h, nf, nl, sf)
    lv = po.pyramid(img, nl, sf)
    pat = _pattern()
    for l in range(nl):
        sel = kps["octave"] == l
        k, d = kps[sel], desc[sel]
        assert len(k) <= nfeat[l]
        if len(k) == 0:
            continue
        if l == 0:
            xs, ys = k["x"].astype(np.int64), k["y"].astype(np.int64)
            assert (xs == k["x"]).all() and (ys == k["y"]).all()
        else:
            xs = np.rint(k["x"] / scale[l] - np.float32(0.5)).astype(np.int64)
            ys = np.rint(k["y"] / scale[l] - np.float32(0.5)).astype(np.int64)
            np.testing.assert_array_equal(((xs.astype(np.float32) + np.float32(0.5)) * scale[l]).astype(np.float32), k["x"])
            np.testing.assert_array_equal(((ys.astype(np.float32) + np.float32(0.5)) * scale[l]).astype(np.float32), k["y"])
        im = lv[l]
        np.testing.assert_array_equal(po.ic_angle(im, xs, ys), k["angle"], err_msg=f"level {l} angles")
        np.testing.assert_array_equal(po.orb_descriptor(im, xs, ys, k["angle"], pat), d, err_msg=f"level {l} descriptors")
        assert (k["size"] == np.float32(int(31 * scale[l]))).all()
        # (i) membership in a cell's FAST output
        rects, cols, rows, quota = po.cell_grid(im.shape[1], im.shape[0], w, h, nfeat[l])
        cand = set()
        for r in rects:
            if r is None:
                continue
            x0, y0, x1, y1 = r
            assert 0 <= x0 and 0 <= y0 and x1 <= im.shape[1] and y1 <= im.shape[0]
            sub = np.ascontiguousarray(im[y0:y1, x0:x1])
            c = po.fast_detect(sub, 20)
            if len(c) <= 3:
                c = po.fast_detect(sub, 7)
            cand.update((int(a) + x0, int(b) + y0, int(s)) for a, b, s in c)
        got = set(zip(xs.tolist(), ys.tolist(), k["response"].astype(np.int64).tolist()))
        assert got <= cand, f"level {l}: {len(got - cand)} oracle keypoints are not cv::FAST keypoints of any cell"


@pytest.mark.parametrize("k,depth,al,seed", [(10, 3, 8, 0), (10, 4, 8, 1), (9, 2, 32, 2), (32, 2, 16, 3), (6, 5, 8, 4)])
def test_bow_descent_agrees(oracle, k, depth, al, seed):
    import struct

    from ucoslam_cv3_amd.bow import PARAMS_FMT

    params, blob, meta = synth.vocabulary(k=k, depth=depth, seed=seed, aligment=al)
    f = struct.unpack(PARAMS_FMT, params)
    pd = dict(m_k=f[10], desc_size_bytes_wp=f[3], block_size_bytes_wp=f[4], feature_off_start=f[5], child_off_start=f[6])
    rng = np.random.default_rng(seed)
    desc = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    oracle.oracle_bow_transform.argtypes = [VP, VP, VP, I, oracle_lib.SZ, I, VP, VP, VP, VP]
    pbuf = np.frombuffer(params, np.uint8).copy()
    bbuf = np.frombuffer(blob, np.uint8).copy()
    for level in (0, 1, 3, depth - 1, depth + 2):
        word, weight = np.empty(len(desc), np.uint32), np.empty(len(desc), np.float32)
        node, valid = np.empty(len(desc), np.uint32), np.empty(len(desc), np.uint8)
        assert oracle.oracle_bow_transform(P(pbuf), P(bbuf), P(desc), len(desc), 32, level, P(word), P(weight), P(node), P(valid)) == 0
        w2, wt2, n2, v2 = po.bow_transform(pd, blob, desc, level)
        np.testing.assert_array_equal(w2, word)
        np.testing.assert_array_equal(wt2, weight)
        np.testing.assert_array_equal(v2, valid)
        np.testing.assert_array_equal(n2[valid == 1], node[valid == 1])
