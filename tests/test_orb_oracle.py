"""Known-answer tests that pin the ORB oracle's restatement of the OpenCV primitives (parity is otherwise unpinned:
OpenCV is neither vendored in the reference nor installed here — see DESIGN.md)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import synth

P = oracle_lib.P


def test_level_plan_matches_survey_table(oracle):
    """Level sizes / budgets derived independently in SURVEY.md §8 from ORBextractor.cpp:468-515,1369-1370."""
    lw, lh, nf = (np.zeros(8, np.int32) for _ in range(3))
    sc = np.zeros(8, np.float32)
    oracle.oracle_orb_level_plan(1241, 376, 2000, 8, C.c_float(1.2), P(lw), P(lh), P(nf), P(sc))
    assert lw.tolist() == [1241, 1034, 862, 718, 598, 499, 416, 346]
    assert lh.tolist() == [376, 313, 261, 218, 181, 151, 126, 105]
    assert nf.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    oracle.oracle_orb_level_plan(640, 480, 4000, 8, C.c_float(1.2), P(lw), P(lh), P(nf), P(sc))
    assert lw.tolist() == [640, 533, 444, 370, 309, 257, 214, 179]
    assert lh.tolist() == [480, 400, 333, 278, 231, 193, 161, 134]
    assert nf.tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    assert abs(sc[7] - 1.2 ** 7) < 1e-5


def test_gaussian_kernel_and_blur(oracle):
    k = np.zeros(7, np.int32)
    oracle.oracle_orb_gauss_kernel(P(k))
    assert k.tolist() == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256
    const = np.full((20, 33), 137, np.uint8)
    out = np.empty_like(const)
    oracle.oracle_orb_blur(P(const), 33, 20, C.c_size_t(33), P(out))
    assert (out == 137).all()                                   # taps sum to 1 exactly
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = np.empty_like(imp)
    oracle.oracle_orb_blur(P(imp), 21, 21, C.c_size_t(21), P(out))
    expect = (np.outer(k, k).astype(np.int64) * 255 + 32768) >> 16
    np.testing.assert_array_equal(out[7:14, 7:14], expect)
    assert out.sum() == expect.sum()
    # reflect-101 border: a left-edge column ramp stays symmetric about column 0
    ramp = np.tile(np.arange(0, 200, 10, dtype=np.uint8), (9, 1))
    out = np.empty_like(ramp)
    oracle.oracle_orb_blur(P(ramp), 20, 9, C.c_size_t(20), P(out))
    col0 = (2 * (18 * 30 + 34 * 20 + 48 * 10) + 56 * 0)         # reflect: x=-1 -> 1, -2 -> 2, -3 -> 3
    assert out[4, 0] == (col0 * 256 + 32768) >> 16


def test_resize_identity_constant_and_taps(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    out = np.empty_like(img)
    oracle.oracle_orb_resize_cubic(P(img), 53, 37, P(out), 53, 37)
    np.testing.assert_array_equal(out, img)                     # scale 1: taps are (0,2048,0,0)
    const = np.full((40, 60), 201, np.uint8)
    out = np.empty((33, 50), np.uint8)
    oracle.oracle_orb_resize_cubic(P(const), 60, 40, P(out), 50, 33)
    assert (out == 201).all()
    # 2:1 down-scale samples the half-pixel phase: cubic(0.5), A=-0.75 = (-0.09375, 0.59375, 0.59375, -0.09375) * 2048
    ofs = np.zeros(30, np.int32)
    coef = np.zeros((30, 4), np.int16)
    oracle.oracle_orb_cubic_taps(60, 30, P(ofs), P(coef))
    assert ofs.tolist() == [2 * i for i in range(30)]
    assert (coef == np.array([-192, 1216, 1216, -192], np.int16)).all()


def test_fast_known_answers(oracle):
    # isolated dark dot on a bright field: all 16 circle pixels are brighter by 100 -> score 99
    img = np.full((15, 15), 150, np.uint8)
    img[7, 7] = 50
    kp = oracle_lib.fast_detect(oracle, img, 20)
    assert kp.tolist() == [[7, 7, 99]]
    sm = oracle_lib.fast_score_map(oracle, img)
    assert sm[7, 7] == 99
    # 9 contiguous brighter pixels = corner, 8 = not a corner
    circle = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
              (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    for n, expect in ((9, 1), (8, 0)):
        img = np.full((15, 15), 100, np.uint8)
        for k in range(n):
            dx, dy = circle[(k + 5) % 16]
            img[7 + dy, 7 + dx] = 160
        kp = oracle_lib.fast_detect(oracle, img, 20)
        hit = [r for r in kp.tolist() if r[:2] == [7, 7]]
        assert len(hit) == expect
        if expect:
            assert hit[0][2] == 59                               # min |diff| over the arc - 1


def test_fast_rowbuffer_equals_scoremap_formulation(oracle):
    """cv::FAST (row buffers, thresholded scores) == strict in-image 3x3 maxima of the threshold-free strength map with
    score >= t: the formulation the GPU uses per cell."""
    rng = np.random.default_rng(3)
    for trial in range(12):
        h, w = int(rng.integers(7, 60)), int(rng.integers(7, 90))
        img = synth.frame(w + 40, h + 40, seed=trial)[20:20 + h, 20:20 + w].copy()
        if trial % 3 == 0:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        sm = oracle_lib.fast_score_map(oracle, img).astype(np.int32)
        for t in (7, 20, 35):
            ref = oracle_lib.fast_detect(oracle, img, t)
            pad = np.zeros((h + 2, w + 2), np.int32)
            pad[1:-1, 1:-1] = sm
            keep = sm >= t
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx or dy:
                        keep &= sm > pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
            ys, xs = np.nonzero(keep)
            got = np.stack([xs, ys, sm[ys, xs]], 1) if len(xs) else np.zeros((0, 3), int)
            np.testing.assert_array_equal(got, ref.reshape(-1, 3))


def test_fast_atan2_and_umax(oracle):
    oracle.oracle_fast_atan2.restype = C.c_float
    oracle.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
    assert oracle.oracle_fast_atan2(0.0, 5.0) == 0.0
    assert oracle.oracle_fast_atan2(0.0, -5.0) == 180.0
    assert abs(oracle.oracle_fast_atan2(3.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.oracle_fast_atan2(-3.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(1)
    for _ in range(500):
        y, x = rng.integers(-60000, 60000, 2)
        if x == 0 and y == 0:
            continue
        got = oracle.oracle_fast_atan2(float(y), float(x))
        true = np.degrees(np.arctan2(y, x)) % 360
        assert min(abs(got - true), 360 - abs(got - true)) < 0.02
    u = np.zeros(16, np.int32)
    oracle.oracle_orb_umax(P(u))
    assert u.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_retain_best_truncation_semantics(oracle):
    """retainBest keeps ties past n; the reference truncates to n right after, so the first n survivors are
    exactly std::nth_element's prefix."""
    r = np.array([5, 9, 9, 3, 9, 1, 7, 9], np.float32)
    perm = np.zeros(8, np.int32)
    n = oracle.oracle_retain_best(P(r), 8, 2, P(perm))
    assert n == 4 and sorted(r[perm[:n]].tolist()) == [9, 9, 9, 9]
    n = oracle.oracle_retain_best(P(r), 8, 0, P(perm))
    assert n == 0
    n = oracle.oracle_retain_best(P(r), 8, 8, P(perm))
    assert n == 8 and perm.tolist() == list(range(8))


def test_extract_structure_and_angle_zero_descriptor(oracle):
    img = synth.frame(640, 480, seed=2)
    kps, desc = oracle_lib.orb_extract(oracle, img, 2000, 8, 1.2)
    assert 0 < len(kps) <= 2000 and desc.shape == (len(kps), 32)
    assert (np.diff(kps["octave"]) >= 0).all()                  # level-order concatenation
    lw, lh, nf = (np.zeros(8, np.int32) for _ in range(3))
    sc = np.zeros(8, np.float32)
    oracle.oracle_orb_level_plan(640, 480, 2000, 8, C.c_float(1.2), P(lw), P(lh), P(nf), P(sc))
    counts = np.bincount(kps["octave"], minlength=8)
    assert (counts <= nf).all()
    assert (kps["class_id"] == -1).all()
    np.testing.assert_array_equal(kps["size"], np.floor(31 * sc[kps["octave"]]))
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    assert ((kps["response"] >= 7) & (kps["response"] <= 255)).all()
    # every level-0 keypoint sits >= 19 px inside
    l0 = kps[kps["octave"] == 0]
    assert l0["x"].min() >= 19 and l0["x"].max() <= 640 - 20 and l0["y"].min() >= 19 and l0["y"].max() <= 480 - 20
    # horizontal intensity ramp -> centroid angle 0 -> descriptor = raw (unrotated) pattern comparisons
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 3, (64, 1))
    level = oracle_lib.orb_pyramid_level(oracle, ramp, 0, blur=False)
    np.testing.assert_array_equal(level, ramp)


def test_device_sincosf_restatement_equals_libm(tmp_path):
    """csrc/glibc_sincosf.hpp (what the describe kernel evaluates) compiled for the host == the C library's cosf/sinf, the
    functions computeOrbDescriptor calls (ORBextractor.cpp:117-119), on every 13th float of [0, 6.4] (84 M values)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sincosf_host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "host_helpers", "sincosf_host.cpp")])
    out = subprocess.run([exe, "13"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_orb_nonmaxima_switch(oracle):
    """The "orb_nonmaxima" debug switch (ORBextractor.cpp:1176-1205): fewer keypoints, every survivor also present without the
    switch (same position / response / descriptor), survivors carry class_id 1, and no survivor has a surviving strictly
    stronger neighbour within radius 3 on its level."""
    import synth

    img = synth.frame(400, 300, seed=5)
    k0, d0 = oracle_lib.orb_extract(oracle, img, 1500, 4, 1.2)
    k1, d1 = oracle_lib.orb_extract(oracle, img, 1500, 4, 1.2, nonmaxima=True)
    assert 50 < len(k1) < len(k0)
    assert (k0["class_id"] == -1).all() and (k1["class_id"] == 1).all()
    key0 = {(float(a["x"]), float(a["y"]), int(a["octave"])): i for i, a in enumerate(k0)}
    for j, a in enumerate(k1):
        i = key0[(float(a["x"]), float(a["y"]), int(a["octave"]))]
        assert k0["response"][i] == a["response"] and (d0[i] == d1[j]).all()
    # the surviving order is the original order
    idx = [key0[(float(a["x"]), float(a["y"]), int(a["octave"]))] for a in k1]
    assert idx == sorted(idx)
