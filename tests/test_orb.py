"""ORB extractor: HIP path (through the C ABI) vs the CPU oracle — pyramid, FAST strength map, keypoints, descriptors."""
import numpy as np
import pytest

import oracle_lib
import synth

CONFIGS = [
    # (w, h, maxFeatures, nlevels, scaleFactor, blur)
    (640, 480, 2000, 8, 1.2, True),
    (1241, 376, 2000, 8, 1.2, True),
    (1241, 376, 4000, 8, 1.2, True),
    (211, 167, 500, 5, 1.2, True),
    (320, 240, 1000, 3, 1.5, False),
    (97, 131, 300, 4, 1.3, True),
]


def _assert_same(kps, desc, rk, rd, tag):
    assert len(kps) == len(rk), f"{tag}: {len(kps)} keypoints vs oracle {len(rk)}"
    for f in ("octave", "class_id", "x", "y", "size", "response", "angle"):
        np.testing.assert_array_equal(kps[f], rk[f], err_msg=f"{tag}: field {f}")
    np.testing.assert_array_equal(desc, rd, err_msg=f"{tag}: descriptors")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}f_{c[3]}l")
def test_hip_orb_bit_exact(hip_ctx, oracle, cfg):
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    w, h, nf, nl, sf, blur = cfg
    ext = ORBextractor.create(hip_ctx)
    ext.doGaussianBlur(blur)
    for seed in (0, 1):
        img = synth.frame(w, h, seed=seed)
        kps, desc = ext.detectAndCompute(img, None, FeatParams(nf, nl, sf))
        # stage parity first (pinpoints a failure): pyramid levels and the FAST strength map
        for l in range(nl):
            ref_level = oracle_lib.orb_pyramid_level(oracle, img, l, nl, sf, blur)
            got_level = ext.debug_level(0, l, 0)
            np.testing.assert_array_equal(got_level, ref_level, err_msg=f"pyramid level {l}")
            np.testing.assert_array_equal(ext.debug_level(0, l, 1), oracle_lib.fast_score_map(oracle, ref_level),
                                          err_msg=f"FAST strength map level {l}")
        rk, rd = oracle_lib.orb_extract(oracle, img, nf, nl, sf, blur)
        _assert_same(kps, desc, rk, rd, f"{cfg} seed {seed}")
        assert len(kps) > 0.5 * nf or w < 200


@pytest.mark.gpu
def test_hip_orb_edge_inputs(hip_ctx, oracle):
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(500, 4, 1.2)
    # empty image -> silent empty result (ORBextractor.cpp:1254)
    k, d = ext.detectAndCompute(np.zeros((0, 0), np.uint8), None, fp)
    assert len(k) == 0 and d.shape == (0, 32)
    # featureless image -> zero keypoints, as the oracle
    flat = np.full((120, 160), 90, np.uint8)
    k, d = ext.detectAndCompute(flat, None, fp)
    rk, rd = oracle_lib.orb_extract(oracle, flat, 500, 4, 1.2)
    assert len(k) == len(rk) == 0
    # low-contrast texture forces the per-cell 20 -> 7 threshold fallback
    weak = (100 + (synth.frame(300, 200, seed=5).astype(np.float32) - 100) * 0.11).astype(np.uint8)
    k, d = ext.detectAndCompute(weak, None, fp)
    rk, rd = oracle_lib.orb_extract(oracle, weak, 500, 4, 1.2)
    _assert_same(k, d, rk, rd, "weak texture")
    assert (rk["response"] < 20).any()
    # strided host rows (cv::Mat ROI)
    big = synth.frame(400, 300, seed=9)
    roi = big[10:250, 20:340]
    k, d = ext.detectAndCompute(roi, None, fp)
    rk, rd = oracle_lib.orb_extract(oracle, np.ascontiguousarray(roi), 500, 4, 1.2)
    _assert_same(k, d, rk, rd, "strided ROI")


@pytest.mark.gpu
def test_hip_orb_batch_resident_frames(hip_ctx, oracle):
    """Frames resident in HBM, several per launch (the bench path): every frame equals its own oracle run."""
    import torch

    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, FeatParams, ORBextractor

    ext = ORBextractor.create(hip_ctx)
    frames = np.stack([synth.frame(640, 480, seed=s, shift=(3 * s, s)) for s in range(3)])
    kps, desc, counts = ext.extract_batch(torch.from_numpy(frames).cuda(), FeatParams(2000, 8, 1.2))
    torch.cuda.synchronize()
    kps, desc, counts = kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
    for f in range(3):
        rk, rd = oracle_lib.orb_extract(oracle, frames[f], 2000, 8, 1.2)
        n = counts[f]
        got = kps[f, :n].copy().view(KEYPOINT_DTYPE).reshape(-1)
        _assert_same(got, desc[f, :n], rk, rd, f"batch frame {f}")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(640, 480, 2000, 8, 1.2, 1), (1241, 376, 4000, 8, 1.2, 2), (320, 240, 600, 4, 1.2, 3), (500, 400, 3000, 3, 1.5, 4)],
                         ids=lambda c: f"{c[0]}x{c[1]}_n{c[2]}_l{c[3]}")
def test_hip_orb_nonmaxima_switch_bit_exact(hip_ctx, oracle, cfg):
    """debug string "orb_nonmaxima": the sequential, order-dependent radius-3 suppression per level (ORBextractor.cpp:1176-1205)."""
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    w, h, nf, nl, sf, seed = cfg
    img = synth.frame(w, h, seed=seed)
    if seed == 4:
        img = (img.astype(np.int32) // 8 * 8).astype(np.uint8)    # plateaus: many equal responses next to each other
    ext = ORBextractor.create(hip_ctx)
    ext.setNonMaxima(True)
    kps, desc = ext.detectAndCompute(img, None, FeatParams(nf, nl, sf))
    rk, rd = oracle_lib.orb_extract(oracle, img, nf, nl, sf, nonmaxima=True)
    assert len(kps) == len(rk) > 50
    for f in ("x", "y", "angle", "response", "octave", "size", "class_id"):
        np.testing.assert_array_equal(kps[f], rk[f], err_msg=f)
    np.testing.assert_array_equal(desc, rd)
    plain, _ = oracle_lib.orb_extract(oracle, img, nf, nl, sf)
    assert len(rk) < len(plain)
    ext.setNonMaxima(False)                                       # the switch can be cleared again here (sticky in the reference)
    kps2, _ = ext.detectAndCompute(img, None, FeatParams(nf, nl, sf))
    assert len(kps2) == len(plain) and (kps2["class_id"] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(1920, 1080, 4000, 8, 1.2), (3000, 2000, 10000, 8, 1.2), (4095, 700, 6000, 10, 1.2), (2048, 1536, 20000, 4, 1.5)],
                         ids=lambda c: f"{c[0]}x{c[1]}_{c[2]}f_{c[3]}l")
def test_hip_orb_large_frames_bit_exact(hip_ctx, oracle, cfg):
    """Sizes beyond the benchmark's: full-HD and multi-megapixel frames, feature budgets up to 20 000, the widest image the
    12-bit candidate coordinates allow (4095)."""
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    w, h, nf, nl, sf = cfg
    img = synth.frame(w, h, seed=77)
    ext = ORBextractor.create(hip_ctx)
    kps, desc = ext.detectAndCompute(img, None, FeatParams(nf, nl, sf))
    rk, rd = oracle_lib.orb_extract(oracle, img, nf, nl, sf)
    _assert_same(kps, desc, rk, rd, str(cfg))
    assert len(kps) > 0.5 * nf


@pytest.mark.gpu
def test_hip_orb_pinned_buffers_take_the_copy_free_path(hip_ctx, oracle):
    """uh_orb_extract with the frame and the output arrays in pinned host memory (the kernels read / write them where they lie, the
    host polls a completion word) returns exactly what the pageable path returns; a capacity below the keypoint count is refused."""
    import ctypes as C

    import torch

    from ucoslam_cv3_amd._lib import check, lib, np_ptr, UcoslamHipError
    from ucoslam_cv3_amd.orb import KEYPOINT_DTYPE, FeatParams, ORBextractor

    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(2000, 8, 1.2)
    for (w, h, pad) in ((1241, 376, 0), (640, 480, 32)):
        img = synth.frame(w, h, seed=5)
        rk, rd = ext.detectAndCompute(img, None, fp)                        # pageable in / out
        pin_img = torch.zeros((h, w + pad), dtype=torch.uint8).pin_memory()  # (a row stride above the width, too)
        pin_img[:, :w] = torch.from_numpy(img)
        cap = 2000
        pk = torch.zeros(cap * 28, dtype=torch.uint8).pin_memory()
        pd = torch.zeros((cap, 32), dtype=torch.uint8).pin_memory()
        n = C.c_int(0)
        for _ in range(2):
            check(lib().uh_orb_extract(ext._h, C.c_void_p(pin_img.data_ptr()), w, h, w + pad, C.c_void_p(pk.data_ptr()), C.c_void_p(pd.data_ptr()), cap, C.byref(n)))
        kps = pk.numpy().view(KEYPOINT_DTYPE)[: n.value]
        _assert_same(kps, pd.numpy()[: n.value], rk, rd, f"pinned {w}x{h}")
        ref_k, ref_d = oracle_lib.orb_extract(oracle, img, 2000, 8, 1.2, True)
        _assert_same(kps, pd.numpy()[: n.value], ref_k, ref_d, f"pinned {w}x{h} vs oracle")
        # mixed: pinned frame, pageable outputs
        k2 = np.zeros(cap, KEYPOINT_DTYPE); d2 = np.zeros((cap, 32), np.uint8)
        check(lib().uh_orb_extract(ext._h, C.c_void_p(pin_img.data_ptr()), w, h, w + pad, np_ptr(k2), np_ptr(d2), cap, C.byref(n)))
        _assert_same(k2[: n.value], d2[: n.value], rk, rd, "pinned frame, pageable outputs")
        with pytest.raises(UcoslamHipError):
            check(lib().uh_orb_extract(ext._h, C.c_void_p(pin_img.data_ptr()), w, h, w + pad, C.c_void_p(pk.data_ptr()), C.c_void_p(pd.data_ptr()), 100, C.byref(n)))
        assert n.value == len(rk)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"UH_ORB_FAST": "map"}, {"UH_ORB_PYRAMID": "chain"}, {"UH_ORB_PYRAMID": "pair"}, {"UH_ORB_FAST": "map", "UH_ORB_PYRAMID": "chain"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_hip_orb_fallback_forms_bit_exact(hip_ctx, oracle, monkeypatch, env):
    """The two forms the plan falls back to when a cell / a level pair does not fit the fused kernels' staging buffers — the strength map +
    separate NMS launch (UH_ORB_FAST=map) and one resize launch per level (UH_ORB_PYRAMID=chain) — forced at the bench's frame size, where
    the fused forms normally run: the same bits as the oracle (keypoints, descriptors, order)."""
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ext = ORBextractor.create(hip_ctx)
    for (w, h, nf) in ((1241, 376, 2000), (640, 480, 1500)):
        img = synth.frame(w, h, seed=3)
        kps, desc = ext.detectAndCompute(img, None, FeatParams(nf, 8, 1.2))
        rk, rd = oracle_lib.orb_extract(oracle, img, nf, 8, 1.2, True)
        _assert_same(kps, desc, rk, rd, f"{env} {w}x{h}")
