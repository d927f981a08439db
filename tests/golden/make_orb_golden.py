"""Pins the ORB stage to OpenCV — to be run ONCE on any machine where `cv2` is importable (this build container has none).

Every arithmetic primitive of the reference extractor lives in OpenCV (call sites in src/featureextractors/ORBextractor.cpp):
  GaussianBlur :1262, resize(INTER_CUBIC) :1379, FAST :980,986, fastAtan2 :105 (retainBest and cvRound have no Python binding: they are
  pinned by libstdc++'s nth_element in tests/test_introselect.py and by the IEEE round-half-even definition).
This script runs exactly those calls on tests/synth.frame() inputs, stage by stage, and writes tests/golden/orb_golden.npz with the
OpenCV version that produced it.  tests/test_opencv_golden.py then compares the CPU oracle (-m "not gpu") and the HIP kernels
(-m gpu, through uh_orb_debug_level) with the file; while the file is absent those tests report "parity unpinned" and skip.

    python tests/golden/make_orb_golden.py            # needs: cv2, numpy
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

CASES = [dict(w=320, h=240, seed=5), dict(w=640, h=480, seed=9), dict(w=1241, h=376, seed=0)]
NLEVELS, SCALE = 8, 1.2


def level_sizes(w, h):
    """ORBextractor.cpp:1369-1370: cvRound(w * invScale) with the float chain scale[i] = scale[i-1] * 1.2f, inv = 1.0f / scale."""
    out, sc = [], np.float32(1.0)
    for _ in range(NLEVELS):
        inv = np.float32(1.0) / sc
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
        sc = np.float32(sc * np.float32(SCALE))
    return out


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: nothing written.  Run this script where OpenCV's Python module exists.")
        return 2
    out = {"cv_version": np.array(cv2.__version__), "nlevels": np.array(NLEVELS), "scale": np.array(SCALE, np.float32),
           "cases": np.array([[c["w"], c["h"], c["seed"]] for c in CASES], np.int32)}
    for ci, c in enumerate(CASES):
        img = synth.frame(c["w"], c["h"], seed=c["seed"])
        # :1261-1263  GaussianBlur(image, image, Size(7, 7), 2, 2, BORDER_REFLECT_101) on the input only
        cur = cv2.GaussianBlur(img, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
        for lvl, (lw, lh) in enumerate(level_sizes(c["w"], c["h"])):
            if lvl > 0:   # :1379  resize(level l-1, level l, sz, 0, 0, INTER_CUBIC)
                cur = cv2.resize(cur, (lw, lh), interpolation=cv2.INTER_CUBIC)
            out[f"c{ci}_level{lvl}"] = cur.copy()
            if c["w"] <= 640:   # FAST on the whole level at both thresholds (:980 iniThFAST = 20, :986 minThFAST = 7), nonmax on, TYPE_9_16
                for th in (20, 7):
                    det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
                    kps = det.detect(cur, None)
                    out[f"c{ci}_fast{th}_level{lvl}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    # ---- diagnostics: whichever OpenCV this file comes from, a mismatch of a level is traced to ONE primitive in one run
    # (a) the Gaussian taps: cv::getGaussianKernel's floats and what the 8-bit filter really applies — the response to an impulse of 255 in a
    #     flat 0 image (row and column taps x 255, rounded by the fixed-point path: tells the legacy cvRound(k * 256) taps from the bit-exact ones)
    out["diag_gauss_kernel_f64"] = cv2.getGaussianKernel(7, 2).reshape(-1)
    out["diag_gauss_kernel_f32"] = cv2.getGaussianKernel(7, 2, cv2.CV_32F).reshape(-1)
    imp = np.zeros((15, 15), np.uint8); imp[7, 7] = 255
    out["diag_gauss_impulse"] = cv2.GaussianBlur(imp, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
    ramp = (np.add.outer(np.arange(23) * 11, np.arange(37) * 7) % 256).astype(np.uint8)
    out["diag_gauss_ramp_in"] = ramp
    out["diag_gauss_ramp"] = cv2.GaussianBlur(ramp, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)   # (the reflect-101 border included)
    # (b) INTER_CUBIC 8U: a ramp and an impulse image resized by 1 / 1.2 (taps, their 11-bit rounding, the (v + 2^21) >> 22 descale, edge clamping)
    out["diag_resize_ramp"] = cv2.resize(ramp, (int(np.rint(37 / 1.2)), int(np.rint(23 / 1.2))), interpolation=cv2.INTER_CUBIC)
    imp2 = np.zeros((24, 36), np.uint8); imp2[::5, ::7] = 255
    out["diag_resize_impulse_in"] = imp2
    out["diag_resize_impulse"] = cv2.resize(imp2, (30, 20), interpolation=cv2.INTER_CUBIC)
    # (c) cv::FAST on SUB-IMAGES, as ComputeKeyPoints_thread calls it per cell (:980,986): the rows / columns closer than 3 to the sub-image's
    #     edge score 0 and the non-maximum test sees zeros beyond it — per-cell detection is not a crop of whole-image detection
    base = out["c0_level0"]
    rois = [(0, 0, 36, 36), (16, 16, 52, 52), (100, 60, 131, 92), (base.shape[1] - 40, base.shape[0] - 37, base.shape[1], base.shape[0]), (5, 120, 70, 150)]
    out["diag_fast_rois"] = np.array(rois, np.int32)
    for ri, (x0, y0, x1, y1) in enumerate(rois):
        sub = np.ascontiguousarray(base[y0:y1, x0:x1])
        for th in (20, 7):
            det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
            out[f"diag_fast_roi{ri}_th{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in det.detect(sub, None)], np.float32).reshape(-1, 3)
    # (d) KeyPointsFilter::retainBest has no Python binding in stock OpenCV; where a build exposes it, its order is recorded for level 0's strongest 700
    try:
        det = cv2.FastFeatureDetector_create(threshold=7, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        kps = det.detect(base, None)
        kept = cv2.KeyPointsFilter.retainBest(list(kps), 700)   # (AttributeError on stock builds)
        out["diag_retain_best_in"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32)
        out["diag_retain_best_out"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kept], np.float32)
    except Exception as e:   # noqa: BLE001
        out["diag_retain_best_unavailable"] = np.array(repr(e))
    # :105  fastAtan2((float)m_01, (float)m_10) on a grid of integer moments
    rng = np.random.default_rng(3)
    yx = np.concatenate([rng.integers(-40000, 40001, (4000, 2)), [[0, 0], [0, 5], [0, -5], [3, 0], [-3, 0], [1, 1], [-1, -1]]]).astype(np.float32)
    out["atan_yx"] = yx
    out["atan_deg"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in yx], np.float32)
    path = os.path.join(HERE, "orb_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} (OpenCV {cv2.__version__}, {len(out)} arrays)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
