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
