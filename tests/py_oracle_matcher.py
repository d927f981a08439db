"""TEST INFRASTRUCTURE ONLY — pure-Python restatement of the FrameMatcher_Flann filter chain (framematcher.cpp:248-316,
misc.cpp:105-185), written independently of the product's C++ (csrc/matcher.hip) so that the two can be compared.
Parity with the reference is unpinned (Frame needs OpenCV); the known-answer cases live in tests/test_matcher.py."""
import numpy as np


def filter_ambiguous(matches, key):
    if not matches:
        return matches
    used = {}
    needs = False
    for idx, m in enumerate(matches):
        k = m[key]
        if k not in used:
            used[k] = idx
        elif matches[used[k]]["distance"] > m["distance"]:
            matches[used[k]][key] = -1
            used[k] = idx
            needs = True
        else:
            m[key] = -1
            needs = True
    return [m for m in matches if m["trainIdx"] != -1 and m["queryIdx"] != -1] if needs else matches


def match_filter(indices, distances, q, t, map_q, map_t, min_desc_dist, ratio, check_orientation, max_octave_diff):
    F32 = np.float32
    matches = []
    for i in range(indices.shape[0]):
        best, best2 = F32(min_desc_dist), F32(np.finfo(np.float32).max)
        bq = bt = -1
        oct2 = -1
        qi = int(map_q[i]) if map_q is not None else i
        for j in range(indices.shape[1]):
            d = F32(distances[i, j])
            if d > F32(min_desc_dist):
                continue
            if d < best2:
                ti = int(indices[i, j])
                if ti < 0:
                    continue
                tk = int(map_t[ti]) if map_t is not None else ti
                if abs(int(t["octave"][tk]) - int(q["octave"][qi])) > max_octave_diff:
                    continue
                if d < best:
                    best, bq, bt = d, qi, tk
                else:
                    best2, oct2 = d, int(t["octave"][tk])
        if bq != -1 and not (oct2 == int(q["octave"][bq]) and best > best2 * F32(ratio)):
            matches.append(dict(queryIdx=bq, trainIdx=bt, distance=float(best)))
    matches = filter_ambiguous(matches, "trainIdx")
    if check_orientation:
        hist = [[] for _ in range(30)]
        factor = F32(1.0) / F32(30)
        for k, m in enumerate(matches):
            rot = F32(t["angle"][m["trainIdx"]]) - F32(q["angle"][m["queryIdx"]])
            if rot < 0:
                rot = F32(rot + F32(360.0))
            b = int(np.floor(float(F32(rot * factor)) + 0.5))      # C round(): half away from zero (positive here)
            if b == 30:
                b = 0
            hist[b].append(k)
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i, h in enumerate(hist):
            s = len(h)
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if m2 < F32(0.1) * F32(m1):
            i2 = i3 = -1
        elif m3 < F32(0.1) * F32(m1):
            i3 = -1
        for i, h in enumerate(hist):
            if i in (i1, i2, i3):
                continue
            for k in h:
                matches[k]["queryIdx"] = matches[k]["trainIdx"] = -1
        matches = [m for m in matches if m["trainIdx"] != -1 and m["queryIdx"] != -1]
    return matches
