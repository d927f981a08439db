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
    return finish(matches, q, t, check_orientation)


def finish(matches, q, t, check_orientation):
    """filter_ambiguous_train + orientation histogram (framematcher.cpp:288-316 and :504-531)"""
    F32 = np.float32
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


def epipolar_sq_dist(kp1, kp2, F):
    """misc.h:72-81 in float arithmetic; F row-major 3x3 (flat)"""
    F32 = np.float32
    a = F32(F32(F32(kp1[0] * F[0]) + F32(kp1[1] * F[3])) + F[6])
    b = F32(F32(F32(kp1[0] * F[1]) + F32(kp1[1] * F[4])) + F[7])
    den = F32(F32(a * a) + F32(b * b))
    if den == 0:
        return np.finfo(np.float32).max
    c = F32(F32(F32(kp1[0] * F[2]) + F32(kp1[1] * F[5])) + F[8])
    num = F32(F32(F32(a * kp2[0]) + F32(b * kp2[1])) + c)
    return F32(F32(num * num) / den)


def bow_match(q, t, q_used, t_used, min_desc_dist, ratio, check_orientation, max_octave_diff, F12=None):
    """FrameMatcher_BoW::matchEpipolar (framematcher.cpp:407-535) on frame dicts with `bowvector_level` {node: [kp idx]}."""
    F32 = np.float32
    matches = []
    qb, tb = q["bowvector_level"], t["bowvector_level"]
    sf2 = [F32(v * v) for v in q["scaleFactors"].astype(np.float32)]
    for node in sorted(set(qb) & set(tb)):                      # the merge-join visits the common keys in ascending order
        for qidx in qb[node]:
            if not q_used[qidx]:
                continue
            best, best2 = F32(min_desc_dist), F32(np.finfo(np.float32).max)
            bq = bt = -1
            oct2 = -1
            for tidx in tb[node]:
                if not t_used[tidx]:
                    continue
                if abs(int(t["octave"][tidx]) - int(q["octave"][qidx])) > max_octave_diff:
                    continue
                if F12 is not None and float(epipolar_sq_dist(t["pt"][tidx], q["pt"][qidx], F12)) >= 3.84 * float(sf2[int(q["octave"][qidx])]):
                    continue
                d = F32(int(np.unpackbits(t["desc"][tidx] ^ q["desc"][qidx]).sum()))
                if d < best:
                    best, bq, bt = d, qidx, tidx
                else:
                    best2, oct2 = d, int(t["octave"][tidx])
            if bq != -1 and not (oct2 == int(q["octave"][bq]) and best > F32(best2 * F32(ratio))):
                matches.append(dict(queryIdx=int(bq), trainIdx=int(bt), distance=float(best)))
    return finish(matches, q, t, check_orientation)
