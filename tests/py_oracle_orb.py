"""TEST INFRASTRUCTURE — a SECOND, independent restatement (vectorised numpy, written separately from oracle/orb_oracle.cpp and
oracle/bow_oracle.cpp) of the arithmetic the unpinned stages rest on.  OpenCV is absent from this image, so nothing of the
reference's extractor can be compiled; two restatements that were written along different code paths and agree
bit for bit on the synthetic frames are the strongest pin available here.

What is restated (reference call sites in src/featureextractors/ORBextractor.cpp; OpenCV semantics per SURVEY.md Appendix A):
  gaussian_blur7      cv::GaussianBlur(8U, 7x7, sigma 2, BORDER_REFLECT_101), bit-exact fixed-point path        (:1262)
  resize_cubic        cv::resize(INTER_CUBIC) 8U fixed-point path, taps clamped at the ROI edge                  (:1379)
  pyramid             ComputePyramid level chain with cvRound sizes                                              (:1355-1393)
  fast_strength       cornerScore<16> as a threshold-free map: max over the sixteen 9-arcs of the min difference (:980,986)
  fast_detect         cv::FAST(sub-image, thr, nonmax=true) from the strength map: strict 3x3 maxima inside the sub-image
  fast_atan2          cv::fastAtan2 polynomial, float32, un-fused                                                (:105)
  ic_angle            intensity centroid over the radius-15 disc                                                 (:79-106)
  orb_descriptor      rotated BRIEF-256 with (float)cos / (float)sin of the float angle and half-even rounding   (:113-153)
  level_plan          precalculateParams: scale chain, features per level                                        (:468-515)
  cell_grid           the FAST cell rectangles of ComputeKeyPoints_thread                                        (:899-976)
  bow_transform       fbow::Vocabulary::_transform2<L1_32bytes>, all descriptors descending level by level       (fbow.h:402-447)
The order-defining part (std::nth_element inside retainBest) is NOT restated here: tests/test_introselect.py pins it against
libstdc++ itself; the checks built on this module are tie-tolerant set properties of retainBest instead.
"""
import numpy as np

f32 = np.float32
EDGE = 19
HALF_PATCH = 15


# ------------------------------------------------------------------------------------------------ blur
def gauss_taps7():
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-(x * x) / (2.0 * 2.0 * 2.0))
    k /= k.sum()
    taps = np.zeros(7, np.int64)
    err = 0.0
    for i in range(3):                      # symmetric error diffusion towards the centre, centre takes the remainder
        adj = k[i] * 256.0 + err
        v = int(np.rint(adj))
        err = adj - v
        taps[i] = taps[6 - i] = v
    taps[3] = 256 - 2 * taps[:3].sum()
    return taps


def gaussian_blur7(img):
    t = gauss_taps7()
    p = np.pad(img.astype(np.int64), ((0, 0), (3, 3)), mode="reflect")          # BORDER_REFLECT_101 = numpy 'reflect'
    h = sum(t[i] * p[:, i:i + img.shape[1]] for i in range(7))                  # 8.8 fixed point
    p = np.pad(h, ((3, 3), (0, 0)), mode="reflect")
    v = sum(t[i] * p[i:i + img.shape[0], :] for i in range(7))                  # 16.16
    return ((v + 32768) >> 16).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ resize
def cubic_taps(ssize, dsize):
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(f32)).astype(f32)
    A = f32(-0.75)
    one = f32(1)
    xp1 = x + one
    c0 = ((A * xp1 - f32(5) * A) * xp1 + f32(8) * A) * xp1 - f32(4) * A
    c1 = ((A + f32(2)) * x - (A + f32(3))) * x * x + one
    omx = one - x
    c2 = ((A + f32(2)) * omx - (A + f32(3))) * omx * omx + one
    c3 = one - c0 - c1 - c2
    coef = np.stack([c0, c1, c2, c3], 1).astype(f32)
    q = np.clip(np.rint(coef * f32(2048)).astype(np.int64), -32768, 32767)
    return s, q


def resize_cubic(src, dw, dh):
    sh, sw = src.shape
    xo, xa = cubic_taps(sw, dw)
    yo, ya = cubic_taps(sh, dh)
    S = src.astype(np.int64)
    cols = np.clip(xo[:, None] - 1 + np.arange(4)[None, :], 0, sw - 1)         # [dw,4]
    hor = (S[:, cols] * xa[None, :, :]).sum(2)                                  # [sh,dw]
    rows = np.clip(yo[:, None] - 1 + np.arange(4)[None, :], 0, sh - 1)         # [dh,4]
    acc = (hor[rows, :] * ya[:, :, None]).sum(1)                                # [dh,dw]
    return np.clip((acc + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def level_plan(w, h, max_features, nlevels, scale_factor):
    scale = np.ones(nlevels, f32)
    for i in range(1, nlevels):
        scale[i] = f32(scale[i - 1] * f32(scale_factor))
    inv = (f32(1) / scale).astype(f32)
    factor = f32(1) / f32(scale_factor)
    n_desired = f32(f32(max_features) * (f32(1) - factor) / (f32(1) - f32(np.power(np.float64(factor), np.float64(nlevels)))))
    nfeat, total = [], 0
    for _ in range(nlevels - 1):
        nfeat.append(int(np.rint(n_desired)))
        total += nfeat[-1]
        n_desired = f32(n_desired * factor)
    nfeat.append(max(max_features - total, 0))
    sizes = [(int(np.rint(f32(w) * inv[l])), int(np.rint(f32(h) * inv[l]))) for l in range(nlevels)]
    return scale, nfeat, sizes


def pyramid(img, nlevels=8, scale_factor=1.2, blur=True):
    scale, _, sizes = level_plan(img.shape[1], img.shape[0], 1000, nlevels, scale_factor)
    lv = [gaussian_blur7(img) if blur else img.copy()]
    for l in range(1, nlevels):
        lv.append(resize_cubic(lv[-1], sizes[l][0], sizes[l][1]))
    return lv


# ------------------------------------------------------------------------------------------------ FAST
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_strength(img):
    """score[y][x] = max(0, max over the sixteen 9-arcs of min(v - ring), the same for ring - v) - 1, clamped to [0,255];
    0 in the 3-pixel frame.  "corner at threshold t" <=> score >= t (cornerScore<16>)."""
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    if h < 7 or w < 7:
        return out
    I = img.astype(np.int32)
    v = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in CIRCLE])     # [16, h-6, w-6]
    d = v[None] - ring
    d2 = np.concatenate([d, d[:8]], 0)                                                   # cyclic extension to 24
    win_min = np.stack([d2[k:k + 9].min(0) for k in range(16)]).max(0)                   # ring darker than the centre
    win_max = np.stack([(-d2[k:k + 9]).min(0) for k in range(16)]).max(0)                # ring brighter
    s = np.maximum(np.maximum(win_min, win_max), 0) - 1
    out[3:h - 3, 3:w - 3] = np.clip(s, 0, 255).astype(np.uint8)
    return out


def fast_detect(sub, threshold):
    """cv::FAST(sub, thr, nonmax=true): (x, y, score) in raster order.  A pixel that is not a corner at `threshold` has a buffered
    score of 0; a corner is kept iff its score is strictly greater than its eight neighbours' buffered scores."""
    threshold = min(max(int(threshold), 0), 255)
    s = fast_strength(sub).astype(np.int32)
    buf = np.where(s >= max(threshold, 1), s, 0) if threshold > 0 else s
    # at threshold t a corner needs max(A,B) > t  <=>  score = max(A,B) - 1 >= t; score 0 with t == 0 still needs max(A,B) >= 1... cv::FAST
    # with t = 0 is not used by the extractor (thresholds 20 and 7)
    h, w = s.shape
    p = np.pad(buf, 1)
    nb = np.stack([p[1 + dy:h + 1 + dy, 1 + dx:w + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)]).max(0)
    keep = (buf > 0) & (buf > nb)
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, buf[ys, xs]], 1).astype(np.int32)


# ------------------------------------------------------------------------------------------------ orientation / descriptor
_P1 = f32(f32(0.9997878412794807) * f32(180 / np.pi))
_P3 = f32(f32(-0.3258083974640975) * f32(180 / np.pi))
_P5 = f32(f32(0.1555786518463281) * f32(180 / np.pi))
_P7 = f32(f32(-0.04432655554792128) * f32(180 / np.pi))
_EPS = f32(np.finfo(np.float64).eps)


def fast_atan2(y, x):
    y, x = np.asarray(y, f32), np.asarray(x, f32)
    ax, ay = np.abs(x), np.abs(y)
    swap = ax < ay
    num, den = np.where(swap, ax, ay), np.where(swap, ay, ax)
    c = (num / (den + _EPS)).astype(f32)
    c2 = (c * c).astype(f32)
    a = ((((_P7 * c2 + _P5).astype(f32) * c2 + _P3).astype(f32) * c2 + _P1).astype(f32) * c).astype(f32)
    a = np.where(swap, f32(90) - a, a).astype(f32)
    a = np.where(x < 0, f32(180) - a, a).astype(f32)
    a = np.where(y < 0, f32(360) - a, a).astype(f32)
    return a


def umax_table():
    um = np.zeros(HALF_PATCH + 1, np.int64)
    vmax = int(np.floor(HALF_PATCH * np.sqrt(f32(2)) / 2 + 1))
    vmin = int(np.ceil(HALF_PATCH * np.sqrt(f32(2)) / 2))
    for v in range(vmax + 1):
        um[v] = int(np.rint(np.sqrt(float(HALF_PATCH * HALF_PATCH - v * v))))
    v0 = 0
    for v in range(HALF_PATCH, vmin - 1, -1):
        while um[v0] == um[v0 + 1]:
            v0 += 1
        um[v] = v0
        v0 += 1
    return um


def ic_angle(im, xs, ys):
    """Angles (degrees, float32) of the keypoints at integer level coordinates (xs, ys)."""
    um = umax_table()
    I = im.astype(np.int64)
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    m10 = np.zeros(len(xs), np.int64)
    m01 = np.zeros(len(xs), np.int64)
    for v in range(-HALF_PATCH, HALF_PATCH + 1):
        d = um[abs(v)]
        for u in range(-d, d + 1):
            px = I[ys + v, xs + u]
            m10 += u * px
            m01 += v * px
    return fast_atan2(m01.astype(f32), m10.astype(f32))


def orb_descriptor(im, xs, ys, angles, pattern):
    """pattern: int8 [256,4] = (x0, y0, x1, y1) of the 256 tests.  Returns uint8 [n,32]."""
    I = im.astype(np.int64)
    xs, ys = np.asarray(xs, np.int64), np.asarray(ys, np.int64)
    ang = (np.asarray(angles, f32) * f32(np.pi / f32(180.0))).astype(f32)
    a = np.cos(ang.astype(np.float64)).astype(f32)[:, None]
    b = np.sin(ang.astype(np.float64)).astype(f32)[:, None]
    P = pattern.astype(f32)

    def sample(px, py):
        ry = np.rint((px[None, :] * b).astype(f32) + (py[None, :] * a).astype(f32)).astype(np.int64)    # x*b + y*a, un-fused
        rx = np.rint((px[None, :] * a).astype(f32) - (py[None, :] * b).astype(f32)).astype(np.int64)    # x*a - y*b
        return I[ys[:, None] + ry, xs[:, None] + rx]

    bits = (sample(P[:, 0], P[:, 1]) < sample(P[:, 2], P[:, 3])).astype(np.uint8)                         # [n,256]
    return np.packbits(bits.reshape(len(xs), 32, 8), axis=2, bitorder="little").reshape(len(xs), 32)


# ------------------------------------------------------------------------------------------------ cell grid
def cell_grid(level_w, level_h, w0, h0, n_desired):
    """The FAST cell rectangles of one level: list of (x0, y0, x1, y1) in level coordinates, row-major, plus (cols, rows, quota)."""
    ratio = f32(w0) / f32(h0)
    cols = int(np.sqrt(f32(n_desired) / (f32(5) * ratio)))
    rows = int(ratio * f32(cols))
    if cols <= 0 or rows <= 0:
        return [], 0, 0, 0
    minb, maxbx, maxby = EDGE, level_w - EDGE, level_h - EDGE
    cw = int(np.ceil(f32(maxbx - minb) / f32(cols)))
    ch = int(np.ceil(f32(maxby - minb) / f32(rows)))
    quota = int(np.ceil(f32(n_desired) / f32(rows * cols)))
    rects = []
    for i in range(rows):
        y0 = minb + i * ch - 3
        hy = ch + 6 if i < rows - 1 else maxby + 3 - y0
        for j in range(cols):
            x0 = minb + j * cw - 3
            hx = cw + 6 if j < cols - 1 else maxbx + 3 - x0
            rects.append((x0, y0, x0 + hx, y0 + hy) if hx > 0 and hy > 0 else None)
    return rects, cols, rows, quota


# ------------------------------------------------------------------------------------------------ fbow
def bow_transform(params, blob, desc, level):
    """fbow::Vocabulary::_transform2<L1_32bytes> for all descriptors at once, descending block level by block level.
    params: dict with m_k, desc_size_bytes_wp, block_size_bytes_wp, feature_off_start, child_off_start.
    Returns word (uint32, 0xFFFFFFFF = none), weight (float32), node (uint32), valid (uint8) per descriptor."""
    blob = np.frombuffer(blob, np.uint8)
    n = len(desc)
    k = params["m_k"]
    nbits = int(np.ceil(np.log2(k)))
    bs, fo, co, dwp = params["block_size_bytes_wp"], params["feature_off_start"], params["child_off_start"], params["desc_size_bytes_wp"]
    word = np.full(n, 0xFFFFFFFF, np.uint32)
    weight = np.zeros(n, f32)
    node = np.zeros(n, np.uint32)
    valid = np.zeros(n, np.uint8)
    block = np.zeros(n, np.int64)
    cur = np.zeros(n, np.uint32)
    alive = np.ones(n, bool)
    lvl = 0
    D = np.unpackbits(desc, axis=1)                                                # [n,256] bits
    while alive.any():
        idx = np.nonzero(alive)[0]
        base = block[idx] * bs
        N = blob[base].astype(np.int64) | (blob[base + 1].astype(np.int64) << 8)    # uint16 N at offset 0
        feats = blob[(base + fo)[:, None, None] + (np.arange(k) * dwp)[None, :, None] + np.arange(32)[None, None, :]]   # [m,k,32]
        dist = (np.unpackbits(feats, axis=2) != D[idx][:, None, :]).sum(2)          # Hamming, [m,k]
        dist = np.where(np.arange(k)[None, :] < N[:, None], dist, 1 << 30)          # only the block's N valid children compete
        best = dist.argmin(1)                                                       # FIRST minimum
        if lvl == level:
            node[idx] = cur[idx]; valid[idx] = 1
        info = base + co + best * 8
        idc = (blob[info].astype(np.uint32) | (blob[info + 1].astype(np.uint32) << 8) | (blob[info + 2].astype(np.uint32) << 16)
               | (blob[info + 3].astype(np.uint32) << 24))
        wraw = np.ascontiguousarray(np.stack([blob[info + 4], blob[info + 5], blob[info + 6], blob[info + 7]], 1)).view(f32)[:, 0]
        leaf = (idc & np.uint32(0x80000000)) != 0
        ident = idc & np.uint32(0x7FFFFFFF)
        li = idx[leaf]
        word[li] = ident[leaf]; weight[li] = wraw[leaf]
        if lvl < level:
            node[li] = cur[li]; valid[li] = 1
        go = ~leaf
        gi = idx[go]
        block[gi] = ident[go].astype(np.int64)
        cur[gi] = (cur[gi] << np.uint32(nbits)) | best[go].astype(np.uint32)
        alive[:] = False
        alive[gi[ident[go] != 0]] = True                                            # the loop ends on a leaf or on child id 0
        lvl += 1
    return word, weight, node, valid
