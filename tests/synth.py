"""Seeded synthetic inputs shared by tests, golden generators and bench.py."""
import numpy as np


def match_set(nq=2000, nt=10000, seed=0, flip_p=0.08, matched_frac=0.7):
    """Train = random 256-bit rows; 70 % of queries = a train row with Binomial(256,0.08) bit flips, 30 % random."""
    rng = np.random.default_rng(seed)
    train = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    if nt > 0:
        perm = rng.integers(0, nt, nq)
        matched = rng.random(nq) < matched_frac
        flips = (rng.random((nq, 256)) < flip_p)
        flip_bytes = np.packbits(flips, axis=1, bitorder="little")
        q[matched] = train[perm[matched]] ^ flip_bytes[matched]
    return train, q


def tie_stress_set(nq=200, nt=1500, seed=1, ndistinct=5):
    """Many duplicate / near-duplicate rows so equal distances are the norm (heap-order stress)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (ndistinct, 32), dtype=np.uint8)
    train = base[rng.integers(0, ndistinct, nt)].copy()
    train[:, 0] ^= rng.integers(0, 4, nt).astype(np.uint8)
    q = base[rng.integers(0, ndistinct, nq)].copy()
    q[:, 1] ^= rng.integers(0, 2, nq).astype(np.uint8)
    return train, q


def descending_set(nq=8, nt=700, seed=2):
    """Train rows ordered by DEcreasing distance to the queries: every push is accepted (worst case)."""
    rng = np.random.default_rng(seed)
    q0 = rng.integers(0, 256, 32, dtype=np.uint8)
    bits = np.unpackbits(q0, bitorder="little")
    rows = []
    for i in range(nt):
        nflip = max(0, 255 - (i * 255) // max(nt - 1, 1))
        b = bits.copy()
        b[:nflip] ^= 1
        rows.append(np.packbits(b, bitorder="little"))
    train = np.stack(rows).astype(np.uint8)
    q = np.tile(q0, (nq, 1))
    q[:, 31] ^= np.arange(nq, dtype=np.uint8)
    return train, q


def frame(w=1241, h=376, seed=0, nshapes=3000, shift=(0, 0)):
    """Synthetic CV_8UC1 frame: low-frequency gradient + ~3000 random bright/dark rectangles and
    discs (contrast 30..120) + Gaussian noise sigma 3, clipped. `shift` translates the scene (KITTI-like stream)."""
    rng = np.random.default_rng(1234)            # the SCENE is fixed; `seed` only drives the noise
    big_w, big_h = w + 256, h + 256
    yy, xx = np.mgrid[0:big_h, 0:big_w].astype(np.float32)
    img = 110 + 40 * np.sin(xx / 211.0) + 30 * np.cos(yy / 97.0)
    for _ in range(nshapes):
        cx, cy = rng.integers(0, big_w), rng.integers(0, big_h)
        sx, sy = rng.integers(3, 28), rng.integers(3, 28)
        c = float(rng.integers(30, 121)) * (1 if rng.random() < 0.5 else -1)
        x0, x1 = max(cx - sx, 0), min(cx + sx, big_w)
        y0, y1 = max(cy - sy, 0), min(cy + sy, big_h)
        if rng.random() < 0.6:
            img[y0:y1, x0:x1] += c
        else:
            r = min(sx, sy)
            sub = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
            img[y0:y1, x0:x1] += c * sub
    ox, oy = 128 + int(shift[0]), 128 + int(shift[1])
    out = img[oy:oy + h, ox:ox + w]
    noise = np.random.default_rng(seed).normal(0, 3, out.shape).astype(np.float32)
    return np.clip(np.rint(out + noise), 0, 255).astype(np.uint8)


def _se3_exp(d):
    w, u = np.asarray(d[:3], float), np.asarray(d[3:], float)
    th = np.linalg.norm(w)
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-5:
        R = np.eye(3) + O + 0.5 * O @ O
        V = np.eye(3) + 0.5 * O + O @ O / 6
    else:
        R = np.eye(3) + np.sin(th) / th * O + (1 - np.cos(th)) / th**2 * O @ O
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * O + (th - np.sin(th)) / th**3 * O @ O
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ u
    return T


def _scale_f32(octave, scale_factor=1.2):
    """Frame::scaleFactors[octave]: the float chain 1, 1.2f, 1.2f*1.2f, ... (ORBextractor.cpp:468-475)."""
    s = np.float32(1.0)
    for _ in range(octave):
        s = np.float32(s * np.float32(scale_factor))
    return s


def ba_problem(K=10, P=3000, seed=0, nfixed=2, outlier_frac=0.02, pose_noise=0.01, point_noise=0.05, pix_noise=0.5,
               w=1241, h=376, double_weights=False):
    """Synthetic local BA: K keyframes on a line (baseline 0.3 m), KITTI intrinsics, P points in front,
    observations where in-frame, octave uniform 0..7 -> information 1.2^-octave, first `nfixed` keyframes fixed.
    double_weights=True: the information scalars as full doubles 1 / 1.2^octave (not float-exact, unlike the reference's
    (double)(float) values) — the form tests/golden/ba_golden.npz was recorded with; it exercises the 24-byte observation records."""
    rng = np.random.default_rng(seed)
    fx = fy = 718.856
    cx, cy = 607.19, 185.22
    Tgt = []
    for k in range(K):
        T = np.eye(4)
        T[:3, 3] = -np.array([0.3 * k, 0.02 * np.sin(k), 0.0])
        Tgt.append(_se3_exp(np.r_[0.0, 0.02 * np.sin(0.7 * k), 0.0, 0, 0, 0]) @ T)
    z = rng.uniform(4, 40, P)
    X = np.stack([(rng.uniform(0, w, P) - cx) / fx * z + 0.3 * K / 2, (rng.uniform(0, h, P) - cy) / fy * z, z], 1)
    obs_pt, obs_kf, obs_uv, obs_w = [], [], [], []
    for p in range(P):
        for k in range(K):
            pc = Tgt[k][:3, :3] @ X[p] + Tgt[k][:3, 3]
            if pc[2] <= 0.5:
                continue
            u, v = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
            if 0 <= u < w and 0 <= v < h and rng.random() < 0.9:
                octave = int(rng.integers(0, 8))
                noise = rng.normal(0, pix_noise, 2)
                if rng.random() < outlier_frac:
                    noise += rng.normal(0, 25, 2)
                obs_pt.append(p); obs_kf.append(k); obs_uv.append([u + noise[0], v + noise[1]]); obs_w.append(1.0 / (1.2 ** octave) if double_weights else float(np.float32(1.0 / float(_scale_f32(octave)))))   # (double)(float)(1./f): the reference's vector<float> _InvScaleFactors
    poses = []
    fixed = np.zeros(K, np.uint8)
    fixed[:nfixed] = 1
    for k in range(K):
        T = Tgt[k] if fixed[k] else _se3_exp(rng.normal(0, pose_noise, 6)) @ Tgt[k]
        poses.append(T.astype(np.float32))
    # points need >= 2 observations (globaloptimizer_g2o.cpp:140); drop the others
    obs_pt = np.array(obs_pt, np.int32)
    cnt = np.bincount(obs_pt, minlength=P)
    keep_pt = cnt >= 2
    remap = -np.ones(P, np.int32)
    remap[keep_pt] = np.arange(keep_pt.sum(), dtype=np.int32)
    sel = keep_pt[obs_pt]
    Xn = (X + rng.normal(0, point_noise, X.shape))[keep_pt]
    return dict(
        K=K, P=int(keep_pt.sum()), E=int(sel.sum()),
        poses=np.ascontiguousarray(np.stack(poses).reshape(K, 16)), fixed=fixed,
        intr=np.tile(np.array([fx, fy, cx, cy], np.float32), (K, 1)),
        points=np.ascontiguousarray(Xn.astype(np.float32)),
        obs_pt=np.ascontiguousarray(remap[obs_pt[sel]]), obs_kf=np.ascontiguousarray(np.array(obs_kf, np.int32)[sel]),
        obs_uv=np.ascontiguousarray(np.array(obs_uv, np.float32)[sel]), obs_w=np.ascontiguousarray(np.array(obs_w, np.float64)[sel]),
        poses_gt=np.stack(Tgt),
    )


def vocabulary(k=10, depth=3, seed=0, aligment=8):
    """Synthetic fbow vocabulary in the reference's binary block format (fbow.h:137-197, sizes from fbow.cpp:10-49):
    a complete k-ary tree of `depth` levels; node descriptors = parent descriptor with random bit flips (so that descents are
    meaningful), leaves get increasing word ids and random idf-like weights.  Returns (params120 bytes, blob bytes, meta)."""
    import struct

    rng = np.random.default_rng(seed)
    nblocks = sum(k ** l for l in range(depth))
    desc_wp = -(-32 // aligment) * aligment
    feature_off = -(-8 // aligment) * aligment
    child_off = feature_off + k * desc_wp
    block_size = -(-(feature_off + k * (desc_wp + 8)) // aligment) * aligment
    total = block_size * nblocks
    blob = np.zeros(total, np.uint8)
    # blocks in BFS order: block b's children blocks are first_child[b] + c
    parent_desc = {0: rng.integers(0, 256, 32, dtype=np.uint8)}
    next_block, word = 1, 0
    level_of = {0: 0}
    for b in range(nblocks):
        base = b * block_size
        lvl = level_of[b]
        leaf_level = lvl == depth - 1
        n_children = k if not (b % 7 == 3 and leaf_level) else k - 2      # a few ragged blocks (N < k)
        blob[base:base + 2] = np.frombuffer(struct.pack("<H", n_children), np.uint8)
        blob[base + 2:base + 4] = np.frombuffer(struct.pack("<H", 1 if leaf_level else 0), np.uint8)
        for c in range(n_children):
            flips = np.packbits(rng.random(256) < (0.30 / (lvl + 1)), bitorder="little")
            d = parent_desc[b] ^ flips
            if c == 1 and b % 5 == 0:
                d = (parent_desc[b] ^ np.packbits(rng.random(256) < (0.30 / (lvl + 1)), bitorder="little"))
            blob[base + feature_off + c * desc_wp: base + feature_off + c * desc_wp + 32] = d
            info = base + child_off + c * 8
            if leaf_level:
                blob[info:info + 8] = np.frombuffer(struct.pack("<If", 0x80000000 | word, float(np.float32(rng.uniform(0.5, 6.0)))), np.uint8)
                word += 1
            else:
                blob[info:info + 8] = np.frombuffer(struct.pack("<If", next_block, 0.0), np.uint8)
                parent_desc[next_block] = d
                level_of[next_block] = lvl + 1
                next_block += 1
        if n_children >= 2 and b % 4 == 1:      # duplicate child descriptors: the FIRST minimum must win
            blob[base + feature_off + 1 * desc_wp: base + feature_off + 1 * desc_wp + 32] = blob[base + feature_off: base + feature_off + 32]
    params = struct.pack("<50s2xII4xQQQQQiiI4x", b"orb", aligment, nblocks, desc_wp, block_size, feature_off, child_off, total, 0, 32, k)
    assert len(params) == 120
    return params, blob.tobytes(), dict(nblocks=nblocks, nwords=word, block_size=block_size)


def pnp_problem(n=600, seed=0, outlier_frac=0.15, pose_noise=0.03, pix_noise=0.7, w=1241, h=376):
    """Synthetic per-frame pose estimation (PnPSolver::solvePnp inputs): n map points seen by one frame, a perturbed initial
    pose, pixel noise, gross outliers, octave-dependent information and some 'unstable' points with half weight."""
    rng = np.random.default_rng(seed)
    fx = fy = 718.856
    cx, cy = 607.19, 185.22
    Tgt = _se3_exp(np.r_[0.01, -0.02, 0.005, 0, 0, 0]) @ np.eye(4)
    Tgt[:3, 3] = [0.4, -0.1, 0.2]
    z = rng.uniform(4, 40, n)
    Xc = np.stack([(rng.uniform(20, w - 20, n) - cx) / fx * z, (rng.uniform(20, h - 20, n) - cy) / fy * z, z], 1)
    Xw = (Xc - Tgt[:3, 3]) @ Tgt[:3, :3]          # R^T (Xc - t)
    uv = np.stack([fx * Xc[:, 0] / z + cx, fy * Xc[:, 1] / z + cy], 1) + rng.normal(0, pix_noise, (n, 2))
    out = rng.random(n) < outlier_frac
    uv[out] += rng.normal(0, 30, (int(out.sum()), 2))
    octave = rng.integers(0, 8, n)
    invsig = (1.0 / (np.float32(1.2) ** octave.astype(np.float32))).astype(np.float32)
    weight = np.where(rng.random(n) < 0.2, 0.5, 1.0).astype(np.float32)
    T0 = (_se3_exp(rng.normal(0, pose_noise, 6)) @ Tgt).astype(np.float32)
    return dict(pose=np.ascontiguousarray(T0.reshape(16)), intr=np.array([fx, fy, cx, cy], np.float32), n=n,
                p3d=np.ascontiguousarray(Xw.astype(np.float32)), kp=np.ascontiguousarray(uv.astype(np.float32)), invsig=invsig, weight=weight,
                pose_gt=Tgt, outlier=out)


def proj_problem(n_kpts=2000, n_pts=3000, seed=0, w=1241, h=376, n_levels=8, low_entropy=False, pose_noise=0.002):
    """Inputs of Map::matchFrameToMapPoints (map.cpp:651-770): a frame (undistorted keypoints with octaves, descriptors, scale
    factors, intrinsics) and candidate map points (position, mean viewing normal, distance-invariance window, descriptor).
    Two thirds of the map points are back-projections of frame keypoints (descriptor = the keypoint's with a few flipped
    bits), the rest are unrelated (behind the camera, outside the image, wrong scale, ...).  low_entropy=True makes most
    descriptor distances collide so that the best / second-best bookkeeping and its candidate-order dependence are stressed."""
    import oracle_lib

    rng = np.random.default_rng(seed)
    fx = fy = 718.856 if w > 1000 else 517.3
    cx, cy = (607.19, 185.22) if w > 1000 else (318.6, 255.3)
    scale = np.float32(1.2) ** np.arange(n_levels, dtype=np.float32)
    sf = np.ones(n_levels, np.float32)
    for i in range(1, n_levels):
        sf[i] = sf[i - 1] * np.float32(1.2)            # the extractor's float chain (ORBextractor.cpp:468-515)
    kp = np.zeros(n_kpts, oracle_lib.KEYPOINT_DTYPE)
    octv = rng.integers(0, n_levels, n_kpts)
    # level pixel grid positions scaled back to level 0, like the extractor's output (clustered: corners come in groups)
    centers = rng.random((max(n_kpts // 12, 1), 2)) * [w - 60, h - 60] + 30
    pos = centers[rng.integers(0, len(centers), n_kpts)] + rng.normal(0, 9, (n_kpts, 2))
    pos = np.clip(pos, 19, [w - 20, h - 20])
    lv = sf[octv]
    pos = (np.round(pos / lv[:, None]) * lv[:, None]).astype(np.float32)
    kp["x"], kp["y"], kp["octave"] = pos[:, 0], pos[:, 1], octv
    kp["size"] = 31 * lv
    kp["angle"] = rng.random(n_kpts) * 360
    kp["response"] = rng.integers(7, 120, n_kpts)
    if low_entropy:
        desc = np.zeros((n_kpts, 32), np.uint8)
        desc[:, :2] = rng.integers(0, 256, (n_kpts, 2))
    else:
        desc = rng.integers(0, 256, (n_kpts, 32), dtype=np.uint8)
    Tgt = _se3_exp(np.r_[0.01, -0.015, 0.004, 0, 0, 0]) @ np.eye(4)
    Tgt[:3, 3] = [0.3, -0.05, 0.1]
    cam_c = -Tgt[:3, :3].T @ Tgt[:3, 3]
    n_rel = (2 * n_pts) // 3 if n_kpts else 0   # a frame without keypoints: only unrelated map points
    src = rng.integers(0, max(n_kpts, 1), n_rel) if n_kpts else np.zeros(0, np.int64)
    z = rng.uniform(3, 45, n_rel)
    uv = np.stack([kp["x"][src], kp["y"][src]], 1).astype(np.float64) + rng.normal(0, 1.5, (n_rel, 2)) if n_kpts else np.zeros((0, 2))
    Xc = np.stack([(uv[:, 0] - cx) / fx * z, (uv[:, 1] - cy) / fy * z, z], 1)
    Xw = (Xc - Tgt[:3, 3]) @ Tgt[:3, :3]
    # distance-invariance window as Map::updatePointNormalAndDistances makes it: dist * scale[level] and that / scale[last]
    d = np.linalg.norm(Xw - cam_c, axis=1)
    lev = np.clip(kp["octave"][src] + rng.integers(-1, 2, n_rel), 0, n_levels - 1) if n_kpts else np.zeros(0, np.int64)
    maxd = d * sf[lev] * rng.uniform(0.9, 1.1, n_rel)
    mind = maxd / sf[n_levels - 1]
    view = cam_c - Xw
    view /= np.linalg.norm(view, axis=1, keepdims=True)
    nrm = view + rng.normal(0, 0.45, (n_rel, 3))      # some beyond 60 degrees, many between cos 0.98 and 0.5
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    mdesc = desc[src].copy() if n_kpts else np.zeros((0, 32), np.uint8)
    flips = rng.integers(0, 30 if not low_entropy else 3, n_rel)
    for i in range(n_rel):
        for b in rng.integers(0, 256 if not low_entropy else 16, flips[i]):
            mdesc[i, b // 8] ^= 1 << (b % 8)
    n_un = n_pts - n_rel
    Xu = rng.normal(0, 20, (n_un, 3)) + [0, 0, 10]
    du = np.linalg.norm(Xu - cam_c, axis=1)
    maxu = du * rng.uniform(0.5, 3.0, n_un)
    nu = rng.normal(0, 1, (n_un, 3))
    nu /= np.linalg.norm(nu, axis=1, keepdims=True)
    udesc = rng.integers(0, 256, (n_un, 32), dtype=np.uint8) if not low_entropy else np.zeros((n_un, 32), np.uint8)
    perm = rng.permutation(n_pts)
    mp = dict(ids=np.ascontiguousarray((rng.permutation(10 * n_pts)[:n_pts]).astype(np.uint32)),
              pos3d=np.ascontiguousarray(np.concatenate([Xw, Xu])[perm].astype(np.float32)),
              normal=np.ascontiguousarray(np.concatenate([nrm, nu])[perm].astype(np.float32)),
              min_dist=np.ascontiguousarray(np.concatenate([mind, maxu / sf[n_levels - 1]])[perm].astype(np.float32)),
              max_dist=np.ascontiguousarray(np.concatenate([maxd, maxu])[perm].astype(np.float32)),
              desc=np.ascontiguousarray(np.concatenate([mdesc, udesc])[perm]),
              # octave of the previous frame's keypoint that observed the point (tracker's prev-frame search, system.cpp:6150):
              # the source keypoint's level +-1 for related points, anything for the rest (own generator: older streams unchanged)
              octave=np.ascontiguousarray(np.concatenate([lev, np.random.default_rng(seed + 12345).integers(0, n_levels, n_un)])[perm].astype(np.int32)))
    fr = dict(und_kpts=kp, desc=np.ascontiguousarray(desc), scale_factors=sf, fx=fx, fy=fy, cx=cx, cy=cy, min_xy=(0, 0), max_xy=(w, h))
    pose = (_se3_exp(rng.normal(0, pose_noise, 6)) @ Tgt).astype(np.float32)
    return fr, mp, np.ascontiguousarray(pose.reshape(16))


def ba_hard_problem(seed):
    """A local BA that Levenberg-Marquardt does not sail through (rejected trials, lambda factors other than 1/3, passes that end on
    ten rejections in a row): large pose / landmark / pixel noise and 30 % outliers on a small window."""
    rng = np.random.default_rng(seed)
    K = int(rng.integers(3, 12)); P = int(rng.integers(20, 300)); nf = int(rng.integers(1, 3))
    pn = float(rng.choice([0.3, 0.6, 1.0])); ptn = float(rng.choice([2.0, 5.0, 10.0]))
    return ba_problem(K, P, seed=seed, nfixed=nf, pose_noise=pn, point_noise=ptn, outlier_frac=0.3, pix_noise=3.0)
