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
