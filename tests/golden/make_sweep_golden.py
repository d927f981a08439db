"""Generates tests/golden/sweep_golden.npz: OUTPUTS of the real reference code (oracle/_ref/*.so, compiled from /root/reference by
oracle/Makefile) at the sizes BASELINE.json names and at the window sizes the launch-chain BA forms serve — the inputs are NOT stored:
they are regenerated from tests/synth.py seeds, and a SHA-256 of the regenerated input bytes is stored beside every output so that a
drift of synth.py is noticed instead of showing up as a parity failure.

  ba_<K>x<P>_s<seed>_f<nfix>_*   real g2o (both passes of GlobalOptimizerG2O::optimize, globaloptimizer_g2o.cpp:418-464): se3 state (K x 7 doubles),
                                 iteration counts, SHA-256 + count of the bad-association flags, sum of chi2
  knn_<nq>x<nt>_nn10_s<0|1>_*    real xflann Linear (impl/linear.h:68-86): SHA-256 of the index rows and of the distance rows
  hk_<nq>x<nt>_nn10_mc16_*       real xflann HKMeans(32, 0) + search maxChecks 16 (what FrameMatcher_Flann runs)
  kd_2000_*                      real picoflann radiusSearch hits IN ORDER on a 2000-keypoint frame (basictypes/picoflann.h:238-345)
  pnp_<n>_s<seed>_*              real g2o pose-only optimisation (pnpsolver.cpp:116-409): state, per-round iterations, flags

Run in the build container only:  python tests/golden/make_sweep_golden.py   (reproduces the committed file bit for bit; it says so)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

BA_CASES = [(10, 3000, 0, 2), (10, 3000, 1, 2), (10, 3000, 2, 2), (10, 3000, 3, 2),      # bench.py's four problems (rank 0)
            (18, 3000, 16, 2), (19, 3000, 17, 2), (26, 1100, 57, 2), (34, 450, 65, 2), (34, 3000, 32, 2), (50, 450, 81, 2), (66, 400, 97, 2)]
KNN_CASES = [(2000, 10000, 0), (8000, 10000, 50)]
PNP_CASES = [(800, 3), (1300, 4), (3001, 5)]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def ba_input_sha(pr):
    return sha(*[pr[k] for k in ("poses", "fixed", "intr", "points", "obs_pt", "obs_kf", "obs_uv", "obs_w")])


def pnp_input_sha(pr):
    return sha(*[pr[k] for k in ("pose", "intr", "p3d", "kp", "invsig", "weight")])


def kd_queries(xy, nq=400, seed=5):
    rng = np.random.default_rng(seed)
    q = np.where(rng.random((nq, 1)) < 0.5, xy[rng.integers(len(xy), size=nq)], (rng.random((nq, 2)) * [1241, 376])).astype(np.float32)
    r = np.float32([4.0, 15.0, 15.0 * 1.2 ** 3, 15.0 * 1.2 ** 7 * 1.6])[rng.integers(4, size=nq)]
    return q, r


def main():
    g2o, xf, pf = oracle_lib.load_ref("g2o"), oracle_lib.load_ref("xflann"), oracle_lib.load_ref("picoflann")
    assert g2o is not None and xf is not None and pf is not None, "build oracle/_ref first (make -C oracle ref)"
    P = oracle_lib.P
    out = {}
    for K, Pn, seed, nfix in BA_CASES:
        pr = synth.ba_problem(K, Pn, seed, nfixed=nfix)
        r = oracle_lib.ba_optimize_ref(g2o, pr, 5)
        key = f"ba_{K}x{Pn}_s{seed}_f{nfix}"
        out[key + "_in_sha"] = ba_input_sha(pr)
        out[key + "_state"] = r["state"]; out[key + "_iters"] = r["iters"]
        out[key + "_bad_sha"] = sha(r["bad"]); out[key + "_nbad"] = np.int64(r["bad"].sum()); out[key + "_chi2_sum"] = np.float64(r["chi2"].sum())
        print(key, "E", pr["E"], "iters", r["iters"].tolist(), "bad", int(r["bad"].sum()))
    for nq, nt, seed in KNN_CASES:
        train, q = synth.match_set(nq, nt, seed=seed)
        for s in (0, 1):
            idx = np.empty((nq, 10), np.int32); dist = np.empty((nq, 10), np.int32)
            assert xf.xflann_ref_linear_search(P(train), nt, P(q), nq, 10, s, 1, P(idx), P(dist)) == 0
            key = f"knn_{nq}x{nt}_nn10_s{s}"
            out[key + "_in_sha"] = sha(train, q); out[key + "_idx_sha"] = sha(idx); out[key + "_dist_sha"] = sha(dist)
            out[key + "_first_rows"] = idx[:4].copy()
            print(key, idx[0].tolist())
    train, q = synth.match_set(2000, 10000, seed=0)
    idx, dist = oracle_lib.ref_hkmeans_search(xf, train, q, 10, 32, 0, 16, 0)
    out["hk_2000x10000_nn10_mc16_in_sha"] = sha(train, q); out["hk_2000x10000_nn10_mc16_idx_sha"] = sha(idx); out["hk_2000x10000_nn10_mc16_dist_sha"] = sha(dist)
    print("hk", idx[0].tolist())
    fr, mp, pose = synth.proj_problem(2000, 3000, 0)
    xy = np.stack([fr["und_kpts"]["x"], fr["und_kpts"]["y"]], 1).astype(np.float32)
    kd = oracle_lib.KdOracle(pf, "picoflann_ref", xy)
    qs, rs = kd_queries(xy)
    hits, offs = [], [0]
    for (qx, qy), r in zip(qs, rs):
        i, d = kd.radius(qx, qy, r)
        hits.append(i); offs.append(offs[-1] + len(i))
    out["kd_2000_in_sha"] = sha(xy, qs, rs); out["kd_2000_off"] = np.array(offs, np.int64); out["kd_2000_idx_sha"] = sha(np.concatenate(hits).astype(np.uint32))
    print("kd hits", offs[-1])
    for n, seed in PNP_CASES:
        pr = synth.pnp_problem(n, seed=seed)
        r = oracle_lib.pnp_solve_ref(g2o, pr)
        key = f"pnp_{n}_s{seed}"
        out[key + "_in_sha"] = pnp_input_sha(pr); out[key + "_state"] = r["state"]; out[key + "_iters"] = r["iters"]
        out[key + "_bad_sha"] = sha(r["bad"]); out[key + "_ngood"] = np.int32(r["ngood"])
        print(key, r["iters"].tolist(), r["ngood"])
    path = os.path.join(HERE, "sweep_golden.npz")
    if os.path.exists(path):
        old = np.load(path)
        diff = [k for k in out if k not in old.files or not np.array_equal(old[k], out[k])]
        print("sweep_golden.npz:", "regenerated bit for bit" if not diff and len(old.files) == len(out) else f"DIFFERS from the committed file in {diff}")
    np.savez_compressed(path, **out)
    print("wrote", path, sum(np.asarray(v).nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
