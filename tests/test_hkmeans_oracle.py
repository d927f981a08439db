"""Pins the restated hierarchical k-means index (oracle/knn_oracle.cpp, second part) against the REAL xflann compiled from the
reference (oracle/_ref/libxflann_ref.so): the serialised block data byte for byte, the search rows element for element —
for the configuration FrameMatcher_Flann uses (HKMeansParams(32,0), nn=10, maxChecks=16, unsorted; framematcher.cpp:213,239)
and around it."""
import numpy as np
import pytest

import oracle_lib
import synth


def _sets():
    rng = np.random.default_rng(5)
    out = {}
    out["match2000"] = synth.match_set(300, 2000, seed=1)
    out["match10000"] = synth.match_set(200, 10000, seed=2)
    out["ties"] = synth.tie_stress_set(200, 1500, seed=3)
    out["tiny5"] = (rng.integers(0, 256, (5, 32), dtype=np.uint8), rng.integers(0, 256, (20, 32), dtype=np.uint8))
    out["one"] = (rng.integers(0, 256, (1, 32), dtype=np.uint8), rng.integers(0, 256, (4, 32), dtype=np.uint8))
    out["k_plus_1"] = (rng.integers(0, 256, (33, 32), dtype=np.uint8), rng.integers(0, 256, (40, 32), dtype=np.uint8))
    low = np.zeros((900, 32), np.uint8)
    low[:, :2] = rng.integers(0, 256, (900, 2))          # 16 random bits: many equal distances, few (<= 32) exact duplicates
    out["low_entropy"] = (low, low[rng.integers(0, 900, 150)].copy())
    return out


@pytest.mark.parametrize("name", list(_sets().keys()))
def test_hkmeans_restatement_matches_real_xflann(oracle, name):
    ref = oracle_lib.load_ref("xflann")
    if ref is None or not hasattr(ref, "xflann_ref_hkmeans_stream"):
        pytest.skip("oracle/_ref/libxflann_ref.so not built (reference tree absent)")
    a, b = _sets()[name]
    train, queries = (b, a) if name.startswith(("match", "ties")) else (a, b)   # synth returns (train, query) for match sets
    if name.startswith(("match", "ties")):
        train, queries = a, b
    for k, mi in ((32, 0), (8, 0), (32, 11), (8, 3), (8, -1)):
        blob = oracle_lib.hkmeans_blob(oracle, train, k, mi)
        if isinstance(blob, int):
            assert blob == -2      # > k identical rows: the reference recurses without end; nothing to compare
            continue
        stream = oracle_lib.ref_hkmeans_stream(ref, train, k, mi)
        params = np.frombuffer(stream[24:64].tobytes(), np.uint32)
        assert params[0] == 8 and params[7] == 32 and params[8] == len(train)      # alignment, descriptor size, npoints
        assert int(np.frombuffer(stream[40:48].tobytes(), np.uint64)[0]) == len(blob)
        assert stream[64:].tobytes() == blob.tobytes()
        for nn, mc, srt in ((10, 16, 0), (10, 16, 1), (5, 1, 0), (3, 40, 0), (10, 200, 1), (2, 3, 0), (1, 2, 0), (10, -1, 0)):
            i0, d0 = oracle_lib.hkmeans_search(oracle, blob, queries, nn, mc, srt)
            i1, d1 = oracle_lib.ref_hkmeans_search(ref, train, queries, nn, k, mi, mc, srt)
            np.testing.assert_array_equal(i0, i1, err_msg=f"{name} k={k} maxIters={mi} nn={nn} maxChecks={mc} sorted={srt}")
            np.testing.assert_array_equal(d0, d1)


GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "hkmeans_golden.npz")


@pytest.mark.parametrize("case", ["rand", "low_entropy", "tiny", "k_plus_1"])
def test_hkmeans_oracle_and_host_build_match_committed_golden(oracle, case):
    """Same comparison against vectors recorded from the real xflann (tests/golden/make_hkmeans_golden.py): works where
    /root/reference and oracle/_ref do not exist.  The product's host-side build is checked against the same bytes."""
    import hashlib

    from ucoslam_cv3_amd.knn import kmeans_build_host

    g = np.load(GOLD)
    train, q = g[f"{case}_train"], g[f"{case}_q"]
    for k in (32, 8):
        blob = oracle_lib.hkmeans_blob(oracle, train, k, 0)
        assert len(blob) == int(g[f"{case}_k{k}_blob_size"][0])
        assert hashlib.sha256(blob.tobytes()).digest() == g[f"{case}_k{k}_blob_sha256"].tobytes()
        if f"{case}_k{k}_blob" in g.files:
            assert blob.tobytes() == g[f"{case}_k{k}_blob"].tobytes()
        assert kmeans_build_host(train, k).tobytes() == blob.tobytes()
        for nn, mc, s in ((10, 16, 0), (10, 16, 1), (5, 1, 0), (3, 40, 0), (2, 3, 0)):
            i, d = oracle_lib.hkmeans_search(oracle, blob, q, nn, mc, s)
            np.testing.assert_array_equal(i, g[f"{case}_k{k}_nn{nn}_mc{mc}_s{s}_idx"])
            np.testing.assert_array_equal(d, g[f"{case}_k{k}_nn{nn}_mc{mc}_s{s}_dist"])
