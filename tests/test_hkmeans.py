"""Hierarchical k-means index (what FrameMatcher_Flann builds and searches, framematcher.cpp:213,239): the product's host-side
build must produce the reference's block data byte for byte, the GPU search the reference's rows element for element."""
import numpy as np
import pytest

import oracle_lib
from test_hkmeans_oracle import _sets


@pytest.mark.parametrize("name", list(_sets().keys()))
def test_host_build_equals_oracle_blob(oracle, name):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import kmeans_build_host

    train = _sets()[name][0]
    for k in (32, 8, 2):
        for mi in (0, 1, 11, -1):      # HKMeansParams maxIters: the matcher's 0, the library default 11, "until convergence"
            ref = oracle_lib.hkmeans_blob(oracle, train, k, mi)
            if isinstance(ref, int):
                assert ref == -2
                with pytest.raises(u.UcoslamHipError):
                    kmeans_build_host(train, k, mi)
                continue
            assert kmeans_build_host(train, k, mi).tobytes() == ref.tobytes(), (k, mi)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_sets().keys()))
def test_hip_hkmeans_search_matches_oracle(hip_ctx, oracle, name):
    import torch
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    train, queries = _sets()[name]
    for k in (32, 8):
        ref_blob = oracle_lib.hkmeans_blob(oracle, train, k, 0)
        idx = Index(hip_ctx)
        if isinstance(ref_blob, int):
            with pytest.raises(u.UcoslamHipError):
                idx.build_kmeans(train, k, 0)
            continue
        idx.build_kmeans(train, k, 0)
        assert idx.kmeans_blob().tobytes() == ref_blob.tobytes()
        for nn, mc, srt in ((10, 16, 0), (10, 16, 1), (5, 1, 0), (3, 40, 0), (2, 3, 0), (1, 2, 0), (10, -1, 0), (64, 30, 0)):
            ri, rd = oracle_lib.hkmeans_search(oracle, ref_blob, queries, nn, mc, srt)
            gi, gd = idx.search_kmeans(queries, nn, mc, bool(srt))
            np.testing.assert_array_equal(gi, ri, err_msg=f"{name} k={k} nn={nn} maxChecks={mc} sorted={srt}")
            np.testing.assert_array_equal(gd, rd)
        # the library's default parameters (k-means rounds) on the same data
        for mi in (11, -1):
            rb = oracle_lib.hkmeans_blob(oracle, train, k, mi)
            if isinstance(rb, int):
                continue
            idx2 = Index(hip_ctx).build_kmeans(train, k, mi)
            assert idx2.kmeans_blob().tobytes() == rb.tobytes()
            ri, rd = oracle_lib.hkmeans_search(oracle, rb, queries, 10, 16, 0)
            gi, gd = idx2.search_kmeans(queries, 10, 16, False)
            np.testing.assert_array_equal(gi, ri)
            np.testing.assert_array_equal(gd, rd)
        # device-resident queries
        qd = torch.from_numpy(queries).cuda()
        gi, gd = idx.search_kmeans(qd, 10, 16, False)
        ri, rd = oracle_lib.hkmeans_search(oracle, ref_blob, queries, 10, 16, 0)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(gi.cpu().numpy(), ri)
        np.testing.assert_array_equal(gd.cpu().numpy(), rd)


@pytest.mark.gpu
def test_hip_hkmeans_errors(hip_ctx):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    rng = np.random.default_rng(1)
    train = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (10, 32), dtype=np.uint8)
    idx = Index(hip_ctx)
    with pytest.raises(u.UcoslamHipError):          # not built: loud, like Index::_search (index.cpp:82-85)
        idx.search_kmeans(q, 10, 16)
    with pytest.raises(u.UcoslamHipError):
        idx.build_kmeans(train, 32, -5)
    idx.build_kmeans(train, 32, 0)
    for nn, mc in ((1, 1), (2, 2), (2, 1)):          # the reference's greedy shortcuts
        with pytest.raises(u.UcoslamHipError):
            idx.search_kmeans(q, nn, mc)
    i, d = idx.search_kmeans(q[:0], 10, 16)
    assert i.shape == (0, 10)
