"""On-disk / stream formats touching the path: Feature2DSerializable::toStream/fromStream for the ORB extractor
(feature2dserializable.cpp:76-113, ORBextractor.cpp:417-423) and xflann::Index::toStream/fromStream for the k-means index
(index.cpp:153-188, kmeansindex.cpp:209-229) against bytes written by the REAL xflann (tests/golden/hkmeans_stream_golden.npz)."""
import os
import struct

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hkmeans_stream_golden.npz")


@pytest.mark.gpu
def test_extractor_stream_round_trip_and_errors(hip_ctx):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.orb import FeatParams, ORBextractor

    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(1234, 6, 1.3, nthreads=2, sensitivity=0.25)
    check = u._lib.check
    check(u.lib().uh_orb_set_params(ext._h, fp))
    s = ext.toStream("some=params")
    # layout: sig, type tag F2D_ORB, u32 length + string, raw FeatParams {nthreads, maxFeatures, nOctaveLevels, scaleFactor, sensitivity}
    assert s[:16] == struct.pack("<QQ", 1828374733, 0) and s[16:20] == struct.pack("<I", 11) and s[20:31] == b"some=params"
    assert s[31:] == struct.pack("<iiiff", 2, 1234, 6, np.float32(1.3), np.float32(0.25)) and len(s) == 51
    ext2, sp, used = ORBextractor.fromStream(hip_ctx, s + b"trailing bytes of the enclosing .slm stream")
    g = ext2.getParams()
    assert sp == "some=params" and used == 51
    assert (g.nthreads, g.maxFeatures, g.nOctaveLevels) == (2, 1234, 6) and g.scaleFactor == np.float32(1.3) and g.sensitivity == np.float32(0.25)
    assert ext2.toStream("some=params") == s
    with pytest.raises(u.UcoslamHipError, match="signature error"):
        ORBextractor.fromStream(hip_ctx, struct.pack("<Q", 1828374734) + s[8:])
    with pytest.raises(u.UcoslamHipError, match="not F2D_ORB"):
        ORBextractor.fromStream(hip_ctx, s[:8] + struct.pack("<Q", 3) + s[16:])        # F2D_GRID_ORB
    with pytest.raises(u.UcoslamHipError, match="truncated"):
        ORBextractor.fromStream(hip_ctx, s[:40])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["s40", "s300"])
def test_kmeans_index_stream_equals_real_xflann_bytes(hip_ctx, case):
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.knn import Index

    g = np.load(GOLD)
    train, q = g[f"{case}_train"], torch.from_numpy(g[f"{case}_q"]).cuda()
    for k in (32, 8):
        ref_stream = g[f"{case}_k{k}_stream"].tobytes()
        built = Index(hip_ctx).build_kmeans(train, k, 0)
        assert built.toStream() == ref_stream, "uh_knn_to_stream differs from xflann::Index::toStream"
        loaded = Index(hip_ctx).fromStream(ref_stream)                      # a stream written by the real library
        assert loaded.toStream() == ref_stream
        for idx in (built, loaded):
            i, d = idx.search_kmeans(q, 10, 16, sorted=False)
            np.testing.assert_array_equal(i.cpu().numpy(), g[f"{case}_k{k}_idx"])
            np.testing.assert_array_equal(d.cpu().numpy(), g[f"{case}_k{k}_dist"])
    ref_stream = g[f"{case}_k32_stream"].tobytes()
    with pytest.raises(u.UcoslamHipError, match="Invalid signature"):
        Index(hip_ctx).fromStream(b"\0" * 8 + ref_stream[8:])
    with pytest.raises(u.UcoslamHipError, match="type of implementation"):
        Index(hip_ctx).fromStream(ref_stream[:8] + b"\x01" * 8 + ref_stream[16:])
    with pytest.raises(u.UcoslamHipError, match="truncated"):
        Index(hip_ctx).fromStream(ref_stream[: len(ref_stream) // 2])
    with pytest.raises(u.UcoslamHipError):
        Index(hip_ctx).build(train).toStream()                               # Linear: no stream form (reference: "Not yet")
