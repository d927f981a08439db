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


# ------------------------------------------------------------------------------------------------ FrameExtractor (.slm extractor block)
def _str(s: bytes) -> bytes:
    return struct.pack("<I", len(s)) + s


def _ref_aruco_stream(detect_mode=0, min_size=-1.0, dictionary=b"ALL_DICTS", corner=0) -> bytes:
    """aruco::MarkerDetector::toStream, field by field as 3rdparty/aruco/aruco/markerdetector.cpp:256-277 writes it behind MarkerDetector_Impl's
    u64 13213 (defaults: markerdetector.h:162-195; `ts` goes out with sizeof(pyrfactor) = 4 bytes)."""
    return (struct.pack("<Q", 13213) + struct.pack("<iifi", detect_mode, 1, np.float32(0.015), 20) + struct.pack("<fi", np.float32(min_size), -1) + struct.pack("<B", 0)
            + struct.pack("<iiiiiii", 0, 3, -1, 7, 0, 5, corner) + struct.pack("<B", 0) + struct.pack("<ffi", np.float32(0.25), 0.0, 0) + _str(dictionary))


def _ref_params_stream(detect_markers=1, detect_kp=1, remove_kp_in_markers=1, max_desc=np.finfo(np.float32).max, marker_size=1.0, max_features=4000,
                       n_levels=8, scale=1.2, run_sequential=0, extra=b"") -> bytes:
    """ucoslam::Params::toStream, field by field as src/ucoslamtypes.cpp:63-121 writes it (defaults: Params::Params() :23-53 and ucoslamtypes.h:90-153)."""
    f = np.float32
    return (struct.pack("<Q", 9837138769928) + struct.pack("<BBffBi", detect_markers, detect_kp, f(-1), f(0.6), 0, 350)
            + struct.pack("<BBfffiiii", 0, remove_kp_in_markers, f(max_desc), f(0.01), f(marker_size), 15, 2, 10, 3)
            + struct.pack("<ffiifb", f(0.8), f(0.9), max_features, n_levels, f(scale), 1)
            + struct.pack("<fiBfBfi", f(3), 3, 0, f(0.07), run_sequential, f(0.5), 5)
            + _str(b"g2o") + _str(b"ARUCO_MIP_36h12") + _str(b"DM_NORMAL") + _str(b"CORNER_SUBPIX")
            + struct.pack("<ffBBBB", f(0), f(1), 0, 1, 1, 0) + _str(extra) + struct.pack("<Q", 1837138769921))


def _ref_frame_extractor_stream(feat_stream: bytes, counter, flags, marker_size, fp, max_desc, aruco: bytes, params: bytes) -> bytes:
    """FrameExtractor::toStream (src/utils/frameextractor.cpp:651-884)."""
    return (struct.pack("<Q", 1923123) + feat_stream + struct.pack("<I", counter) + struct.pack("<BBB", *flags) + struct.pack("<f", np.float32(marker_size))
            + struct.pack("<iiiff", *fp) + struct.pack("<f", np.float32(max_desc)) + aruco + params)


@pytest.mark.gpu
def test_frame_extractor_stream_against_a_writer_that_follows_the_reference(hip_ctx):
    """FrameExtractor::toStream / fromStream (sig 1923123): bytes equal to a field-by-field writer of the cited lines, for the marker-less
    configuration (default ArUco / Params sub-streams) and for a stream carrying the host's own detector and parameter blocks; round trips;
    the reference's error behaviour; the marker refusal."""
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd.orb import FeatParams, FrameExtractorState, ORBextractor

    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(2000, 8, 1.2, nthreads=2)
    u._lib.check(u.lib().uh_orb_set_params(ext._h, fp))
    feat = ext.toStream("")
    # ---- marker-less: the sub-streams are what default-constructed reference objects write (with the members the state fixes)
    st = FrameExtractorState(17, 0, 0, 1, np.float32(1.0), fp, np.float32(50.0))
    got = ext.frameExtractorToStream(st)
    want = _ref_frame_extractor_stream(feat, 17, (0, 0, 1), 1.0, (2, 2000, 8, np.float32(1.2), np.float32(0)), 50.0, _ref_aruco_stream(),
                                       _ref_params_stream(detect_markers=0, detect_kp=1, remove_kp_in_markers=0, max_desc=50.0, marker_size=1.0, max_features=2000))
    assert got == want
    ext2, st2, sp, aru, par, used = ORBextractor.frameExtractorFromStream(hip_ctx, got + b"the rest of the .slm stream")
    assert used == len(got) and sp == "" and aru == _ref_aruco_stream() and par == want[len(want) - len(par):]
    assert (st2.counter, st2.remove_from_markers, st2.detect_markers, st2.detect_keypoints) == (17, 0, 0, 1)
    assert st2.marker_size == np.float32(1.0) and st2.max_desc_distance == np.float32(50.0) and st2.feat_params.maxFeatures == 2000
    g = ext2.getParams()
    assert (g.nthreads, g.maxFeatures, g.nOctaveLevels) == (2, 2000, 8) and g.scaleFactor == np.float32(1.2)
    assert ext2.frameExtractorToStream(st2, sp, aru, par) == got
    # ---- a checkpoint of a marker-using session: the host's detector / parameter blocks travel verbatim; reading needs allow_markers
    aruco = _ref_aruco_stream(detect_mode=1, min_size=0.02, dictionary=b"ARUCO_MIP_36h12", corner=1)
    params = _ref_params_stream(run_sequential=1, extra=b"some extra=1")
    ref = _ref_frame_extractor_stream(feat, 5, (1, 1, 1), 0.12, (2, 2000, 8, np.float32(1.2), np.float32(0)), 50.0, aruco, params)
    with pytest.raises(u.UcoslamHipError, match="marker detection"):
        ORBextractor.frameExtractorFromStream(hip_ctx, ref)
    ext3, st3, sp3, aru3, par3, used3 = ORBextractor.frameExtractorFromStream(hip_ctx, ref, allow_markers=True)
    assert used3 == len(ref) and aru3 == aruco and par3 == params and st3.detect_markers == 1 and st3.marker_size == np.float32(0.12)
    assert ext3.frameExtractorToStream(st3, sp3, aru3, par3) == ref
    with pytest.raises(u.UcoslamHipError, match="no marker-detector stream"):
        ext3.frameExtractorToStream(st3)
    # ---- the reference's refusals
    with pytest.raises(u.UcoslamHipError, match="invalid signature"):
        ORBextractor.frameExtractorFromStream(hip_ctx, struct.pack("<Q", 1923124) + got[8:])
    with pytest.raises(u.UcoslamHipError, match="signature error"):
        ORBextractor.frameExtractorFromStream(hip_ctx, got[:8] + struct.pack("<Q", 7) + got[16:])
    with pytest.raises(u.UcoslamHipError):
        ORBextractor.frameExtractorFromStream(hip_ctx, got[: len(got) - 9])                     # the Params end signature is cut off
    bad = bytearray(got)
    bad[len(got) - len(par) - len(aru)] ^= 1                                                       # MarkerDetector_Impl's signature
    with pytest.raises(u.UcoslamHipError, match="MarkerDetector_Impl"):
        ORBextractor.frameExtractorFromStream(hip_ctx, bytes(bad))
    with pytest.raises(u.UcoslamHipError, match="not a ucoslam::Params stream"):
        ext.frameExtractorToStream(st, "", None, params[:-3])


# ------------------------------------------------------------------------------------------------ System (.slm) sections owned by the path
def _ref_system_stream(map_b, params_b, pose, cur_kf, init, state, mode, f_cur, f_prev, fe_b, mm_b, mat, last_i64, last_u64) -> bytes:
    """System::saveToFile (src/utils/system.cpp:8099-8720, statement order) with the host's blocks as given bytes; cv::Mat as io_utils.cpp:21-37."""
    if mat is None or mat.size == 0:
        mat_b = struct.pack("<iii", 0, 0, 0)
    else:
        cv_type = {np.dtype(np.uint8): 0, np.dtype(np.float32): 5, np.dtype(np.float64): 6}[mat.dtype]
        mat_b = struct.pack("<iii", mat.shape[0], mat.shape[1], cv_type) + np.ascontiguousarray(mat).tobytes()
    return (struct.pack("<Q", 182312) + map_b + params_b + struct.pack("<6f", *pose) + struct.pack("<q", cur_kf) + struct.pack("<B", init) + struct.pack("<ii", state, mode)
            + f_cur + f_prev + fe_b + mm_b + mat_b + struct.pack("<q", last_i64) + struct.pack("<Q", last_u64))


def test_params_stream_and_system_sections_on_the_host():
    """Params::fromStream and the sectioned reader of the System stream need no GPU: checked against field-by-field writers of the cited
    reference lines, including the reference's error messages and its 'read until the end signature' loop."""
    import ucoslam_cv3_amd as u
    from ucoslam_cv3_amd import slm

    par = _ref_params_stream(detect_markers=0, detect_kp=1, remove_kp_in_markers=0, max_desc=50.0, marker_size=0.25, max_features=2000, n_levels=8, scale=1.2,
                             run_sequential=1, extra=b"k=v")
    pv, used = slm.params_from_stream(par + b"tail")
    assert used == len(par)
    assert (pv.detect_markers, pv.detect_keypoints, pv.run_sequential, pv.kp_descriptor_type) == (0, 1, 1, 1)
    assert pv.max_desc_distance == np.float32(50.0) and pv.proj_dist_thr == 15 and pv.nthreads_feature_detector == 2
    assert (pv.max_features, pv.n_octave_levels) == (2000, 8) and pv.scale_factor == np.float32(1.2) and pv.aruco_marker_size == np.float32(0.25)
    assert pv.global_optimizer == b"g2o" and pv.max_new_points == 350 and pv.kf_min_confidence == np.float32(0.6) and pv.kf_culling == np.float32(0.8)
    # a newer writer may append fields before the end signature: the reader slides forward in 8-byte steps like the reference's loop
    grown = par[:-8] + b"\x01" * 16 + par[-8:]
    assert slm.params_from_stream(grown)[1] == len(grown)
    with pytest.raises(u.UcoslamHipError, match="Invalid signature"):
        slm.params_from_stream(b"\0" * 8 + par[8:])
    with pytest.raises(u.UcoslamHipError, match="Reached EOF"):
        slm.params_from_stream(par[:-8])
    # ---- the System stream, sectioned
    rng = np.random.default_rng(5)
    blk = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    map_b, f_cur, f_prev, mm_b, fe_b = blk(1234), blk(333), blk(77), blk(901), struct.pack("<Q", 1923123) + blk(40)
    pose = [0.1, -0.2, 0.3, 1.0, 2.0, 3.0]
    mat = rng.random((4, 4)).astype(np.float32)
    ref = _ref_system_stream(map_b, par, pose, 42, 1, 0, 1, f_cur, f_prev, fe_b, mm_b, mat, -1, 9)
    off = slm.system_stream_begin(ref)
    assert off == 8
    after_map = ref[off + len(map_b):]                                   # (the host's Map::fromStream consumed its block)
    pv2, st, pbytes, used = slm.system_stream_state(after_map)
    assert pbytes == len(par) and used == len(par) + 24 + 8 + 1 + 8
    assert list(st.cur_pose_rt) == [np.float32(v) for v in pose] and (st.current_keyframe, st.is_initialized, st.state, st.mode) == (42, 1, 0, 1)
    assert pv2.max_features == 2000
    after_mm = after_map[used + len(f_cur) + len(f_prev) + len(fe_b) + len(mm_b):]
    tail, tused = slm.system_stream_tail(after_mm)
    assert tused == len(after_mm) and (tail.mat_rows, tail.mat_cols, tail.mat_type) == (4, 4, 5) and tail.mat_data_bytes == 64
    assert after_mm[tail.mat_data_offset: tail.mat_data_offset + 64] == mat.tobytes() and (tail.last_value_i64, tail.last_value_u64) == (-1, 9)
    # ---- the writer composes the same bytes from the host's blocks; an empty matrix writes 0 0 0
    t = slm.SystemTail(4, 4, 5, 0, 0, -1, 9)
    got = slm.system_to_stream(map_b, st, f_cur, f_prev, fe_b, mm_b, t, mat.tobytes(), params=par)
    assert got == ref
    t0 = slm.SystemTail(0, 0, 0, 0, 0, 7, 0)
    got0 = slm.system_to_stream(map_b, st, f_cur, f_prev, fe_b, mm_b, t0, params=par)
    assert got0 == _ref_system_stream(map_b, par, pose, 42, 1, 0, 1, f_cur, f_prev, fe_b, mm_b, None, 7, 0)
    # ---- refusals: foreign file, blocks that are the host's, misplaced sections
    with pytest.raises(u.UcoslamHipError, match="invalid file type"):
        slm.system_stream_begin(struct.pack("<Q", 182313) + ref[8:])
    with pytest.raises(u.UcoslamHipError, match="Map block is missing"):
        slm.system_to_stream(b"", st, f_cur, f_prev, fe_b, mm_b, t, mat.tobytes(), params=par)
    with pytest.raises(u.UcoslamHipError, match="Frame block is missing"):
        slm.system_to_stream(map_b, st, f_cur, b"", fe_b, mm_b, t, mat.tobytes(), params=par)
    with pytest.raises(u.UcoslamHipError, match="MapManager block is missing"):
        slm.system_to_stream(map_b, st, f_cur, f_prev, fe_b, b"", t, mat.tobytes(), params=par)
    with pytest.raises(u.UcoslamHipError, match="not a FrameExtractor stream"):
        slm.system_to_stream(map_b, st, f_cur, f_prev, blk(48), mm_b, t, mat.tobytes(), params=par)
    with pytest.raises(u.UcoslamHipError, match="FOLLOW Map::fromStream"):
        slm.system_stream_state(ref[off:])                                # the Map block has not been consumed
    with pytest.raises(u.UcoslamHipError, match="not values of the reference's enums"):
        slm.system_stream_state(par + struct.pack("<6f", *pose) + struct.pack("<q", 1) + b"\x01" + struct.pack("<ii", 5, 0))


@pytest.mark.gpu
def test_system_checkpoint_with_the_real_extractor_block(hip_ctx):
    """A whole `.slm` stream around the HIP extractor's own FrameExtractor block: params = NULL writes the Params block the extractor block
    ends with; reading the sections back restores the extractor and the state."""
    from ucoslam_cv3_amd import slm
    from ucoslam_cv3_amd.orb import FeatParams, FrameExtractorState, ORBextractor
    import ucoslam_cv3_amd as u

    ext = ORBextractor.create(hip_ctx)
    fp = FeatParams(2000, 8, 1.2, nthreads=2)
    u._lib.check(u.lib().uh_orb_set_params(ext._h, fp))
    fe_b = ext.frameExtractorToStream(FrameExtractorState(3, 0, 0, 1, np.float32(1.0), fp, np.float32(50.0)))
    rng = np.random.default_rng(9)
    blk = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    map_b, f_cur, f_prev, mm_b = blk(500), blk(60), blk(61), blk(62)
    st = slm.SystemState((C_float6 := (0.0, 0.0, 0.0, 0.5, 0.25, 0.125)), 7, 1, 0, 0)
    got = slm.system_to_stream(map_b, st, f_cur, f_prev, fe_b, mm_b, slm.SystemTail(0, 0, 0, 0, 0, -1, 0))
    par = _ref_params_stream(detect_markers=0, detect_kp=1, remove_kp_in_markers=0, max_desc=50.0, marker_size=1.0, max_features=2000)
    assert got == _ref_system_stream(map_b, par, C_float6, 7, 1, 0, 0, f_cur, f_prev, fe_b, mm_b, None, -1, 0)
    pv, st2, pbytes, used = slm.system_stream_state(got[8 + len(map_b):])
    assert pv.max_features == 2000 and pv.max_desc_distance == np.float32(50.0) and st2.current_keyframe == 7
    rest = got[8 + len(map_b) + used + len(f_cur) + len(f_prev):]
    ext2, st_fe, sp, aru, parb, fe_used = ORBextractor.frameExtractorFromStream(hip_ctx, rest)
    assert fe_used == len(fe_b) and st_fe.counter == 3 and ext2.getParams().maxFeatures == 2000
    tail, tused = slm.system_stream_tail(rest[fe_used + len(mm_b):])
    assert tused == 12 + 16 and tail.mat_data_bytes == 0 and tail.last_value_i64 == -1
