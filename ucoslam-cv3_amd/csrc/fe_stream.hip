// FrameExtractor::toStream / fromStream (src/utils/frameextractor.cpp:651-884 / :886-1136, signature 1923123) around the HIP ORB
// extractor — the block a reference `.slm` checkpoint holds for its extractor.  Pure host code.
//
// Layout, in the order the reference writes it (token-pasted source read through the preprocessor; members named by what
// FrameExtractor::setParams assigns to them, :1295-1314):
//   u64   1923123
//   Feature2DSerializable::toStream                    -> uh_orb_to_stream (feature2dserializable.cpp:76-113)
//   u32   frame counter
//   bool  removeFromMarkers, detectMarkers, detectKeyPoints          (one byte each)
//   f32   marker size                                   (Params::aruco_markerSize)
//   FeatParams (20 bytes, raw struct)
//   f32   maxDescDistance
//   aruco::MarkerDetector::toStream                     u64 13213 + MarkerDetector::Params::toStream (3rdparty/aruco/aruco/markerdetector.cpp:256-277;
//                                                        note its `ts` field is written with sizeof(pyrfactor) = 4 bytes) + dictionary string
//   ucoslam::Params::toStream                           u64 9837138769928 ... u64 1837138769921 (src/ucoslamtypes.cpp:63-121)
// The two trailing sub-streams belong to subsystems this repository does not replace (ArUco detection, System parameters): they
// are PARSED (their lengths are only known by walking their fields) and handed back / written verbatim; with no bytes given the
// writer emits what the reference's DEFAULT-constructed objects write.  The HIP extractor detects no markers: a stream whose
// detectMarkers flag is set is refused unless the caller says it runs the marker detector itself.
#include <cfloat>

#include "common.hpp"

namespace {

struct Reader {
    const uint8_t* p; uint64_t n, at = 0; bool ok = true;
    template <class T> T get() { T v{}; if (at + sizeof(T) > n) { ok = false; return v; } std::memcpy(&v, p + at, sizeof(T)); at += sizeof(T); return v; }
    void skip(uint64_t k) { if (at + k > n) ok = false; else at += k; }
    void str() { const uint32_t s = get<uint32_t>(); if (ok) skip(s); }
};
struct Writer {
    std::vector<uint8_t> b;
    template <class T> void put(T v) { const uint8_t* q = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), q, q + sizeof(T)); }
    void str(const char* s) { const uint32_t n = (uint32_t)std::strlen(s); put(n); b.insert(b.end(), s, s + n); }
    void raw(const void* q, size_t n) { const uint8_t* c = static_cast<const uint8_t*>(q); b.insert(b.end(), c, c + n); }
};

// aruco::MarkerDetector::toStream -> length of the sub-stream at r.at (0 on failure)
bool walk_aruco(Reader& r) {
    if (r.get<uint64_t>() != 13213ull) return false;             // MarkerDetector_Impl::toStream
    r.skip(4 * 6);                                               // detectMode maxThreads borderDistThres lowResMarkerSize minSize minSize_pix
    r.skip(1);                                                   // enclosedMarker
    r.skip(4 * 7);                                               // thresMethod NAttemptsAutoThresFix AdaptiveThresWindowSize ThresHold ..._range markerWarpPixSize cornerRefinementM
    r.skip(1);                                                   // autoSize
    r.skip(4 * 3);                                               // ts (4 bytes) error_correction_rate trackingMinDetections
    r.str();                                                     // dictionary
    return r.ok;
}
void default_aruco(Writer& w) {   // a default-constructed aruco::MarkerDetector (markerdetector.h:162-195)
    w.put<uint64_t>(13213ull);
    w.put<int32_t>(0); w.put<int32_t>(1); w.put<float>(0.015f); w.put<int32_t>(20); w.put<float>(-1.f); w.put<int32_t>(-1);
    w.put<uint8_t>(0);
    w.put<int32_t>(0); w.put<int32_t>(3); w.put<int32_t>(-1); w.put<int32_t>(7); w.put<int32_t>(0); w.put<int32_t>(5); w.put<int32_t>(0);
    w.put<uint8_t>(0);
    w.put<float>(0.25f); w.put<float>(0.f); w.put<int32_t>(0);
    w.str("ALL_DICTS");
}
bool walk_params(Reader& r) {   // ucoslam::Params::toStream (ucoslamtypes.cpp:63-121)
    if (r.get<uint64_t>() != 9837138769928ull) return false;
    r.skip(1 + 1 + 4 + 4 + 1 + 4);          // detectMarkers detectKeyPoints targetFocus KFMinConfidence KPNonMaximaSuppresion maxNewPoints
    r.skip(1 + 1 + 4 + 4 + 4 + 4 + 4 + 4 + 4);   // forceInit removeKpIntoMarkers maxDescDistance baseline_ratio markerSize projDistThr nthreads maxVisibleFramesPerMarker minNumProjPoints
    r.skip(4 + 4 + 4 + 4 + 4 + 1);          // KFCulling thRefRatio maxFeatures nOctaveLevels scaleFactor kpDescriptorType (int8)
    r.skip(4 + 4 + 1 + 4 + 1 + 4 + 4);      // aruco_minerrratio_valid aruco_minNumFramesRequired allowOneFrameInit minBaseLine runSequential markersOptWeight minMarkersForMaxWeight
    r.str(); r.str(); r.str(); r.str();     // global_optimizer aruco_Dictionary aruco_DetectionMode aruco_CornerRefimentMethod
    r.skip(4 + 4 + 1 + 1 + 1 + 1);          // aruco_minMarkerSize kptImageScaleFactor autoAdjustKpSensitivity reLocKeyPoints reLocMarkers inPlaneMarkers
    r.str();                                // extraParams
    return r.ok && r.get<uint64_t>() == 1837138769921ull && r.ok;
}
void default_params(Writer& w, const uh_frame_extractor_state& st) {
    // ucoslam::Params() (ucoslamtypes.cpp:23-53, ucoslamtypes.h:90-153) with the members the extractor state fixes set consistently
    w.put<uint64_t>(9837138769928ull);
    w.put<uint8_t>(st.detect_markers); w.put<uint8_t>(st.detect_keypoints); w.put<float>(-1.f); w.put<float>(0.6f); w.put<uint8_t>(0); w.put<int32_t>(350);
    w.put<uint8_t>(0); w.put<uint8_t>(st.remove_from_markers); w.put<float>(st.max_desc_distance); w.put<float>(0.01f); w.put<float>(st.marker_size); w.put<int32_t>(15);
    w.put<int32_t>(2); w.put<int32_t>(10); w.put<int32_t>(3);
    w.put<float>(0.8f); w.put<float>(0.9f); w.put<int32_t>(st.feat_params.maxFeatures); w.put<int32_t>(st.feat_params.nOctaveLevels); w.put<float>(st.feat_params.scaleFactor); w.put<int8_t>(1);
    w.put<float>(3.f); w.put<int32_t>(3); w.put<uint8_t>(0); w.put<float>(0.07f); w.put<uint8_t>(0); w.put<float>(0.5f); w.put<int32_t>(5);
    w.str("g2o"); w.str("ARUCO_MIP_36h12"); w.str("DM_NORMAL"); w.str("CORNER_SUBPIX");
    w.put<float>(0.f); w.put<float>(1.f); w.put<uint8_t>(0); w.put<uint8_t>(1); w.put<uint8_t>(1); w.put<uint8_t>(0);
    w.str("");
    w.put<uint64_t>(1837138769921ull);
}

}  // namespace

extern "C" {

int uh_frame_extractor_to_stream(const uh_orb* orb, const char* str_params, const uh_frame_extractor_state* st, const uint8_t* aruco_stream,
                                 uint64_t aruco_bytes, const uint8_t* params_stream, uint64_t params_bytes, uint8_t* out, uint64_t cap, uint64_t* size) {
    UH_REQUIRE(orb && st && size, "uh_frame_extractor_to_stream: NULL argument");
    UH_REQUIRE(!st->detect_markers || aruco_stream, "uh_frame_extractor_to_stream: detectMarkers is set but no marker-detector stream is given "
               "(the HIP extractor detects no markers; a host that runs the ArUco detector itself passes the detector's own stream)");
    Writer w;
    w.put<uint64_t>(1923123ull);
    uint64_t fs = 0;
    int rc = uh_orb_to_stream(orb, str_params, nullptr, 0, &fs);
    if (rc) return rc;
    const size_t at = w.b.size();
    w.b.resize(at + fs);
    if ((rc = uh_orb_to_stream(orb, str_params, w.b.data() + at, fs, &fs))) return rc;
    w.put<uint32_t>(st->counter);
    w.put<uint8_t>(st->remove_from_markers ? 1 : 0); w.put<uint8_t>(st->detect_markers ? 1 : 0); w.put<uint8_t>(st->detect_keypoints ? 1 : 0);
    w.put<float>(st->marker_size);
    w.raw(&st->feat_params, sizeof(uh_feat_params));
    w.put<float>(st->max_desc_distance);
    if (aruco_stream) {
        Reader r{aruco_stream, aruco_bytes};
        UH_REQUIRE(walk_aruco(r) && r.at == aruco_bytes, "uh_frame_extractor_to_stream: the marker-detector bytes are not an aruco::MarkerDetector stream");
        w.raw(aruco_stream, aruco_bytes);
    } else default_aruco(w);
    if (params_stream) {
        Reader r{params_stream, params_bytes};
        UH_REQUIRE(walk_params(r) && r.at == params_bytes, "uh_frame_extractor_to_stream: the parameter bytes are not a ucoslam::Params stream");
        w.raw(params_stream, params_bytes);
    } else default_params(w, *st);
    *size = w.b.size();
    if (!out) return UH_OK;
    if (cap < *size) { uh::set_error("uh_frame_extractor_to_stream: %llu bytes needed, capacity %llu", (unsigned long long)*size, (unsigned long long)cap); return UH_ECAPACITY; }
    std::memcpy(out, w.b.data(), w.b.size());
    return UH_OK;
}

int uh_frame_extractor_from_stream(uh_orb* orb, const uint8_t* data, uint64_t nbytes, int allow_markers, uh_frame_extractor_state* st, char* str_params_out,
                                   uint64_t str_cap, uint64_t* aruco_off, uint64_t* aruco_bytes, uint64_t* params_off, uint64_t* params_bytes, uint64_t* consumed) {
    UH_REQUIRE(orb && data && st, "uh_frame_extractor_from_stream: NULL argument");
    Reader r{data, nbytes};
    UH_REQUIRE(r.get<uint64_t>() == 1923123ull && r.ok, "void ucoslam::FrameExtractor::fromStream(std::istream&)invalid signature");   // frameextractor.cpp:922
    uint64_t used = 0;
    int rc = uh_orb_from_stream(orb, data + r.at, nbytes - r.at, str_params_out, str_cap, &used);
    if (rc) return rc;
    r.skip(used);
    uh_frame_extractor_state s{};
    s.counter = r.get<uint32_t>();
    s.remove_from_markers = r.get<uint8_t>(); s.detect_markers = r.get<uint8_t>(); s.detect_keypoints = r.get<uint8_t>();
    s.marker_size = r.get<float>();
    if (r.at + sizeof(uh_feat_params) <= nbytes) std::memcpy(&s.feat_params, data + r.at, sizeof(uh_feat_params));
    r.skip(sizeof(uh_feat_params));
    s.max_desc_distance = r.get<float>();
    UH_REQUIRE(r.ok, "uh_frame_extractor_from_stream: truncated stream");
    const uint64_t a0 = r.at;
    UH_REQUIRE(walk_aruco(r), "MarkerDetector_Impl::fromStream invalid signature");   // the reference's message for a foreign block
    const uint64_t p0 = r.at;
    UH_REQUIRE(walk_params(r), "uh_frame_extractor_from_stream: Invalid signature (ucoslam::Params) or truncated stream");
    UH_REQUIRE(allow_markers || !s.detect_markers, "uh_frame_extractor_from_stream: the stream asks for marker detection (detectMarkers = true); the HIP extractor "
               "detects keypoints only — pass allow_markers if the host runs the ArUco detector itself");
    // the FeatParams the reference runs the extractor with (FrameExtractor keeps its own copy beside the extractor's)
    if ((rc = uh_orb_set_params(orb, &s.feat_params))) return rc;
    *st = s;
    if (aruco_off) *aruco_off = a0;
    if (aruco_bytes) *aruco_bytes = p0 - a0;
    if (params_off) *params_off = p0;
    if (params_bytes) *params_bytes = r.at - p0;
    if (consumed) *consumed = r.at;
    return UH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------
// System::saveToFile / readFromFile (src/utils/system.cpp:8099-8720; token-pasted source, statement starts) — the `.slm` checkpoint:
//   u64 182312 | Map::toStream | Params::toStream | se3 current pose (6 x f32, raw) | i64 current keyframe | bool initialised |
//   STATE (i32) | MODES (i32) | Frame current | Frame previous | FrameExtractor::toStream | MapManager::toStream |
//   cv::Mat (io_utils.cpp:21-37: i32 rows, cols, type + rows x cols x elemSize bytes) | i64 | u64
// The hot path owns two of these blocks: Params (the thresholds and extractor settings every stage of the path is driven by) and
// the FrameExtractor block (uh_frame_extractor_*).  Map / Frame / MapManager are the host's own containers (SURVEY.md §2 row 18,
// out of scope): they carry no length, so only the host's own fromStream knows where they end — the reader below is therefore
// SECTIONED (the host reads its blocks, hands the remaining bytes back) and the writer takes those blocks as opaque bytes.  A
// block that is missing is refused by name, never guessed.
namespace {

size_t cv_elem_size(int32_t type) {   // CV_ELEM_SIZE: depth = type & 7, channels = (type >> 3) + 1
    static const int depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return (size_t)depth_bytes[type & 7] * (size_t)(((type >> 3) & 511) + 1);
}

bool read_params(Reader& r, uh_params_view* v) {   // ucoslamtypes.cpp:123-180, field for field
    if (r.get<uint64_t>() != 9837138769928ull) return false;
    uh_params_view p{};
    p.detect_markers = r.get<uint8_t>(); p.detect_keypoints = r.get<uint8_t>(); p.target_focus = r.get<float>(); p.kf_min_confidence = r.get<float>();
    p.kp_non_maxima_suppression = r.get<uint8_t>(); p.max_new_points = r.get<int32_t>();
    p.force_initialization_from_markers = r.get<uint8_t>(); p.remove_keypoints_into_markers = r.get<uint8_t>();
    p.max_desc_distance = r.get<float>(); p.baseline_median_depth_ratio_min = r.get<float>(); p.aruco_marker_size = r.get<float>();
    p.proj_dist_thr = r.get<int32_t>(); p.nthreads_feature_detector = r.get<int32_t>(); p.max_visible_frames_per_marker = r.get<int32_t>();
    p.min_num_proj_points = r.get<int32_t>(); p.kf_culling = r.get<float>(); p.th_ref_ratio = r.get<float>();
    p.max_features = r.get<int32_t>(); p.n_octave_levels = r.get<int32_t>(); p.scale_factor = r.get<float>(); p.kp_descriptor_type = r.get<int8_t>();
    r.skip(4 + 4 + 1);                       // aruco_minerrratio_valid aruco_minNumFramesRequired aruco_allowOneFrameInitialization
    p.min_base_line = r.get<float>(); p.run_sequential = r.get<uint8_t>();
    r.skip(4 + 4);                           // markersOptWeight minMarkersForMaxWeight
    {   // global_optimizer
        const uint32_t n = r.get<uint32_t>();
        if (!r.ok || r.at + n > r.n) return false;
        const uint32_t k = n < sizeof(p.global_optimizer) - 1 ? n : (uint32_t)sizeof(p.global_optimizer) - 1;
        std::memcpy(p.global_optimizer, r.p + r.at, k);
        r.skip(n);
    }
    r.str(); r.str(); r.str();               // aruco_Dictionary aruco_DetectionMode aruco_CornerRefimentMethod
    r.skip(4);                               // aruco_minMarkerSize
    p.kpt_image_scale_factor = r.get<float>(); p.auto_adjust_kp_sensitivity = r.get<uint8_t>();
    p.relocalization_with_keypoints = r.get<uint8_t>(); p.relocalization_with_markers = r.get<uint8_t>(); r.skip(1);   // inPlaneMarkers
    r.str();                                 // extraParams
    // "read until end signature" (:176-179): the reference slides an 8-byte window forward 8 bytes at a time
    uint64_t sig = 0;
    while (r.ok && sig != 1837138769921ull) sig = r.get<uint64_t>();
    if (!r.ok) return false;
    if (v) *v = p;
    return true;
}

}  // namespace

extern "C" {

int uh_params_from_stream(const uint8_t* data, uint64_t nbytes, uh_params_view* out, uint64_t* consumed) {
    UH_REQUIRE(data && out, "uh_params_from_stream: NULL argument");
    Reader r{data, nbytes};
    const uint64_t sig = nbytes >= 8 ? r.get<uint64_t>() : 0;
    UH_REQUIRE(sig == 9837138769928ull, "Invalid signature");   // ucoslamtypes.cpp:126
    r.at = 0;
    UH_REQUIRE(read_params(r, out), "Reached EOF without finding end signature");   // :179
    if (consumed) *consumed = r.at;
    return UH_OK;
}

int uh_system_stream_begin(const uint8_t* data, uint64_t nbytes, uint64_t* map_offset) {
    UH_REQUIRE(data && map_offset, "uh_system_stream_begin: NULL argument");
    uint64_t sig = 0;
    if (nbytes >= 8) std::memcpy(&sig, data, 8);
    UH_REQUIRE(sig == 182312ull, "void ucoslam::System::readFromFile(std::string)invalid file type:");   // system.cpp:8446
    *map_offset = 8;   // Map::fromStream is the host's (it carries no length: only its reader knows where it ends)
    return UH_OK;
}

int uh_system_stream_state(const uint8_t* data, uint64_t nbytes, uh_params_view* params, uh_system_state* state, uint64_t* params_bytes, uint64_t* consumed) {
    UH_REQUIRE(data && params && state, "uh_system_stream_state: NULL argument (data = the bytes behind the host's Map block)");
    Reader r{data, nbytes};
    UH_REQUIRE(nbytes >= 8 && [&] { uint64_t s; std::memcpy(&s, data, 8); return s == 9837138769928ull; }(),
               "uh_system_stream_state: no ucoslam::Params block here (Invalid signature) — pass the bytes that FOLLOW Map::fromStream");
    UH_REQUIRE(read_params(r, params), "Reached EOF without finding end signature");
    if (params_bytes) *params_bytes = r.at;
    uh_system_state s{};
    for (int i = 0; i < 6; i++) s.cur_pose_rt[i] = r.get<float>();
    s.current_keyframe = r.get<int64_t>();
    s.is_initialized = r.get<uint8_t>();
    s.state = r.get<int32_t>();
    s.mode = r.get<int32_t>();
    UH_REQUIRE(r.ok, "uh_system_stream_state: truncated stream");
    UH_REQUIRE((s.state == 0 || s.state == 1) && (s.mode == 0 || s.mode == 1), "uh_system_stream_state: STATE %d / MODES %d are not values of the reference's enums", s.state, s.mode);
    *state = s;
    if (consumed) *consumed = r.at;   // the current Frame's block starts here (host code)
    return UH_OK;
}

int uh_system_stream_tail(const uint8_t* data, uint64_t nbytes, uh_system_tail* tail, uint64_t* consumed) {
    UH_REQUIRE(data && tail, "uh_system_stream_tail: NULL argument (data = the bytes behind the host's MapManager block)");
    Reader r{data, nbytes};
    uh_system_tail t{};
    t.mat_rows = r.get<int32_t>(); t.mat_cols = r.get<int32_t>(); t.mat_type = r.get<int32_t>();
    UH_REQUIRE(r.ok && t.mat_rows >= 0 && t.mat_cols >= 0 && t.mat_rows <= 65536 && t.mat_cols <= 65536, "uh_system_stream_tail: no cv::Mat header here — pass the bytes that FOLLOW MapManager::fromStream");
    t.mat_data_offset = r.at;
    t.mat_data_bytes = (uint64_t)t.mat_rows * t.mat_cols > 0 ? (uint64_t)t.mat_rows * t.mat_cols * cv_elem_size(t.mat_type) : 0;   // io_utils.cpp:46: r*c > 0
    r.skip(t.mat_data_bytes);
    t.last_value_i64 = r.get<int64_t>();
    t.last_value_u64 = r.get<uint64_t>();
    UH_REQUIRE(r.ok, "uh_system_stream_tail: truncated stream");
    *tail = t;
    if (consumed) *consumed = r.at;
    return UH_OK;
}

int uh_system_to_stream(const uh_system_parts* parts, uint8_t* out, uint64_t cap, uint64_t* size) {
    UH_REQUIRE(parts && size, "uh_system_to_stream: NULL argument");
    UH_REQUIRE(parts->map && parts->map_bytes, "uh_system_to_stream: the Map block is missing — Map::toStream is host code (SURVEY.md section 2 row 18), pass its bytes");
    UH_REQUIRE(parts->cur_frame && parts->cur_frame_bytes && parts->prev_frame && parts->prev_frame_bytes,
               "uh_system_to_stream: a Frame block is missing — Frame::toStream is host code, pass the bytes of the current and the previous frame");
    UH_REQUIRE(parts->map_manager && parts->map_manager_bytes, "uh_system_to_stream: the MapManager block is missing — MapManager::toStream is host code, pass its bytes");
    UH_REQUIRE(parts->extractor && parts->extractor_bytes >= 8, "uh_system_to_stream: the FrameExtractor block is missing (uh_frame_extractor_to_stream writes it)");
    {
        uint64_t sig; std::memcpy(&sig, parts->extractor, 8);
        UH_REQUIRE(sig == 1923123ull, "uh_system_to_stream: the extractor bytes are not a FrameExtractor stream (signature 1923123)");
    }
    Writer w;
    w.put<uint64_t>(182312ull);
    w.raw(parts->map, parts->map_bytes);
    if (parts->params) {
        Reader r{parts->params, parts->params_bytes};
        UH_REQUIRE(walk_params(r) && r.at == parts->params_bytes, "uh_system_to_stream: the parameter bytes are not a ucoslam::Params stream");
        w.raw(parts->params, parts->params_bytes);
    } else {   // the Params block the extractor block carries (FrameExtractor::toStream ends with one)
        Reader r{parts->extractor, parts->extractor_bytes};
        r.skip(8);
        // Feature2DSerializable stream: u64 sig, u64 type, string, FeatParams
        r.skip(16); r.str(); r.skip(sizeof(uh_feat_params));
        r.skip(4 + 3 + 4 + sizeof(uh_feat_params) + 4);
        UH_REQUIRE(r.ok && walk_aruco(r), "uh_system_to_stream: cannot walk the extractor block to its Params (pass parts->params)");
        const uint64_t p0 = r.at;
        UH_REQUIRE(walk_params(r), "uh_system_to_stream: cannot walk the extractor block's Params (pass parts->params)");
        w.raw(parts->extractor + p0, r.at - p0);
    }
    for (int i = 0; i < 6; i++) w.put<float>(parts->state.cur_pose_rt[i]);
    w.put<int64_t>(parts->state.current_keyframe);
    w.put<uint8_t>(parts->state.is_initialized ? 1 : 0);
    w.put<int32_t>(parts->state.state);
    w.put<int32_t>(parts->state.mode);
    w.raw(parts->cur_frame, parts->cur_frame_bytes);
    w.raw(parts->prev_frame, parts->prev_frame_bytes);
    w.raw(parts->extractor, parts->extractor_bytes);
    w.raw(parts->map_manager, parts->map_manager_bytes);
    const uint64_t mat_bytes = (uint64_t)parts->tail.mat_rows * parts->tail.mat_cols > 0 ? (uint64_t)parts->tail.mat_rows * parts->tail.mat_cols * cv_elem_size(parts->tail.mat_type) : 0;
    UH_REQUIRE(mat_bytes == 0 || parts->mat_data, "uh_system_to_stream: the cv::Mat of the tail has %llu bytes of data but mat_data is NULL", (unsigned long long)mat_bytes);
    w.put<int32_t>(mat_bytes ? parts->tail.mat_rows : 0); w.put<int32_t>(mat_bytes ? parts->tail.mat_cols : 0); w.put<int32_t>(mat_bytes ? parts->tail.mat_type : 0);   // io_utils.cpp:23-28: an empty Mat writes 0 0 0
    if (mat_bytes) w.raw(parts->mat_data, mat_bytes);
    w.put<int64_t>(parts->tail.last_value_i64);
    w.put<uint64_t>(parts->tail.last_value_u64);
    *size = w.b.size();
    if (!out) return UH_OK;
    if (cap < *size) { uh::set_error("uh_system_to_stream: %llu bytes needed, capacity %llu", (unsigned long long)*size, (unsigned long long)cap); return UH_ECAPACITY; }
    std::memcpy(out, w.b.data(), w.b.size());
    return UH_OK;
}

}  // extern "C"
