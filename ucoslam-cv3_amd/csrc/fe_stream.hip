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
