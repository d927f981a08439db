/*
 * ucoslam_hip.h — C ABI of the MI355X (gfx950) tracking hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface (file:line in lambdaloop/ucoslam-cv3)
 * whose work it replaces.  The C++ adaptors in include/ucoslam_hip/ mirror the
 * reference class surfaces on top of these calls; INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative UH_E* code on failure;
 *     uh_last_error() returns a thread-local message for the last failure.
 *   - "_dev" entry points take DEVICE pointers, enqueue on the context stream
 *     and do not synchronise; the others take HOST pointers, copy in/out and
 *     return when results are in the caller's buffers.
 *   - there is NO CPU fallback: without a usable HIP device every call fails
 *     with UH_ENODEVICE.
 */
#ifndef UCOSLAM_HIP_H
#define UCOSLAM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UH_OK          0
#define UH_EINVAL     -1   /* bad argument (the reference throws std::runtime_error here) */
#define UH_ENODEVICE  -2   /* no HIP device / HIP runtime failure */
#define UH_ENOTBUILT  -3   /* index/vocabulary not built (reference: search() returns false) */
#define UH_ENOMEM     -4
#define UH_ECAPACITY  -5   /* caller's output buffer too small */

typedef struct uh_ctx uh_ctx;

/* Context = one GPU + one HIP stream (+ scratch owned by the stage objects).
 * hip_stream is a hipStream_t; NULL means the device's default (null) stream, which is also
 * what torch.cuda.current_stream().cuda_stream is (0) unless a side stream is current.
 * uh_ctx_create_private() creates and owns a non-blocking stream instead. */
int         uh_ctx_create(int device, void* hip_stream, uh_ctx** out);
int         uh_ctx_create_private(int device, uh_ctx** out);
/* A private stream confined to a share of the GPU's compute units: mask bits [first_bit, first_bit + n_bits) of
 * hipExtStreamCreateWithCUMask — on MI355X bit p is CU p / 8 of XCD p % 8, so a contiguous range takes the same number of CUs from every
 * XCD.  For hosts that run the mapper's local BA (latency-bound, <= 94 workgroups) beside the tracker's wide launches, as the reference
 * runs them on two threads (system.cpp / mapmanager.cpp:150): disjoint ranges keep the two from sharing a CU. */
int         uh_ctx_create_private_cus(int device, int first_bit, int n_bits, uh_ctx** out);
void        uh_ctx_destroy(uh_ctx* ctx);
int         uh_ctx_synchronize(uh_ctx* ctx);
void*       uh_ctx_stream(uh_ctx* ctx);
/* pinned host memory for the buffers handed to the host-pointer entry points (asynchronous DMA instead of staged copies) */
void*       uh_host_alloc(size_t bytes);
void        uh_host_free(void* p);
const char* uh_last_error(void);
/* Per-kernel timing for measurement (bench.py roofline leg): when enabled every kernel launch of the stage objects
 * bound to this context is bracketed by HIP events on the context stream.  report: "<kernel> <calls> <total_ms>\n" lines. */
int         uh_prof_enable(uh_ctx* ctx, int on);
int         uh_prof_reset(uh_ctx* ctx);
int         uh_prof_report(uh_ctx* ctx, char* buf, size_t cap);
/* library/ABI version: major*10000 + minor*100 + patch */
int         uh_version(void);

/* ------------------------------------------------------------------------
 * Hamming brute-force kNN  — replaces xflann::Index (Linear) build/search:
 *   3rdparty/xflann/xflann/index.cpp:45 (build), :77-103 (search + sort),
 *   impl/linear.h:68-86 (_knnsearch<Hamming_x64_32bytes>), impl/resultset.h:64-135
 *   (max-heap ResultSet), index.h:119-134 (exchange sort).
 * Results are bit-identical to the reference INCLUDING the heap array order of
 * unsorted rows and the order of equal distances in sorted rows.
 * Unfilled slots (nt < nn): index -1, distance 0   (linear.h:82-85).
 * ------------------------------------------------------------------------ */
typedef struct uh_knn uh_knn;

int  uh_knn_create(uh_ctx* ctx, uh_knn** out);
void uh_knn_destroy(uh_knn* idx);

/* build: train = nt rows of desc_bytes (must be 32 = ORB) bytes, row stride in bytes.
 * Host pointer; the rows are copied to HBM (reference LinearParams(store_data=1)). */
int uh_knn_build(uh_knn* idx, const uint8_t* train, int nt, size_t stride, int desc_bytes);
/* build from rows already resident in HBM (contiguous nt x 32, 16-byte aligned); not copied. */
int uh_knn_build_dev(uh_knn* idx, const uint8_t* d_train, int nt);
/* restrict the scan to train rows [begin,end) (multi-GPU sharding of the train set);
 * indices stay global.  Default = [0,nt). */
int uh_knn_set_shard(uh_knn* idx, int begin, int end);
int uh_knn_size(const uh_knn* idx);
/* Exact search: how many queries one wave serves (1 = default, 2, 4; rows with nn > 15 always use 1).  Results are identical.  With
 * more queries per wave the search itself takes longer (8000 x 10 000 x nn=10: 163 / 185 / 250 us) but moves 1/2 or 1/4 of the
 * L1/L2 traffic and keeps 1/2 or 1/4 of the waves resident — the setting for a matcher that runs beside latency-bound work on
 * another stream (next to the local BA the whole tracking step is 6 % faster at 4). */
int  uh_knn_set_queries_per_wave(uh_knn* index, int queries_per_wave);

/* search: queries nq x 32 (row stride q_stride bytes), outputs nq x nn int32 row-major.
 * sorted: 0/1 as KnnSearchParams(maxChecks,sorted); max_dist: -1 = kNN, >=0 = radius bound
 * (SearchParams maxDist, resultset.h:66). */
int uh_knn_search(uh_knn* idx, const uint8_t* queries, int nq, size_t q_stride, int nn,
                  int32_t* indices, int32_t* distances, int sorted, int max_dist);
int uh_knn_search_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn,
                      int32_t* d_indices, int32_t* d_distances, int sorted, int max_dist);

/* Hierarchical k-means form of the index — xflann::Index::build(features, HKMeansParams(k, maxIters)) (index.cpp:52-57,
 * impl/kmeansindexcreator.{h,cpp}) and search(..., KnnSearchParams(maxChecks, sorted)) through KMeansIndex::_knnsearch_nn
 * (impl/kmeansindex.h:356-410): the APPROXIMATE index FrameMatcher_Flann builds per train frame (framematcher.cpp:213 k=32,
 * maxIters=0; :239 nn=10, maxChecks=16, unsorted).  Results are bit-identical to the reference: same tree (std::shuffle of a
 * default std::mt19937 picks the centres), same best-bin-first order, same ResultSet rows; unfilled slots index -1, distance 0.
 * build: host pointer, contiguous n x 32 bytes; 2 <= k <= 64; maxIters as HKMeansParams (0 = the matcher's setting, -1 = until
 * convergence; rounds move the centres to the bitwise majority of their clusters).  More than k identical descriptors make
 * the reference recurse without end: reported as UH_EINVAL.  uh_knn_kmeans_blob exposes the block data in the reference's own
 * serialised layout (the bytes KMeansIndex::toStream writes after its 48-byte header).
 * search: maxChecks as the reference (<= 0 returns empty rows, like the reference's loop condition); the pairs (nn=1,maxChecks=1)
 * and (nn=2,maxChecks<=2) select the reference's greedy descents (kmeansindex.h:226-352), which are defective for binary
 * descriptors — their int32 "best distance" is initialised with uint32 max = -1, so no child ever compares smaller and every
 * query returns the same row with distance -1 (observed on the real library) — and are refused here with UH_EINVAL. */
int uh_knn_build_kmeans(uh_knn* idx, const uint8_t* features, int n, int k, int max_iters);
int uh_knn_kmeans_blob(uh_knn* idx, const uint8_t** data, uint64_t* size);
/* host-only build (test hook, no GPU): writes min(cap, size) bytes of the block data to out, the full size to *size */
int uh_knn_kmeans_build_host(const uint8_t* features, int n, int k, int max_iters, uint8_t* out, uint64_t cap, uint64_t* size);
int uh_knn_search_kmeans(uh_knn* idx, const uint8_t* queries, int nq, int nn, int max_checks, int sorted,
                         int32_t* indices, int32_t* distances);
int uh_knn_search_kmeans_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int max_checks, int sorted,
                             int32_t* d_indices, int32_t* d_distances);

/* Sharded search (multi-GPU): phase 1 emits, per query, the candidates the local
 * heap ACCEPTED while scanning this shard, in train-index order (a superset of what
 * the global heap accepts).  cand = nq x cap entries, each (dist<<32 | global index);
 * counts = nq int32 (count > cap means overflow: that row must be rescanned).
 * Phase 2 replays the concatenation of all shards' candidate lists (in shard order)
 * through the exact ResultSet semantics.  lists: nshards blocks of [nq x cap],
 * counts: nshards blocks of [nq]. */
int uh_knn_scan_shard_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int max_dist,
                          uint64_t* d_cand, int32_t* d_counts, int cap);
/* Query rows [*d_valid_rows, nq) of every following uh_knn_scan_shard_dev are not scanned and emit EMPTY lists (NULL: all rows count).
 * For a caller whose query buffer is a fixed-capacity frame block of which only the first `count` rows — a number that lives on the
 * device — are this frame's descriptors (FrameExtractor's output, frameextractor.cpp:270-340, fed to the sharded matcher): rows left
 * over from an earlier frame must not be able to overflow a list. */
int uh_knn_set_valid_rows_dev(uh_knn* idx, const int32_t* d_valid_rows);
int uh_knn_replay_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                      const uint64_t* d_cand_all, const int32_t* d_counts_all, int nshards, int cap,
                      int32_t* d_indices, int32_t* d_distances);
/* Ranks that hold ONLY their tile of a sharded train set ("each GPU emits local top-k per query with GLOBAL train
 * indices"): the index is built over the tile's rows, uh_knn_set_row_offset gives row 0's global index (used by
 * uh_knn_scan_shard_dev), and uh_knn_replay_tiles_dev replays the gathered lists without ever touching train rows — a list
 * that overflowed its cap sets *d_overflow (device int32, zeroed by the caller) instead of being rescanned. */
/* xflann::Index::toStream / fromStream (index.cpp:153-188) for the hierarchical k-means index: 8-byte signature 12837333433, 8-byte
 * std::hash<std::string>("kmeans"), then KMeansIndex::toStream (kmeansindex.cpp:209-216) = signature 55824124 + 40-byte params + block
 * data.  Byte-identical to what the real library writes for the same features (tests/golden/hkmeans_stream_golden.npz); a stream
 * written by the real library loads and searches identically.  Linear has no stream form in the reference ("Not yet", linear.cpp:78). */
int uh_knn_to_stream(uh_knn* idx, uint8_t* out, uint64_t cap, uint64_t* size);
int uh_knn_from_stream(uh_knn* idx, const uint8_t* data, uint64_t nbytes);
int uh_knn_set_row_offset(uh_knn* idx, int offset);
int uh_knn_replay_tiles_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                            const uint64_t* d_cand_all, const int32_t* d_counts_all, int nshards, int cap,
                            int32_t* d_indices, int32_t* d_distances, int32_t* d_overflow);
/* ... on lists that still sit inside gathered messages: shard s's blocks begin cand_stride (uint64 elements) / count_stride (int32
 * elements) behind shard s-1's — what uh_fstream_finish_dev uses, nothing is unpacked before the replay */
int uh_knn_replay_tiles_strided_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                                    const uint64_t* d_cand_all, size_t cand_stride, const int32_t* d_counts_all, size_t count_stride,
                                    int nshards, int cap, int32_t* d_indices, int32_t* d_distances, int32_t* d_overflow);

/* ------------------------------------------------------------------------
 * ORB extractor — replaces ucoslam::ORBextractor behind Feature2DSerializable:
 *   src/featureextractors/feature2dserializable.h:30-95 (plugin surface, FeatParams :34-60)
 *   src/featureextractors/ORBextractor.cpp:1139 detectAndCompute_impl -> :1247-1353 compute
 * Output is layout-compatible with what the reference hands back:
 *   uh_keypoint == cv::KeyPoint (7 x 4 bytes, ORBextractor.cpp:1297 memcpy), descriptors N x 32 CV_8U.
 * Keypoints come in level order, within a level in the reference's cell-row-major /
 * retainBest-permuted order.
 * ------------------------------------------------------------------------ */
typedef struct uh_orb uh_orb;

typedef struct uh_keypoint {       /* cv::KeyPoint */
    float x, y;                    /* pt, level-0 coordinates */
    float size;                    /* 31 * scale[octave], truncated to int (ORBextractor.cpp:1046) */
    float angle;                   /* degrees [0,360), cv::fastAtan2 of the intensity centroid */
    float response;                /* FAST-9/16 corner score */
    int32_t octave;
    int32_t class_id;              /* -1 */
} uh_keypoint;

typedef struct uh_feat_params {    /* Feature2DSerializable::FeatParams (feature2dserializable.h:34-60) */
    int32_t nthreads;              /* accepted for interface parity; the GPU path ignores it */
    int32_t maxFeatures;           /* default 4000 */
    int32_t nOctaveLevels;         /* default 8 */
    float   scaleFactor;           /* default 1.2 */
    float   sensitivity;           /* default 0 */
} uh_feat_params;

int  uh_orb_create(uh_ctx* ctx, uh_orb** out);
void uh_orb_destroy(uh_orb* orb);
/* detectAndCompute_impl re-runs precalculateParams when params change (:1142-1145); thresholds reset to 20/7 */
int  uh_orb_set_params(uh_orb* orb, const uh_feat_params* params);
int  uh_orb_get_params(const uh_orb* orb, uh_feat_params* params);
/* Feature2DSerializable::toStream / fromStream (feature2dserializable.cpp:76-113) + ORBextractor::toStream_impl / fromStream_impl
 * (ORBextractor.cpp:417-423): u64 signature 1828374733, u64 type tag (F2D_ORB = 0), parameter string (u32 length + bytes), raw FeatParams.
 * to_stream with out == NULL only reports *size.  from_stream fails with the reference's message on a wrong signature and refuses
 * the grid-extractor type tags. */
int  uh_orb_to_stream(const uh_orb* orb, const char* str_params, uint8_t* out, uint64_t cap, uint64_t* size);
int  uh_orb_from_stream(uh_orb* orb, const uint8_t* data, uint64_t nbytes, char* str_params_out, uint64_t str_cap, uint64_t* consumed);
/* FrameExtractor::toStream / fromStream (src/utils/frameextractor.cpp:651-884 / :886-1136, signature 1923123): the extractor block of a
 * reference `.slm` checkpoint — the Feature2DSerializable stream above, the frame counter, three flags, marker size, the FeatParams in
 * use, maxDescDistance, then the aruco::MarkerDetector stream (3rdparty/aruco/aruco/markerdetector.cpp:256-277, behind u64 13213) and the
 * ucoslam::Params stream (src/ucoslamtypes.cpp:63-121).  The two trailing sub-streams belong to subsystems that stay with the host:
 * fromStream walks them and reports where they lie (offsets / lengths into `data`), toStream writes the bytes it is given verbatim or,
 * with NULL, what the reference's default-constructed objects write.  The HIP extractor detects no markers: detect_markers = 1 needs
 * the host's own detector stream on writing and allow_markers on reading, and is refused otherwise. */
typedef struct uh_frame_extractor_state {
    uint32_t counter;               /* frames processed so far */
    uint8_t  remove_from_markers, detect_markers, detect_keypoints;
    float    marker_size;           /* Params::aruco_markerSize */
    uh_feat_params feat_params;     /* the FeatParams the extractor is run with */
    float    max_desc_distance;     /* Params::maxDescDistance */
} uh_frame_extractor_state;
int  uh_frame_extractor_to_stream(const uh_orb* orb, const char* str_params, const uh_frame_extractor_state* state, const uint8_t* aruco_stream,
                                  uint64_t aruco_bytes, const uint8_t* params_stream, uint64_t params_bytes, uint8_t* out, uint64_t cap, uint64_t* size);
int  uh_frame_extractor_from_stream(uh_orb* orb, const uint8_t* data, uint64_t nbytes, int allow_markers, uh_frame_extractor_state* state,
                                    char* str_params_out, uint64_t str_cap, uint64_t* aruco_off, uint64_t* aruco_bytes, uint64_t* params_off,
                                    uint64_t* params_bytes, uint64_t* consumed);
/* ucoslam::Params as a stream carries it (src/ucoslamtypes.cpp:63-121 toStream, :123-180 fromStream: u64 9837138769928 ... u64 1837138769921):
 * the members the hot path and its host glue are driven by, parsed; the ArUco members are walked over.  Errors carry the reference's
 * messages ("Invalid signature", "Reached EOF without finding end signature"). */
typedef struct uh_params_view {
    uint8_t detect_markers, detect_keypoints, kp_non_maxima_suppression, force_initialization_from_markers, remove_keypoints_into_markers,
            run_sequential, auto_adjust_kp_sensitivity, relocalization_with_keypoints, relocalization_with_markers;
    int8_t  kp_descriptor_type;             /* DescriptorTypes::Type (int8): 1 = DESC_ORB */
    float   target_focus, kf_min_confidence, max_desc_distance, baseline_median_depth_ratio_min, aruco_marker_size, kf_culling, th_ref_ratio,
            scale_factor, min_base_line, kpt_image_scale_factor;
    int32_t max_new_points, proj_dist_thr, nthreads_feature_detector, max_visible_frames_per_marker, min_num_proj_points, max_features, n_octave_levels;
    char    global_optimizer[32];           /* "g2o" */
} uh_params_view;
int uh_params_from_stream(const uint8_t* data, uint64_t nbytes, uh_params_view* out, uint64_t* consumed);

/* System::saveToFile / readFromFile (src/utils/system.cpp:8099-8720), the `.slm` checkpoint:
 *   u64 182312 | Map | Params | se3 pose (6 x f32) | i64 current keyframe | bool initialised | STATE i32 | MODES i32 | Frame current |
 *   Frame previous | FrameExtractor | MapManager | cv::Mat (i32 rows, cols, type + data) | i64 | u64
 * The hot path owns Params and the FrameExtractor block; Map / Frame / MapManager are the host's containers
 * and carry no length — only their own readers know where they end.  Reading is therefore sectioned:
 *   uh_system_stream_begin   checks the signature, says where the Map block starts (8)
 *   [host: Map::fromStream]
 *   uh_system_stream_state   on the bytes behind the Map: Params + the five state members; *consumed = where the current Frame starts
 *   [host: Frame::fromStream x 2]
 *   uh_frame_extractor_from_stream
 *   [host: MapManager::fromStream]
 *   uh_system_stream_tail    the cv::Mat header (its data stays in place: offset / length) and the two trailing integers
 * and uh_system_to_stream composes a checkpoint from the host's blocks (opaque bytes; a missing one is refused by name), the extractor
 * block of uh_frame_extractor_to_stream and the values below.  params == NULL writes the Params block the extractor block ends with. */
typedef struct uh_system_state {
    float   cur_pose_rt[6];                 /* se3: rx ry rz tx ty tz (NaN = invalid) */
    int64_t current_keyframe;               /* -1 = none */
    uint8_t is_initialized;
    int32_t state;                          /* STATE_TRACKING = 0, STATE_LOST = 1 */
    int32_t mode;                           /* MODE_SLAM = 0, MODE_LOCALIZATION = 1 */
} uh_system_state;
typedef struct uh_system_tail {
    int32_t  mat_rows, mat_cols, mat_type;  /* cv::Mat header (io_utils.cpp:21-37); rows x cols == 0: no data */
    uint64_t mat_data_offset, mat_data_bytes;   /* (reader) where the matrix data lies behind `data` */
    int64_t  last_value_i64;                /* System's trailing int64 member (-1 by default) */
    uint64_t last_value_u64;                /* System's trailing uint64 member (0 by default) */
} uh_system_tail;
typedef struct uh_system_parts {
    const uint8_t* map;         uint64_t map_bytes;          /* host: Map::toStream */
    const uint8_t* params;      uint64_t params_bytes;       /* Params::toStream bytes, or NULL */
    uh_system_state state;
    const uint8_t* cur_frame;   uint64_t cur_frame_bytes;    /* host: Frame::toStream */
    const uint8_t* prev_frame;  uint64_t prev_frame_bytes;
    const uint8_t* extractor;   uint64_t extractor_bytes;    /* uh_frame_extractor_to_stream */
    const uint8_t* map_manager; uint64_t map_manager_bytes;  /* host: MapManager::toStream */
    uh_system_tail tail;        const uint8_t* mat_data;
} uh_system_parts;
int uh_system_stream_begin(const uint8_t* data, uint64_t nbytes, uint64_t* map_offset);
int uh_system_stream_state(const uint8_t* data, uint64_t nbytes, uh_params_view* params, uh_system_state* state, uint64_t* params_bytes, uint64_t* consumed);
int uh_system_stream_tail(const uint8_t* data, uint64_t nbytes, uh_system_tail* tail, uint64_t* consumed);
int uh_system_to_stream(const uh_system_parts* parts, uint8_t* out, uint64_t cap, uint64_t* size);
int  uh_orb_set_blur(uh_orb* orb, int do_blur);            /* ORBextractor::doGaussianBlur() (ORBextractor.h:112) */
int  uh_orb_set_sensitivity(uh_orb* orb, float v);         /* ORBextractor::setSensitivity (ORBextractor.cpp:457-466) */
int  uh_orb_set_nonmaxima(uh_orb* orb, int on);            /* debug string "orb_nonmaxima" (ORBextractor.cpp:1146-1148,1176-1205): radius-3
                                                              suppression per level before the descriptors; kept keypoints get class_id 1 */
/* Multi-GPU extraction of one frame ("pyramid levels sharded"): this instance extracts pyramid levels
 * [first_level, end_level) only (end_level < 0: to the last level).  Per-level budgets, thresholds and cell grids stay those
 * of the full level count (ORBextractor.cpp:501-513 fixes them per level), the resize chain is built redundantly up to
 * end_level-1 (level l reads level l-1, :1379), so the shards' outputs concatenated in level order are the full extraction's
 * rows, bit for bit.  An empty range yields zero keypoints. */
int  uh_orb_set_level_range(uh_orb* orb, int first_level, int end_level);
int  uh_orb_max_keypoints(const uh_orb* orb);              /* = maxFeatures: upper bound of *n_out */

/* One frame, host buffers: img = h rows of w bytes (CV_8UC1), row stride in bytes.
 * kps/desc must hold `cap` entries; *n_out receives the keypoint count (UH_ECAPACITY if n > cap).
 * An empty image (NULL / w<=0 / h<=0) returns 0 keypoints silently (ORBextractor.cpp:1254). */
int uh_orb_extract(uh_orb* orb, const uint8_t* img, int w, int h, size_t stride,
                   uh_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* The frame as the camera delivers it and the keypoints as Frame stores them — the two per-frame steps FrameExtractor performs around the
 * extractor (src/utils/frameextractor.cpp:2960,3046 cv::cvtColor(COLOR_BGR2GRAY) for three-channel input; :3985 undistortPoints(kpts,
 * ImageParams) = src/basictypes/misc.cpp:269-293: cv::undistortPoints, then x*fx + cx in float) done on the device inside the same launches:
 *   channels  1 (CV_8UC1), 3 (BGR) or 4 (BGRA) interleaved bytes per pixel, row stride in bytes
 *   und_xy    NULL, or cap x 2 floats receiving Frame::und_kpts[i].pt for every returned keypoint (needs uh_orb_set_camera)
 * uh_orb_set_camera: ImageParams' CameraMatrix (CV_32F) and Distorsion (k1 k2 p1 p2 [k3 [k4 k5 k6]], n_dist of them; 0 = none: the result is
 * still the float round trip (x - cx)/fx*fx + cx of the reference, not the input).  NULL switches the camera off.
 * uh_undistort_points_host: the same arithmetic on the host (no GPU) for n points. */
typedef struct uh_camera { float fx, fy, cx, cy; float dist[8]; int32_t n_dist; } uh_camera;
int uh_orb_set_camera(uh_orb* orb, const uh_camera* cam);
int uh_orb_extract_frame(uh_orb* orb, const uint8_t* img, int w, int h, size_t stride, int channels,
                         uh_keypoint* kps, uint8_t* desc, float* und_xy, int cap, int* n_out);
int uh_undistort_points_host(const uh_camera* cam, const float* xy, int n, float* out_xy);
/* The Frame that FrameExtractor::process produces, kept ON THE DEVICE for the tracker (src/utils/frameextractor.cpp:430-520 detectAndCompute,
 * :3985 undistortPoints, :4258 Frame::create_kdtree -> src/basictypes/picoflann.h:150-163,238-345): uh_orb_extract_frame_dev is
 * uh_orb_extract_frame (same host outputs, same completion) that also leaves the descriptors and the undistorted keypoints in `frame`
 * (HBM) and enqueues ONE more launch behind its completion word which builds picoflann's kd-tree there, node for node what the reference
 * builds on the CPU (up to 4096 keypoints; needs uh_orb_set_camera).  uh_projmatch_set_frame_dev adopts such a frame: nothing goes
 * device -> host -> device, no tree is built on the host.  A frame object is overwritten by the next extraction into it: keep two to
 * hold the previous frame.  uh_dev_frame_tree (inspection / tests) waits for the build and copies the tree out: nodes24_out as in
 * uh_projmatch_debug_tree with room for 2n/5 + 2 nodes, leaf_idx_out / leaf_octave_out n entries, leaf_xy_out 2n floats (any may be NULL). */
typedef struct uh_dev_frame uh_dev_frame;
int  uh_dev_frame_create(uh_ctx* ctx, uh_dev_frame** out);
void uh_dev_frame_destroy(uh_dev_frame* frame);
/* Who builds the kd-tree of a device frame.  on_host = 0 (default): the build launches behind the extraction (the host does nothing but wait).
 * on_host = 1: no build launch; uh_projmatch_set_frame_dev builds the tree on the calling core from the host copy of the undistorted keypoints
 * (frame->und_kpts / n_kpts of its uh_proj_frame argument, exactly the arrays uh_orb_extract_frame_dev returned) and uploads nodes and leaf records
 * into the frame object — the descriptors still never cross the host link a second time.  Same tree, byte for byte; the faster route while a
 * core is free (Frame::create_kdtree, frameextractor.cpp:4258). */
int  uh_dev_frame_set_tree_builder(uh_dev_frame* frame, int32_t on_host);
/* a frame that was not extracted here (read from a file, made by another extractor) into a device frame object: n undistorted keypoints (x, y, octave
 * are used) and their n x 32 descriptor bytes; the tree is built as for an extracted frame.  Synchronous — a utility, not the per-frame path. */
int  uh_dev_frame_upload(uh_dev_frame* frame, const uh_keypoint* und_kpts, int32_t n, const uint8_t* desc);
int  uh_orb_extract_frame_dev(uh_orb* orb, const uint8_t* img, int w, int h, size_t stride, int channels,
                              uh_keypoint* kps, uint8_t* desc, float* und_xy, int cap, int* n_out, uh_dev_frame* frame);
/* the same call in two halves: _begin returns as soon as the undistorted keypoints — x, y, octave; the other fields come with _end — are on the host
 * (*und_kpts_early: this object's array, valid until the next extraction), so that the caller builds the kd-tree (uh_projmatch_set_frame_dev with
 * uh_dev_frame_set_tree_builder(frame, 1)) while the descriptors are still being computed; _end completes kps / desc / und_xy.  No other
 * extraction on this object in between. */
int  uh_orb_extract_frame_dev_begin(uh_orb* orb, const uint8_t* img, int w, int h, size_t stride, int channels,
                                    uh_keypoint* kps, uint8_t* desc, float* und_xy, int cap, int* n_out, uh_dev_frame* frame, const uh_keypoint** und_kpts_early);
int  uh_orb_extract_frame_dev_end(uh_orb* orb, int* n_out);
int  uh_dev_frame_tree(uh_dev_frame* frame, int32_t* n_kpts, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out, float* leaf_xy_out,
                       int32_t* leaf_octave_out, double* root_box4, int32_t* max_depth);
/* `batch` frames resident in HBM (frame f at d_imgs + f*frame_stride); outputs per frame at
 * d_kps + f*cap_per_frame, d_desc + f*cap_per_frame*32, d_counts[f].  Asynchronous on the context stream. */
int uh_orb_extract_dev(uh_orb* orb, const uint8_t* d_imgs, int w, int h, size_t stride, size_t frame_stride,
                       int batch, uh_keypoint* d_kps, uint8_t* d_desc, int cap_per_frame, int32_t* d_counts);
/* The same for `batch` frames in HOST memory, results to host memory (fixed-capacity blocks: frame f's rows at kps + f*cap, desc + f*cap*32,
 * counts[f] of them valid; cap >= maxFeatures): one H2D copy, the batched launch set, three D2H copies, one synchronisation. */
int uh_orb_extract_batch(uh_orb* orb, const uint8_t* imgs, int w, int h, size_t stride, size_t frame_stride, int batch,
                         uh_keypoint* kps, uint8_t* desc, int cap, int32_t* counts);
/* Verification tap: copy level `level` of frame `frame` (which: 0 pyramid, 1 FAST strength map) to `out`
 * (w*h bytes, may be NULL to query the size). */
int uh_orb_debug_level(uh_orb* orb, int frame, int level, int which, uint8_t* out, int* w_out, int* h_out);

/* ------------------------------------------------------------------------
 * Bundle adjustment — replaces GlobalOptimizerG2O behind the GlobalOptimizer plugin:
 *   src/optimization/globaloptimizer.h:28-68       setParams / optimize(bool* stopASAP) / getResults
 *   src/optimization/globaloptimizer_g2o.cpp:77-401 (graph build), :418-464 (two-pass LM), :466-537 (results)
 *   3rdparty/g2o: Levenberg + BlockSolver_6_3 (Schur) + LinearSolverEigen (LDLT) + Huber
 * The caller flattens the map exactly as setParams does (INTEGRATION.md shows the adaptor):
 *   frames  : pose_f2g as the reference stores it (row-major 4x4 float, cv::Mat CV_32F), fixed flag
 *             (FIXED_WITHPOINTS / FIXED_WITHOUTPOINTS both = 1), intrinsics fx fy cx cy
 *   points  : float xyz (MapPoint::getCoordinates)
 *   obs     : one monocular EdgeSE3ProjectXYZ per (point, frame): undistorted keypoint (float x,y) and
 *             information scalar 1/scaleFactor[octave] (globaloptimizer_g2o.cpp:96-97,244)
 * Arithmetic is fp64 like g2o; poses agree with the reference within 1e-6 (se3 state), see DESIGN.md.
 * Monocular edges only.  Up to 64 non-fixed frames run the local-BA form (two launches per LM trial, reduced system in one
 * workgroup's LDS); more (global BA, up to 4096) run the wide form: sparse camera-pair lists, blocked dense LDL^T in HBM.
 * ------------------------------------------------------------------------ */
typedef struct uh_ba uh_ba;

typedef struct uh_ba_problem {
    int32_t n_frames, n_points, n_obs;
    const float*   poses_f2g;       /* n_frames x 16 */
    const uint8_t* fixed;           /* n_frames */
    const float*   intr;            /* n_frames x 4: fx fy cx cy */
    const float*   points;          /* n_points x 3 */
    const int32_t* obs_point;       /* n_obs */
    const int32_t* obs_frame;       /* n_obs */
    const float*   obs_uv;          /* n_obs x 2 */
    const double*  obs_inv_sigma;   /* n_obs: (double)_InvScaleFactors[octave], i.e. (double)(float)(1. / scaleFactor[octave]) — the reference stores
                                     * the inverse scale factors as float (globaloptimizer_g2o.h:76), g2o sees that value widened */
} uh_ba_problem;

typedef struct uh_ba_params {
    int32_t n_iters;                /* ParamSet::nIters: pass 1 runs n_iters, pass 2 runs 2*n_iters (local BA: 5) */
    double  huber_delta;            /* <= 0 -> sqrt(5.99) */
    double  chi2_threshold;         /* <= 0 -> 5.99 */
    float   min_chi2_between_iter;  /* SparseOptimizer::optimize(iters, minChi2BetweenIter): 1 from BA */
} uh_ba_params;

int  uh_ba_create(uh_ctx* ctx, uh_ba** out);
void uh_ba_destroy(uh_ba* ba);
/* = setParams: snapshots the problem into HBM; the caller's arrays may change afterwards. params may be NULL. */
int  uh_ba_set_problem(uh_ba* ba, const uh_ba_problem* problem, const uh_ba_params* params);
/* = optimize(bool* stopASAP): runs both passes from the snapshot; *stop_asap (may be NULL) is polled while waiting. */
int  uh_ba_optimize(uh_ba* ba, const volatile uint8_t* stop_asap);
/* pinned device-visible force-stop byte owned by the optimiser (write 1 from any thread to stop between trials) */
/* asynchronous form: runs uh_ba_optimize on a worker thread owned by the object — the reference runs GlobalOptimizer::optimize
 * on its mapper thread beside the tracker (mapmanager.cpp:150) — and uh_ba_wait returns its result (UH_OK or the error).
 * One optimisation in flight per object; set_problem / get_results only between wait and the next optimize.
 * Both sides of the hand-over spin before they sleep (the worker for the next request: <= 0.3 ms; uh_ba_wait for the result: <= 2 ms),
 * because an optimisation is half a millisecond and a futex wake-up is 10-20 us of it: a caller that wants its core back while a
 * long (global) optimisation runs should do other work before calling uh_ba_wait, not rely on uh_ba_wait to block at once. */
int  uh_ba_optimize_async(uh_ba* ba, const volatile uint8_t* stop_asap);
int  uh_ba_wait(uh_ba* ba);
uint8_t* uh_ba_stop_flag(uh_ba* ba);
/* = getResults: poses n_frames x 16 float (fixed frames returned unchanged), points n_points x 3 float, per-observation
 * chi2 (as last evaluated by the optimiser) and bad-association flag (chi2 > 5.99 or point behind the camera);
 * iters_out[2] = outer iterations executed in pass 1 / pass 2.  Any output pointer may be NULL. */
int  uh_ba_get_results(uh_ba* ba, float* poses_out, float* points_out, double* chi2_out, uint8_t* bad_out, int32_t* iters_out);
int  uh_ba_get_pose_state(uh_ba* ba, double* pose7_out);   /* n_frames x (qx qy qz qw tx ty tz), fp64 */

/* ---- the same three phases without a host copy on either side (what the C++ adaptor's flatten_for_ba / getResults use).
 * A local BA is a NEW problem per keyframe (mapmanager.cpp:11388-11405 setParams + optimize on the mapper thread, :1267-1305
 * getResults on the tracker thread), so setParams and getResults are on the clock as much as optimize:
 *   uh_ba_map_staging   the optimiser owns ONE pinned, device-visible staging block; the caller writes the flattened problem straight
 *                       into it (observations as 24-byte records) — capacities are rounded up and kept across problems
 *   uh_ba_set_problem_staged   = setParams on what the staging block holds: one H2D copy + one kernel that scatters the observation
 *                       indices into a (point x frame) table in HBM; nothing is built on the host, nothing is synchronised.
 *                       An observation whose point / frame index is out of range is refused here; a (point, frame) pair that
 *                       occurs twice is detected by that kernel and reported by the following uh_ba_optimize (UH_EINVAL).
 *   uh_ba_results_view  = getResults in place: pointers into the pinned result block the optimisation kernel wrote (valid until the
 *                       next set_problem on this object); uh_ba_get_results copies from the same block.
 * uh_ba_set_problem (arrays anywhere in host memory) converts into the staging block and takes the same path. */
typedef struct uh_ba_obs {          /* one monocular EdgeSE3ProjectXYZ (globaloptimizer_g2o.cpp:224-249) */
    int32_t point, frame;           /* indices into points / frames of this problem */
    float   u, v;                   /* undistorted keypoint */
    double  inv_sigma;              /* (double)_InvScaleFactors[octave] */
} uh_ba_obs;                        /* 24 bytes */

typedef struct uh_ba_staging {
    float*     poses_f2g;           /* cap_frames x 16 */
    uint8_t*   fixed;               /* cap_frames */
    float*     intr;                /* cap_frames x 4 */
    float*     points;              /* cap_points x 3 */
    uh_ba_obs* obs;                 /* cap_obs */
    int32_t    cap_frames, cap_points, cap_obs;
} uh_ba_staging;

typedef struct uh_ba_results_view {
    const float*   poses;           /* n_frames x 16 (fixed frames unchanged) */
    const float*   points;          /* n_points x 3 */
    const double*  chi2;            /* n_obs */
    const uint8_t* bad;             /* n_obs */
    const double*  pose_state;      /* n_frames x 7 (qx qy qz qw tx ty tz) */
    int32_t        iters[2];
    int32_t        n_frames, n_points, n_obs;
} uh_ba_results_view;

int  uh_ba_map_staging(uh_ba* ba, int n_frames, int n_points, int max_obs, uh_ba_staging* out);
int  uh_ba_set_problem_staged(uh_ba* ba, int n_frames, int n_points, int n_obs, const uh_ba_params* params);
int  uh_ba_results_view_get(uh_ba* ba, uh_ba_results_view* out);
/* setParams + optimize on the object's worker thread (the reference's mapper thread does exactly these two calls back to back);
 * problem == NULL: the staging block (n_frames / n_points / n_obs as given).  The caller's arrays must stay valid until uh_ba_wait. */
int  uh_ba_solve_async(uh_ba* ba, const uh_ba_problem* problem, int n_frames, int n_points, int n_obs, const uh_ba_params* params,
                       const volatile uint8_t* stop_asap);
/* which form the current problem runs in: 0 launch chain (9+ free keyframes that do not fit the persistent form), 1 persistent
 * one-launch kernel, 2 wide (global BA); *lanes_per_landmark_out (may be NULL) = the persistent form's padded number of free cameras */
int  uh_ba_form(uh_ba* ba, int* lanes_per_landmark_out);
/* the per-observation chi2 is an extra of this ABI (GlobalOptimizer::getResults does not return it) and three quarters of the bytes the
 * optimisation kernel hands over: uh_ba_want_chi2(ba, 0) leaves it out from the next set_problem on (chi2_out must then be NULL) */
int  uh_ba_want_chi2(uh_ba* ba, int on);

/* ------------------------------------------------------------------------
 * Bag of words — replaces fbow::Vocabulary::transform / fBow::score:
 *   3rdparty/fbow/fbow/fbow.h:54-116 (class surface), fbow.cpp:51-90 (transform with level), :92-143 (normalised
 *   transform), :171-190 (stream format), :192-243 (score); UcoSLAM call site keyframedatabase.cpp:310-322 (level 3).
 * The GPU does the per-descriptor tree descent; the std::map containers of the reference API are assembled by the
 * host adaptor from the per-descriptor triples, in feature order (keeps the reference's float summation order).
 * ------------------------------------------------------------------------ */
typedef struct uh_bow uh_bow;
int  uh_bow_create(uh_ctx* ctx, uh_bow** out);
void uh_bow_destroy(uh_bow* bow);
/* Vocabulary::fromStream: `stream` = u64 signature 55824124 + raw params struct (120 bytes) + block blob */
int  uh_bow_load(uh_bow* bow, const void* stream, size_t nbytes);
/* same, from an already split params struct (fbow::Vocabulary::params, 120 bytes) and blob */
int  uh_bow_set(uh_bow* bow, const void* params120, const void* blob);
int  uh_bow_get_params(const uh_bow* bow, void* params120);
/* per descriptor i: word[i] (0xFFFFFFFF if the descent ends without a leaf), weight[i], node[i] = node id at `level`
 * (valid[i] = 0 if never recorded).  Empty input fails like the reference ("No input data"). */
int  uh_bow_transform(uh_bow* bow, const uint8_t* desc, int n, size_t stride, int desc_bytes, int level,
                      uint32_t* word, float* weight, uint32_t* node, uint8_t* valid);
int  uh_bow_transform_dev(uh_bow* bow, const uint8_t* d_desc, int n, int level,
                          uint32_t* d_word, float* d_weight, uint32_t* d_node, uint8_t* d_valid);
double uh_bow_score(const uint32_t* ids1, const float* w1, int n1, const uint32_t* ids2, const float* w2, int n2);

/* KPFrameDataBase (src/map_types/keyframedatabase.cpp:150-238): the keyframes' bags of words resident in HBM.  add / del mirror
 * KPFrameDataBase::add / del (the bag is the fBow of uh_bow_transform(desc, 3): words ascending, raw weights).  query is the
 * first half of relocalizationCandidates (:195-238): per keyframe the number of query words it shares and fBow::score (one lane
 * per keyframe, products added in ascending word order like the reference), then on the host maxCommonWords, the 0.8 cut and
 * the min_score cut; it returns the surviving frames (`frame_score`) in ascending id order.  The covisibility accumulation that
 * follows (:241-275) needs CovisGraph and stays with the caller (ucoslam_cv3_amd.bow.KPFrameDataBase shows it). */
typedef struct uh_bowdb uh_bowdb;
int  uh_bowdb_create(uh_ctx* ctx, uh_bowdb** out);
void uh_bowdb_destroy(uh_bowdb* db);
int  uh_bowdb_size(const uh_bowdb* db);
int  uh_bowdb_add(uh_bowdb* db, uint32_t frame_id, const uint32_t* words, const float* weights, int n);
int  uh_bowdb_del(uh_bowdb* db, uint32_t frame_id);
int  uh_bowdb_query(uh_bowdb* db, const uint32_t* words, const float* weights, int n, const uint32_t* excluded, int n_excluded,
                    float min_score, uint32_t* frame_ids_out, uint32_t* nobs_out, double* score_out, int cap);

/* ------------------------------------------------------------------------
 * FrameMatcher_Flann post-filter (host policy around the index) — src/utils/framematcher.cpp:228-319:
 * per query best/second-best scan over the nn columns IN ROW ORDER (min distance gate, octave gap, optional epipolar
 * gate 3.84*sigma^2, ratio test against a same-octave runner-up), filter_ambiguous_train (misc.cpp:153-185), then the
 * 30-bin orientation histogram keeping the three dominant bins (:290-316, computeThreeMaxima :67-108).
 * Returns the number of matches written (>= 0) or a negative UH_E* code.
 * ------------------------------------------------------------------------ */
typedef struct uh_dmatch {          /* cv::DMatch */
    int32_t queryIdx, trainIdx, imgIdx;
    float   distance;
} uh_dmatch;

typedef struct uh_match_filter_args {
    int32_t nq, nn;
    const int32_t* indices;         /* nq x nn rows of uh_knn_search (unsorted heap order, as FrameMatcher passes sorted=false) */
    const int32_t* distances;
    const uint32_t* map_idx_query;  /* row -> keypoint index of the query frame (manageMode), NULL = identity */
    const uint32_t* map_idx_train;  /* index row -> keypoint index of the train frame, NULL = identity */
    const int32_t* q_octave; const float* q_angle; const float* q_pt;   /* query und_kpts fields; q_pt (x,y) only for F12 */
    const int32_t* t_octave; const float* t_angle; const float* t_pt;   /* train und_kpts fields */
    const float* scale_factors;     /* query frame scaleFactors (only with F12) */
    const float* F12;               /* row-major 3x3 fundamental matrix or NULL (no epipolar gate) */
    float min_desc_dist, nn_match_ratio;
    int32_t check_orientation, max_octave_diff;
    /* sizes of the arrays above; > 0: every mapped keypoint index / octave is range-checked (UH_EINVAL), 0: unchecked */
    int32_t n_query_kpts, n_train_kpts, n_levels;
} uh_match_filter_args;

int uh_match_filter(const uh_match_filter_args* args, uh_dmatch* out, int cap);
/* filter_ambiguous_train (by_train != 0) / filter_ambiguous_query (by_train == 0), in place; returns the new count */
int uh_filter_ambiguous(uh_dmatch* matches, int n, int by_train);

/* FrameMatcher_BoW::match / matchEpipolar (framematcher.cpp:395-535): features of the two frames that share a vocabulary node
 * (Frame::bowvector_level, the fbow::fBow2 of uh_bow_transform at level 3) are compared all against all.  A frame comes as the
 * node map in std::map order — ascending node ids, per node the feature indices in push_back order — plus the keypoint arrays
 * and `used` = isUsed(frame, idx, mode) (:373-379: not FLAG_NONMAXIMA, and assigned / unassigned as the mode asks; NULL = all).
 * Hamming distances and the order-dependent best / "last not better" bookkeeping (:441-476) run on the GPU, the acceptance
 * rule, filter_ambiguous_train and the orientation histogram on the host.  Matches: queryIdx / trainIdx = keypoint indices.
 * A node with an empty list is refused (the reference's loop would never advance past it, :436). */
typedef struct uh_bow_frame {
    int32_t n_nodes;
    const uint32_t* node_ids;       /* n_nodes, ascending */
    const int32_t* node_ptr;        /* n_nodes + 1 offsets into feat_idx */
    const uint32_t* feat_idx;       /* keypoint indices, node lists back to back */
    int32_t n_kpts;
    const uint8_t* desc;            /* n_kpts x 32 */
    const int32_t* octave; const float* angle;   /* und_kpts fields */
    const float* pt;                /* n_kpts x 2 (x, y); only read with F12 */
    const uint8_t* used;            /* n_kpts or NULL */
} uh_bow_frame;
typedef struct uh_bow_match_args {
    uh_bow_frame query, train;
    const float* scale_factors; int32_t n_levels;   /* query frame scaleFactors (only with F12) */
    const float* F12;               /* row-major 3x3 fundamental matrix (getFund12) or NULL */
    float min_desc_dist, nn_match_ratio;
    int32_t check_orientation, max_octave_diff;
} uh_bow_match_args;
typedef struct uh_bowmatch uh_bowmatch;
int  uh_bowmatch_create(uh_ctx* ctx, uh_bowmatch** out);
void uh_bowmatch_destroy(uh_bowmatch* bm);
int  uh_bowmatch_match(uh_bowmatch* bm, const uh_bow_match_args* args, uh_dmatch* out, int cap);

/* ------------------------------------------------------------------------
 * One frame stream sharded over the GPUs of a node (BASELINE config 5): pyramid levels of frame t, the map's train tiles
 * and the fbow descent of frame t-1 sharded over `world` ranks (one process per GPU), ONE all-gather of fixed-size messages per frame.
 * The producers write straight into the send buffer and the exact replay reads the gathered lists where they lie (csrc/fstream.hip);
 * counts stay on the device, a step never synchronises with the host.  Results trail the input by one frame (software pipeline).
 *   ext   an extractor with its FeatParams set (uh_orb_set_params); its level range is set per call
 *   tile  a uh_knn built over THIS rank's rows [nt*rank/world, nt*(rank+1)/world) with uh_knn_set_row_offset(first global row)
 *   voc   optional vocabulary (NULL: no bag-of-words slices in the message)
 * The collective is RCCL's all-gather on the context's stream: uh_fstream_comm_unique_id on rank 0 -> the 128 bytes reach every rank by
 * any host channel -> uh_fstream_comm_init everywhere (librccl is loaded with dlopen); or uh_fstream_set_comm with the host's own
 * ncclComm_t; world == 1 needs neither.  uh_fstream_put_message serves hosts with another transport.
 * ------------------------------------------------------------------------ */
typedef struct uh_fstream uh_fstream;
typedef struct uh_fstream_params {
    int32_t rank, world;
    int32_t nn;                     /* neighbours per query (FrameMatcher: 10) */
    int32_t max_features;           /* = FeatParams::maxFeatures: rows per frame, queries per search */
    int32_t cand_cap;               /* accept-list capacity per (query, tile); an overflow is reported, see d_overflow */
    int32_t sorted;                 /* KnnSearchParams::sorted */
    int32_t bow_level;              /* Vocabulary::transform level (keyframedatabase.cpp:319: 3) */
} uh_fstream_params;
int    uh_fstream_create(uh_ctx* ctx, uh_orb* ext, uh_knn* tile, uh_bow* voc, const uh_fstream_params* params, uh_fstream** out);
void   uh_fstream_destroy(uh_fstream* fs);
size_t uh_fstream_message_bytes(const uh_fstream* fs);
void*  uh_fstream_send_buffer(uh_fstream* fs);      /* device, message_bytes */
void*  uh_fstream_recv_buffer(uh_fstream* fs);      /* device, world x message_bytes, rank order */
int    uh_fstream_comm_unique_id(uint8_t id_out[128]);
int    uh_fstream_comm_init(uh_fstream* fs, const uint8_t id[128]);
int    uh_fstream_set_comm(uh_fstream* fs, void* nccl_comm);
int    uh_fstream_comm_ranks(uh_fstream* fs);       /* ncclCommCount of the communicator in use (1 without one); < 0 on error */
int    uh_fstream_put_message(uh_fstream* fs, int rank, const void* d_message);
/* step t: this rank's levels [level_first, level_end) of the frame (device image), its tile's accept lists and its fbow slice for frame t-1 */
int    uh_fstream_local_dev(uh_fstream* fs, const uint8_t* d_frame, int w, int h, size_t stride, int level_first, int level_end);
int    uh_fstream_exchange(uh_fstream* fs);
/* frame t complete (d_kps / d_desc: max_features rows, *d_count valid) and frame t-1's rows (d_prev_*: max_features x nn, the first
 * *d_prev_count valid; 0 on the first step) + its fbow triples (max_features each, or NULL).  *d_overflow: bit 0 an accept list exceeded
 * cand_cap (retry the frame with a larger capacity), bit 1 a message header was out of range.  All pointers are device pointers. */
int    uh_fstream_finish_dev(uh_fstream* fs, uh_keypoint* d_kps, uint8_t* d_desc, int32_t* d_count, int32_t* d_prev_indices,
                             int32_t* d_prev_distances, int32_t* d_prev_count, uint32_t* d_bow_word, float* d_bow_weight, uint32_t* d_bow_node,
                             uint8_t* d_bow_valid, int32_t* d_overflow);

/* ------------------------------------------------------------------------
 * Pose-only optimisation — replaces PnPSolver::solvePnp (monocular matches, no markers):
 *   src/optimization/pnpsolver.h:30-38, pnpsolver.cpp:116-409; edge type typesg2o.h:590-650; kernel :82-105.
 * Inputs per match i: map point p3d[i] (float xyz, MapPoint::getCoordinates), undistorted keypoint kp[i] (float x,y),
 * inv_sigma[i] = 1/scaleFactors[octave] (float, :230), weight[i] = 1 or 0.5 for unstable points (:215-216).
 * Outputs: pose (row-major 4x4 float), bad[i] (1 = outlier: the reference marks DMatch::imgIdx = -1), outer iterations of the
 * four rounds.  Returns the inlier count (>= 0, solvePnp's return value) or a negative UH_E* code.
 * ------------------------------------------------------------------------ */
typedef struct uh_pnp uh_pnp;
int  uh_pnp_create(uh_ctx* ctx, uh_pnp** out);
void uh_pnp_destroy(uh_pnp* pnp);
int  uh_pnp_solve(uh_pnp* pnp, const float* pose_f2g, const float* intr4, int n, const float* p3d, const float* kp,
                  const float* inv_sigma, const float* weight, float* pose_out, uint8_t* bad_out, int32_t* iters_out4,
                  double* state_out7);
/* device-resident form: d_work = n*32 bytes of scratch (16-byte aligned; only touched beyond 3000 matches); d_result5 = {inliers, iters[4]};
 * one launch, asynchronous on the context stream */
int  uh_pnp_solve_dev(uh_pnp* pnp, const float* d_pose_f2g, const float* d_intr4, int n, const float* d_p3d, const float* d_kp,
                      const float* d_inv_sigma, const float* d_weight, void* d_work, float* d_pose_out, uint8_t* d_bad_out,
                      int32_t* d_result5, double* d_state7);

/* ------------------------------------------------------------------------
 * Projection matcher — replaces Map::matchFrameToMapPoints (src/map.cpp:651-770) on flattened inputs:
 *   frame side  Frame::und_kpts / desc / scaleFactors / imageParams.CameraMatrix / minXY,maxXY (map_types/frame.h:60-81) and the
 *               kd-tree over und_kpts (Frame::create_kdtree, frame.h:124; picoflann.h) — rebuilt bit-identically by set_frame
 *   map side    the candidate map points AFTER the reference's id filtering (getMapPointsInFrames / lastFIdxSeen, map.cpp:657-668,
 *               which stays host code): ids, getCoordinates(), normal, get{Min,Max}DistanceInvariance(), descriptor
 * match() returns the DMatch list (queryIdx = keypoint, trainIdx = map point id, imgIdx = -1, distance = Hamming as float)
 * after filter_ambiguous_query, in map-point order like the reference; optional per-point outputs: best keypoint before the
 * ambiguity filter (-1 = none), its distance, and the "visible" flags (markMapPointsAsVisible: the caller applies
 * MapPoint::setVisible() to the flagged points).  Returns the number of matches (>= 0) or a negative UH_E* code.
 * ------------------------------------------------------------------------ */
typedef struct uh_proj_frame {
    const uh_keypoint* und_kpts;    /* cv::KeyPoint: pt and octave are read */
    int32_t n_kpts;
    const uint8_t* desc;            /* n_kpts x 32 */
    const float* scale_factors;     /* Frame::scaleFactors */
    int32_t n_levels;
    float fx, fy, cx, cy;           /* CameraMatrix(0,0) (1,1) (0,2) (1,2) */
    int32_t min_x, min_y, max_x, max_y;   /* Frame::minXY / maxXY (cv::Point, i.e. ints; defaults 0,0 / INT_MAX,INT_MAX) */
} uh_proj_frame;

typedef struct uh_map_points {
    int32_t n;
    const uint32_t* ids;            /* n */
    const float* pos3d;             /* n x 3 */
    const float* normal;            /* n x 3 */
    const float* min_dist;          /* n */
    const float* max_dist;          /* n */
    const uint8_t* desc;            /* n x 32 */
} uh_map_points;

typedef struct uh_projmatch uh_projmatch;
int  uh_projmatch_create(uh_ctx* ctx, uh_projmatch** out);
void uh_projmatch_destroy(uh_projmatch* pm);
int  uh_projmatch_set_frame(uh_projmatch* pm, const uh_proj_frame* frame);
/* the frame uh_orb_extract_frame_dev left on the device; `params` gives scale factors, camera and image bounds (its und_kpts / n_kpts / desc
 * are ignored).  `frame` must stay alive and un-overwritten while the matcher uses it. */
int  uh_projmatch_set_frame_dev(uh_projmatch* pm, uh_dev_frame* frame, const uh_proj_frame* params);
int  uh_projmatch_match(uh_projmatch* pm, const float* pose_f2g /* row-major 4x4 */, const uh_map_points* points,
                        float min_desc_dist, float max_repj_dist, uh_dmatch* matches_out, int32_t cap,
                        int32_t* best_kp_out /* n or NULL */, float* best_dist_out /* n or NULL */, uint8_t* visible_out /* n or NULL */);
/* The tracker's projection search against the PREVIOUS frame (src/utils/system.cpp:5930-6460, private member of System; the
 * source is token-pasted, line numbers are statement starts after preprocessing; called at :6559-6565 with
 * (maxDescDistance*1.5, projDistThr) before PnPSolver::solvePnp).  The caller flattens the loop header (:5969-6089): one item
 * per keypoint i of the previous frame whose ids[i] names a valid, non-bad map point, in keypoint order — that point's id and
 * coordinates, und_kpts[i].octave and descriptor row i.  Per item: Frame::project(p, true, true) (frame.h:140-161) with
 * `pose_f2g` = the current frame's pose, radius maxRepjDist * scaleFactors[octave], candidates of exactly that octave in
 * kd-tree order, best (from minDescDist + 0.01) / second WITHOUT demotion, accepted iff best < 0.7 * second; then
 * filter_ambiguous_query.  The frame is the one given to uh_projmatch_set_frame.  Returns the number of matches or < 0. */
typedef struct uh_prev_points {
    int32_t n;
    const uint32_t* ids;            /* n: map point ids (DMatch::trainIdx) */
    const float* pos3d;             /* n x 3: MapPoint::getCoordinates() */
    const int32_t* octave;          /* n: prev.und_kpts[i].octave, each in [0, n_levels) */
    const uint8_t* desc;            /* n x 32: prev.desc rows */
} uh_prev_points;
int  uh_projmatch_match_prev(uh_projmatch* pm, const float* pose_f2g /* row-major 4x4 */, const uh_prev_points* points,
                             float min_desc_dist, float max_repj_dist, uh_dmatch* matches_out, int32_t cap,
                             int32_t* best_kp_out /* n or NULL */, float* best_dist_out /* n or NULL */);
/* The tracker's pose estimation for one frame as ONE call: the search against the previous frame, PnPSolver::solvePnp, the decision
 * (>= min_inliers inliers: refined pose + small disc, else predicted pose + wide radius), Map::matchFrameToMapPoints, the union of the first
 * solve's inliers with the new matches under filter_ambiguous_query, the per-match look-ups and the second solvePnp (src/utils/system.cpp
 * :5930-6460, :6559-6566, :6762-6881, :6897-6954 after preprocessing; pnpsolver.cpp:116-409) — exactly what
 *     uh_projmatch_match_prev -> look-ups -> uh_pnp_solve -> uh_projmatch_match -> union, uh_filter_ambiguous, look-ups -> uh_pnp_solve
 * return, bit for bit, with the look-ups and list handling done on the device between the launches and ONE wait at the end (seven launches on
 * the context stream; csrc/track.hpp).  The look-ups: first solve — the candidate's own position, weight 1; second solve — the position and
 * weight of row prev_map_row[i] of `map` when that is >= 0 (a previous-frame item that is also in the local map), else the candidate's own;
 * keypoint = the frame's undistorted keypoint, inv_sigma = inv_sigma_levels[its octave].  Needs a device-resident frame
 * (uh_orb_extract_frame_dev + uh_projmatch_set_frame_dev, either tree builder) of <= 4096 keypoints; pm and pnp on the same context. */
typedef struct uh_track_args {
    const float* pose0;                 /* predicted pose f2g, row-major 4x4 */
    const float* intr4;                 /* fx fy cx cy */
    const float* inv_sigma_levels;      /* n_levels: 1 / scaleFactor per octave (the solver's per-match inv_sigma) */
    int32_t n_levels;
    const uh_prev_points* prev;         /* candidates of the previous-frame search */
    const int32_t* prev_map_row;        /* prev->n: row of the same map point in `map`, or -1; NULL = all -1 */
    const uh_map_points* map;           /* candidates of the local-map search */
    const float* map_weight;            /* map->n: the solver weight of each map point (0.5 for unstable ones); NULL = all 1 */
    float prev_min_desc_dist, prev_max_repj_dist;   /* system.cpp:6559-6565: maxDescDistance * 1.5, projDistThr */
    float map_min_desc_dist, map_radius_tracked, map_radius_lost;   /* :6762-6881: maxDescDistance * 2, 4 px, projDistThr */
    int32_t min_inliers;                /* 30 */
} uh_track_args;
typedef struct uh_track_result {
    uh_dmatch* matches_prev; uint8_t* bad_prev; int32_t cap_prev;    /* in: buffers (>= prev->n); out: the first search's matches and the first solve's outlier flags */
    uh_dmatch* matches_map; int32_t cap_map;                          /* >= map->n: the second search's own matches */
    uh_dmatch* matches_all; uint8_t* bad_all; int32_t cap_all;        /* >= prev->n + map->n: the union the second solve ran on, its outlier flags */
    int32_t n_prev, n_map, n_all, tracked, inliers1, inliers2;
    int32_t iters1[4], iters2[4];
    float pose1[16], pose2[16];
} uh_track_result;
int  uh_track_pose(uh_projmatch* pm, uh_pnp* pnp, const uh_track_args* args, uh_track_result* result);
/* test hook: the flattened kd-tree of the current frame (24-byte nodes {float divlow, divhigh; int32 left, right, leaf_begin;
 * int16 leaf_count, col}), the leaf index list, the root box {x.min, x.max, y.min, y.max} and the tree depth */
int  uh_projmatch_debug_tree(uh_projmatch* pm, int32_t* n_nodes, const void** nodes24, const uint32_t** leaf_idx,
                             double* root_box4, int32_t* max_depth);
/* the same build as a host-only function (no GPU needed): nodes24_out has room for 2n+2 nodes, leaf_idx_out for n entries */
int  uh_kdtree_build_host(const float* xy, int32_t n, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out,
                          double* root_box4, int32_t* max_depth);
/* test hooks of the device builder (csrc/kdbuild.hpp): the same outputs from the build kernel (n <= 4096; threads 0 = default, 256 or
 * 512), and — host only — the permutation its restatement of libstdc++'s std::sort gives n float keys (compared with std::sort itself) */
int  uh_kdtree_build_dev(uh_ctx* ctx, const float* xy, int32_t n, int32_t threads, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out,
                         double* root_box4, int32_t* max_depth);
int  uh_kdtree_sort_restated_host(const float* keys, int32_t n, uint32_t* perm_out);

/* ------------------------------------------------------------------------
 * Measurement hooks (used by scripts/time_ba.py and scripts/knn_push_bench.py; not part of the drop-in surface).
 * uh_ba_debug_clocks: the 64 phase timestamps the BA kernels of the latest LM step left behind (100 MHz wall clock; slots in
 * csrc/ba.hip, UH_BA_CLK).  uh_knn_debug_push_cycles: shader cycles of n heap pushes at row width k for the cross-lane and the
 * scalar formulation of the result heap and an empty loop (out3[0..2]).
 * ------------------------------------------------------------------------ */
int uh_ba_debug_clocks(uh_ba* ba, int64_t* out64);
/* a background launch of a chosen character (0 sleeping waves, 1 integer VALU, 2 streaming loads over d_buf, 3 LDS traffic) on ctx's
 * stream, for scripts/time_interference.py: what about a neighbouring launch slows the latency-bound BA chain down */
int uh_debug_background(uh_ctx* ctx, int mode, int blocks, int iters, const void* d_buf, size_t buf_bytes, void* d_sink);
int uh_knn_debug_push_cycles(uh_knn* knn, int k, int n, long long* out3);
/* shader-clock stamps of the pose-only solves that follow (on = 1) — out512[0] kernel entry, [1] matches staged, [2] rounds done,
 * [3] results posted, [4] passes over the matches, [5 ..] per-pass stamps; out512 (may be NULL) receives the stamps of the last solve and
 * MUST hold 512 entries (4096 bytes: the whole stamp block is copied) */
int uh_pnp_debug_clocks(uh_pnp* pnp, int on, long long* out512);

#ifdef __cplusplus
}
#endif
#endif /* UCOSLAM_HIP_H */
