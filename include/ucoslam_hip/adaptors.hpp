// C++ host adaptors over the C ABI (ucoslam_hip.h): the reference's plugin surfaces for the tracking hot path.
//
//   ucoslam_hip::ORBextractor     ~ ucoslam::Feature2DSerializable / ORBextractor   (feature2dserializable.h:30-95)
//   ucoslam_hip::Index            ~ xflann::Index (Linear)                          (3rdparty/xflann/xflann/index.h:41-135)
//   ucoslam_hip::Vocabulary/fBow  ~ fbow::Vocabulary / fbow::fBow                   (3rdparty/fbow/fbow/fbow.h:54-116)
//   ucoslam_hip::GlobalOptimizer  ~ ucoslam::GlobalOptimizer                        (src/optimization/globaloptimizer.h:28-68)
//
// Same method names, argument meaning and error behaviour (std::runtime_error where the reference throws, `false` where
// xflann returns false).  This header needs no OpenCV: images/descriptors are plain pointers, keypoints are uh_keypoint
// (layout of cv::KeyPoint), matches are uh_dmatch (layout of cv::DMatch).  The classes that literally derive from the
// reference's bases (so that System/MapManager can hold them) are NOT in this header: they need the reference's headers and
// OpenCV, neither of which exists where this repository is built and tested, so they have never been compiled — INTEGRATION.md
// sections 1 and 4 list them as the binding a maintainer adds inside the reference tree, on top of the classes below.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../ucoslam_hip.h"
#include "flatten_ba.hpp"

namespace ucoslam_hip {

inline void check(int rc) { if (rc < 0) throw std::runtime_error(uh_last_error()); }

class Context {
   public:
    explicit Context(int device = 0, void* hip_stream = nullptr) { check(uh_ctx_create(device, hip_stream, &c_)); }
    ~Context() { uh_ctx_destroy(c_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    uh_ctx* get() const { return c_; }
    void synchronize() { check(uh_ctx_synchronize(c_)); }
   private:
    uh_ctx* c_ = nullptr;
};

// ------------------------------------------------------------------------------------------------ extractor
struct FeatParams {   // Feature2DSerializable::FeatParams (same defaults)
    int nthreads = -1, maxFeatures = 4000, nOctaveLevels = 8;
    float scaleFactor = 1.2f, sensitivity = 0;
    FeatParams() {}
    FeatParams(int MaxFeatures, int NOctaveLevels, float ScaleFactor, int NThreads)
        : nthreads(NThreads), maxFeatures(MaxFeatures), nOctaveLevels(NOctaveLevels), scaleFactor(ScaleFactor) {}
    bool operator==(const FeatParams& fp) const {
        return nthreads == fp.nthreads && maxFeatures == fp.maxFeatures && nOctaveLevels == fp.nOctaveLevels && scaleFactor == fp.scaleFactor;
    }
};

class ORBextractor {
   public:
    explicit ORBextractor(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_orb_create(ctx_->get(), &o_)); }
    ~ORBextractor() { uh_orb_destroy(o_); }
    static std::shared_ptr<ORBextractor> create(std::shared_ptr<Context> ctx) { return std::make_shared<ORBextractor>(std::move(ctx)); }

    // detectAndCompute(image, mask, keypoints, descriptors, params); mask is ignored like the reference (ORBextractor.cpp:1253)
    void detectAndCompute(const uint8_t* image, int width, int height, size_t stride, std::vector<uh_keypoint>& keypoints,
                          std::vector<uint8_t>& descriptors, const FeatParams& params) {
        uh_feat_params fp{params.nthreads, params.maxFeatures, params.nOctaveLevels, params.scaleFactor, params.sensitivity};
        check(uh_orb_set_params(o_, &fp));
        const int cap = std::max(uh_orb_max_keypoints(o_), 1);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(uh_orb_extract(o_, image, width, height, stride, keypoints.data(), descriptors.data(), cap, &n));
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
    }
    FeatParams getParams() const {
        uh_feat_params fp;
        check(uh_orb_get_params(o_, &fp));
        FeatParams r(fp.maxFeatures, fp.nOctaveLevels, fp.scaleFactor, fp.nthreads);
        r.sensitivity = fp.sensitivity;
        return r;
    }
    float getMinDescDistance() const { return 50; }          // ORBextractor.h:105
    void setSensitivity(float v) { check(uh_orb_set_sensitivity(o_, v)); }
    void doGaussianBlur(bool b) { check(uh_orb_set_blur(o_, b)); }
    // multi-GPU extraction of one frame: this instance extracts pyramid levels [first, end) (end < 0: all)
    void setLevelRange(int first, int end) { check(uh_orb_set_level_range(o_, first, end)); }
    uh_orb* handle() const { return o_; }
   private:
    std::shared_ptr<Context> ctx_;
    uh_orb* o_ = nullptr;
};

// ------------------------------------------------------------------------------------------------ kNN index
struct Matrix {   // xflann::Matrix as a non-owning view (types.h:275-400): rows x cols elements, row stride in bytes
    void* data = nullptr; int rows = 0, cols = 0; size_t stride = 0; int elem_size = 1;
    Matrix() {}
    Matrix(void* d, int r, int c, int esize, size_t s = 0) : data(d), rows(r), cols(c), stride(s ? s : (size_t)c * esize), elem_size(esize) {}
};

class Index {
   public:
    explicit Index(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_knn_create(ctx_->get(), &k_)); }
    ~Index() { uh_knn_destroy(k_); }
    // build(features, LinearParams): uint8 rows of 32 bytes
    void build(const Matrix& features) {
        if (features.elem_size != 1) throw std::runtime_error("Index::build: features must be XFLANN_8U");
        check(uh_knn_build(k_, static_cast<const uint8_t*>(features.data), features.rows, features.stride, features.cols));
    }
    // search(features, nn, indices, distances, KnnSearchParams(maxChecks, sorted)): outputs pre-allocated nq x nn int32
    bool search(const Matrix& q, int nn, Matrix indices, Matrix distances, bool sorted = false, int maxDist = -1) {
        if (q.rows != indices.rows || distances.rows != q.rows)
            throw std::runtime_error("knnsearch indices and distances must be already allocated with the same size as the number of features");
        if (distances.cols != nn || indices.cols != nn) throw std::runtime_error("knnsearch indices and distances number of cols must == nn");
        if (indices.elem_size != 4 || distances.elem_size != 4) throw std::runtime_error("Index::search undefined search distance type");
        const int rc = uh_knn_search(k_, static_cast<const uint8_t*>(q.data), q.rows, q.stride, nn, static_cast<int32_t*>(indices.data),
                                     static_cast<int32_t*>(distances.data), sorted, maxDist);
        if (rc == UH_ENOTBUILT) return false;   // index.cpp:82-85
        check(rc);
        return true;
    }
    // build(features, HKMeansParams(k, maxIters)) — the index FrameMatcher_Flann uses (framematcher.cpp:213); rows contiguous
    void buildKMeans(const Matrix& features, int k = 32, int maxIters = 0) {
        if (features.elem_size != 1 || features.cols != 32 || (features.rows > 1 && features.stride != 32))
            throw std::runtime_error("Index::buildKMeans: features must be contiguous XFLANN_8U rows of 32 bytes");
        check(uh_knn_build_kmeans(k_, static_cast<const uint8_t*>(features.data), features.rows, k, maxIters));
        kmeans_ = features.rows > 0;
    }
    // search(features, nn, indices, distances, KnnSearchParams(maxChecks, sorted)) on the k-means form
    bool searchKMeans(const Matrix& q, int nn, Matrix indices, Matrix distances, int maxChecks = 1, bool sorted = false) {
        if (q.rows != indices.rows || distances.rows != q.rows)
            throw std::runtime_error("KMeanIndex::knnsearch indices and distances must be already allocated with the same size as the number of features");
        if (distances.cols != nn || indices.cols != nn) throw std::runtime_error("KMeanIndex::knnsearch indices and distances number of cols must == nn");
        if (q.rows > 1 && q.stride != 32) throw std::runtime_error("Index::searchKMeans: query rows must be contiguous");
        const int rc = uh_knn_search_kmeans(k_, static_cast<const uint8_t*>(q.data), q.rows, nn, maxChecks, sorted, static_cast<int32_t*>(indices.data),
                                            static_cast<int32_t*>(distances.data));
        if (rc == UH_ENOTBUILT) return false;   // index.cpp:82-85
        check(rc);
        return true;
    }
    int size() const { return uh_knn_size(k_); }
    uh_knn* handle() const { return k_; }
   private:
    std::shared_ptr<Context> ctx_;
    uh_knn* k_ = nullptr;
    bool kmeans_ = false;
};

// ------------------------------------------------------------------------------------------------ bag of words
struct fBow : std::map<uint32_t, float> {
    static double score(const fBow& a, const fBow& b) {
        std::vector<uint32_t> ia, ib;
        std::vector<float> wa, wb;
        for (auto& e : a) { ia.push_back(e.first); wa.push_back(e.second); }
        for (auto& e : b) { ib.push_back(e.first); wb.push_back(e.second); }
        return uh_bow_score(ia.data(), wa.data(), (int)ia.size(), ib.data(), wb.data(), (int)ib.size());
    }
};
struct fBow2 : std::map<uint32_t, std::vector<uint32_t>> {};

class Vocabulary {
   public:
    explicit Vocabulary(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_bow_create(ctx_->get(), &b_)); }
    ~Vocabulary() { uh_bow_destroy(b_); }
    void fromStream(const void* bytes, size_t n) { check(uh_bow_load(b_, bytes, n)); }
    // transform(features, level, fBow&, fBow2&) (fbow.cpp:51-90)
    void transform(const uint8_t* features, int rows, int desc_bytes, size_t stride, int level, fBow& r1, fBow2& r2) {
        if (rows == 0) throw std::runtime_error("Vocabulary::transform No input data");
        std::vector<uint32_t> word(rows), node(rows);
        std::vector<float> weight(rows);
        std::vector<uint8_t> valid(rows);
        check(uh_bow_transform(b_, features, rows, stride, desc_bytes, level, word.data(), weight.data(), node.data(), valid.data()));
        r1.clear(); r2.clear();
        for (int i = 0; i < rows; i++) {
            if (word[i] != 0xFFFFFFFFu) r1[word[i]] += weight[i];
            if (valid[i]) r2[node[i]].push_back((uint32_t)i);
        }
    }
    // fBow transform(features): raw bag, then L2 normalisation (fbow.cpp:92-143)
    fBow transform(const uint8_t* features, int rows, int desc_bytes, size_t stride) {
        fBow r; fBow2 unused;
        transform(features, rows, desc_bytes, stride, -1, r, unused);
        double norm = 0;
        for (auto& e : r) norm += e.second * e.second;
        if (norm > 0.0) { const double inv = 1. / std::sqrt(norm); for (auto& e : r) e.second *= inv; }
        return r;
    }
   private:
    std::shared_ptr<Context> ctx_;
    uh_bow* b_ = nullptr;
};

// ------------------------------------------------------------------------------------------------ bundle adjustment
class GlobalOptimizer {
   public:
    using ParamSet = BAParamSet;   // globaloptimizer.h:31-47: used_frames, fixed_frames, fixFirstFrame, nIters, verbose
    explicit GlobalOptimizer(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_ba_create(ctx_->get(), &b_)); }
    ~GlobalOptimizer() { uh_ba_destroy(b_); }
    // GlobalOptimizer::create(type): "" / "hip" select this implementation, anything else throws (globaloptimizer.cpp:27-33)
    static std::shared_ptr<GlobalOptimizer> create(std::shared_ptr<Context> ctx, const std::string& type = "") {
        if (!type.empty() && type != "hip") throw std::runtime_error("GlobalOptimizer::create invalid type " + type);
        return std::make_shared<GlobalOptimizer>(std::move(ctx));
    }
    std::string getName() const { return "hip"; }
    // setParams(map, params) (globaloptimizer_g2o.cpp:77-401): the selection rules of flatten_ba.hpp, written straight into the optimiser's
    // pinned staging block; then ONE H2D copy + one kernel on the device.  Snapshots everything: the map may change afterwards.
    template <class MapView>
    void setParams(const MapView& map, const ParamSet& p) {
        StagingSink sink(b_);
        index_ = flatten_for_ba(map, p, sink);
        uh_ba_params bp{p.nIters, 0.0, 0.0, 1.0f};
        check(uh_ba_set_problem_staged(b_, (int)index_.frame_of.size(), (int)index_.point_of.size(), index_.n_obs, &bp));
        staged_ = true;
        obs_keep_.assign(sink.st.obs, sink.st.obs + index_.n_obs);   // (getResults names the bad associations by these)
    }
    // setParams on a map the caller has flattened already
    void setParams(const uh_ba_problem& problem, const ParamSet& p) {
        uh_ba_params bp{p.nIters, 0.0, 0.0, 1.0f};
        check(uh_ba_set_problem(b_, &problem, &bp));
        staged_ = false;
        obs_keep_.resize(problem.n_obs);
        for (int e = 0; e < problem.n_obs; e++) obs_keep_[e] = uh_ba_obs{problem.obs_point[e], problem.obs_frame[e], 0.f, 0.f, 0.0};
        index_ = FlatBAIndex{};
        index_.n_obs = problem.n_obs;
        K_ = problem.n_frames; P_ = problem.n_points;
    }
    void optimize(bool* stopASAP = nullptr) { check(uh_ba_optimize(b_, reinterpret_cast<const volatile uint8_t*>(stopASAP))); }
    // getResults(map) (:466-537): poses of the free frames, point coordinates, updatePointNormalAndDistances; fills getBadAssociations()
    template <class MapView>
    void getResults(MapView& map) {
        if (!staged_) throw std::runtime_error("GlobalOptimizer::getResults(map): setParams(map, ...) has not been called");
        std::vector<float> poses(16 * index_.frame_of.size()), points(3 * index_.point_of.size() + 1);
        std::vector<uint8_t> bad(index_.n_obs + 1);
        check(uh_ba_get_results(b_, poses.data(), points.data(), nullptr, bad.data(), nullptr));
        bad_ = apply_results(map, index_, poses.data(), points.data(), bad.data(), obs_keep_.data());
    }
    // getResults into plain arrays: poses K x 16 float, points P x 3 float (flattened indices)
    void getResults(std::vector<float>& poses_f2g, std::vector<float>& points) {
        const size_t K = staged_ ? index_.frame_of.size() : (size_t)K_, P = staged_ ? index_.point_of.size() : (size_t)P_;
        poses_f2g.resize(K * 16);
        points.resize(P * 3);
        std::vector<uint8_t> bad(obs_keep_.size() + 1);
        check(uh_ba_get_results(b_, poses_f2g.data(), points.data(), nullptr, bad.data(), nullptr));
        bad_.clear();
        for (size_t e = 0; e < obs_keep_.size(); e++)
            if (bad[e]) bad_.push_back(staged_ ? std::make_pair(index_.point_of[obs_keep_[e].point], index_.frame_of[obs_keep_[e].frame])
                                               : std::make_pair((uint32_t)obs_keep_[e].point, (uint32_t)obs_keep_[e].frame));
    }
    // one function to do everything (globaloptimizer_g2o.cpp:404-414)
    template <class MapView>
    void optimize(MapView& map, const ParamSet& p = ParamSet()) { setParams(map, p); optimize(); getResults(map); }
    std::vector<std::pair<uint32_t, uint32_t>> getBadAssociations() { return bad_; }
    const FlatBAIndex& index() const { return index_; }
    uh_ba* handle() const { return b_; }
   private:
    std::shared_ptr<Context> ctx_;
    uh_ba* b_ = nullptr;
    int K_ = 0, P_ = 0;
    bool staged_ = false;
    FlatBAIndex index_;
    std::vector<uh_ba_obs> obs_keep_;
    std::vector<std::pair<uint32_t, uint32_t>> bad_;
};

// ---- ucoslam::PnPSolver::solvePnp (optimization/pnpsolver.h:30-38) on flattened matches ----------------------------------
class PnPSolver {
   public:
    explicit PnPSolver(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_pnp_create(ctx_->get(), &p_)); }
    ~PnPSolver() { uh_pnp_destroy(p_); }
    // pose_io: row-major 4x4 float, refined in place; bad[i] = 1 marks an outlier match (the reference sets imgIdx = -1);
    // returns the number of inliers like solvePnp (pnpsolver.cpp:116-409)
    int solvePnp(float* pose_io, const float intr4[4], int n, const float* p3d, const float* kp, const float* inv_sigma, const float* weight,
                 std::vector<uint8_t>& bad) {
        bad.assign(n > 0 ? n : 1, 0);
        float out[16];
        int32_t iters[4];
        const int rc = uh_pnp_solve(p_, pose_io, intr4, n, p3d, kp, inv_sigma, weight, out, bad.data(), iters, nullptr);
        if (rc < 0) check(rc);
        std::copy(out, out + 16, pose_io);
        bad.resize(n);
        return rc;
    }
   private:
    std::shared_ptr<Context> ctx_;
    uh_pnp* p_ = nullptr;
};

// ---- ucoslam::Map::matchFrameToMapPoints (map.cpp:651-770) on a flattened frame / candidate list -------------------------
// ucoslam::FrameMatcher (TYPE_BOW), framematcher.cpp:395-535, on flattened frames (see uh_bow_frame)
class FrameMatcherBoW {
   public:
    explicit FrameMatcherBoW(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_bowmatch_create(ctx_->get(), &h_)); }
    ~FrameMatcherBoW() { uh_bowmatch_destroy(h_); }
    std::vector<uh_dmatch> matchEpipolar(const uh_bow_match_args& args) {
        std::vector<uh_dmatch> out(args.query.n_kpts > 0 ? args.query.n_kpts : 1);
        const int k = uh_bowmatch_match(h_, &args, out.data(), (int)out.size());
        if (k < 0) check(k);
        out.resize(k);
        return out;
    }
   private:
    std::shared_ptr<Context> ctx_;
    uh_bowmatch* h_ = nullptr;
};

class ProjectionMatcher {
   public:
    explicit ProjectionMatcher(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) { check(uh_projmatch_create(ctx_->get(), &h_)); }
    ~ProjectionMatcher() { uh_projmatch_destroy(h_); }
    // once per frame, where the reference calls Frame::create_kdtree (frame.h:124)
    void setFrame(const uh_proj_frame& f) { check(uh_projmatch_set_frame(h_, &f)); }
    // returns the reference's vector<cv::DMatch> (uh_dmatch has cv::DMatch's layout); visible (optional) gets the
    // markMapPointsAsVisible flags for the caller to apply MapPoint::setVisible()
    std::vector<uh_dmatch> matchFrameToMapPoints(const float pose_f2g[16], const uh_map_points& pts, float minDescDist, float maxRepjDist,
                                                 std::vector<uint8_t>* visible = nullptr) {
        std::vector<uh_dmatch> out(pts.n > 0 ? pts.n : 1);
        if (visible) visible->assign(pts.n > 0 ? pts.n : 1, 0);
        const int k = uh_projmatch_match(h_, pose_f2g, &pts, minDescDist, maxRepjDist, out.data(), (int32_t)out.size(), nullptr, nullptr,
                                         visible ? visible->data() : nullptr);
        if (k < 0) check(k);
        out.resize(k);
        if (visible) visible->resize(pts.n);
        return out;
    }
    // the tracker's search against the previous frame (System's private member at system.cpp:5930-6460, called at :6559-6565):
    // pose_f2g = the current frame's predicted pose, pts = the previous frame's keypoints that carry a good map point
    std::vector<uh_dmatch> matchFrameToPrevFrame(const float pose_f2g[16], const uh_prev_points& pts, float minDescDist, float maxRepjDist) {
        std::vector<uh_dmatch> out(pts.n > 0 ? pts.n : 1);
        const int k = uh_projmatch_match_prev(h_, pose_f2g, &pts, minDescDist, maxRepjDist, out.data(), (int32_t)out.size(), nullptr, nullptr);
        if (k < 0) check(k);
        out.resize(k);
        return out;
    }
   private:
    std::shared_ptr<Context> ctx_;
    uh_projmatch* h_ = nullptr;
};

}  // namespace ucoslam_hip
