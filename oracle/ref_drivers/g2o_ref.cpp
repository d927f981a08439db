// TEST INFRASTRUCTURE ONLY — driver around the REAL reference g2o (3rdparty/g2o core + vendored Eigen, compiled from
// /root/reference where they lie; output to oracle/_ref/, never into git).
//
// The reference's own vertex/edge header (src/optimization/typesg2o.h) cannot be compiled here: it includes
// map_types/marker.h -> OpenCV, which this image lacks, and writing a stand-in header is not allowed.  So this driver
// defines the three graph types it needs (SE3 pose vertex, XYZ point vertex, monocular reprojection edge) directly on
// g2o's BaseVertex/BaseBinaryEdge/SE3Quat, with the same parameterisation as typesg2o.h:36-79,249-323, and runs the
// reference's solver stack and schedule exactly as globaloptimizer_g2o.cpp:176-181,191-249,418-461 sets them up:
// BlockSolver_6_3 + LinearSolverEigen + OptimizationAlgorithmLevenberg, Huber sqrt(5.99), info = I/scaleFactor[octave],
// optimize(nIters,1) -> relabel chi2>5.99 / depth<=0 to level 1, drop kernels -> optimize(2*nIters,1).
// Everything numerical that matters for parity (LM control, Schur complement, LDLT, robustification, SE3 exp) is the
// reference's code; the edge Jacobian is cross-checked by central differences in tests/test_ba.py (test_oracle_edge_jacobian_central_differences).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "g2o/core/base_binary_edge.h"
#include "g2o/core/base_unary_edge.h"
#include "g2o/core/base_vertex.h"
#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/solvers/eigen/linear_solver_eigen.h"
#include "g2o/types/slam3d/se3quat.h"

namespace {

class PointV : public g2o::BaseVertex<3, g2o::Vector3> {
   public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    void setToOriginImpl() override { _estimate.setZero(); }
    void oplusImpl(const number_t* u) override { _estimate += Eigen::Map<const g2o::Vector3>(u); }
};

class PoseV : public g2o::BaseVertex<6, g2o::SE3Quat> {
   public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    void setToOriginImpl() override { _estimate = g2o::SE3Quat(); }
    void oplusImpl(const number_t* u) override {
        Eigen::Map<const g2o::Vector6> d(u);
        setEstimate(g2o::SE3Quat::exp(d) * estimate());
    }
};

class ReprojE : public g2o::BaseBinaryEdge<2, Eigen::Vector2d, PointV, PoseV> {
   public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    double fx = 1, fy = 1, cx = 0, cy = 0;
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    Eigen::Vector3d in_camera() const {
        return static_cast<const PoseV*>(_vertices[1])->estimate().map(static_cast<const PointV*>(_vertices[0])->estimate());
    }
    void computeError() override {
        const Eigen::Vector3d c = in_camera();
        _error = _measurement - Eigen::Vector2d((c[0] / c[2]) * fx + cx, (c[1] / c[2]) * fy + cy);
    }
    bool depth_positive() const { return in_camera()[2] > 0.0; }
    void linearizeOplus() override {
        const Eigen::Vector3d c = in_camera();
        const double x = c[0], y = c[1], z = c[2], z2 = z * z;
        Eigen::Matrix<double, 2, 3> dproj;   // d(-projection)/d(camera point), up to the -1/z factor
        dproj << fx, 0, -x / z * fx, 0, fy, -y / z * fy;
        _jacobianOplusXi = -1. / z * dproj * static_cast<const PoseV*>(_vertices[1])->estimate().rotation().toRotationMatrix();
        _jacobianOplusXj << x * y / z2 * fx, -(1 + (x * x / z2)) * fx, y / z * fx, -1. / z * fx, 0, x / z2 * fx,
            (1 + y * y / z2) * fy, -x * y / z2 * fy, -x / z * fy, 0, -1. / z * fy, y / z2 * fy;
    }
};

}  // namespace

extern "C" int g2o_ref_ba_optimize(int K, int P, int E, const float* poses_f2g, const uint8_t* fixed, const float* intr,
                                   const float* points, const int32_t* obs_pt, const int32_t* obs_kf, const float* obs_uv,
                                   const double* obs_invsigma, int nIters, float* poses_out, float* points_out,
                                   double* chi2_out, uint8_t* bad_out, int32_t* iters_out, double* pose_state_out) {
    g2o::SparseOptimizer opt;
    auto linearSolver = g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>>();
    auto* solver = new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolver_6_3>(std::move(linearSolver)));
    opt.setAlgorithm(solver);
    opt.setVerbose(false);
    std::vector<PoseV*> vp(K);
    for (int k = 0; k < K; k++) {   // frames first (vertex ids 0..K-1), points after
        const float* M = poses_f2g + 16 * k;
        Eigen::Matrix3d R;
        R << M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10];
        Eigen::Vector3d t(M[3], M[7], M[11]);
        auto* v = new PoseV();
        v->setEstimate(g2o::SE3Quat(R, t));
        v->setId(k);
        if (fixed[k]) v->setFixed(true);
        opt.addVertex(v);
        vp[k] = v;
    }
    std::vector<PointV*> vx(P);
    for (int p = 0; p < P; p++) {
        auto* v = new PointV();
        v->setEstimate(Eigen::Vector3d(points[3 * p], points[3 * p + 1], points[3 * p + 2]));
        v->setId(K + p);
        v->setMarginalized(true);
        opt.addVertex(v);
        vx[p] = v;
    }
    const double thHuber2D = std::sqrt(5.99), Chi2D = 5.99;
    std::vector<ReprojE*> ed(E);
    for (int e = 0; e < E; e++) {
        auto* ee = new ReprojE();
        const int k = obs_kf[e];
        ee->fx = intr[4 * k]; ee->fy = intr[4 * k + 1]; ee->cx = intr[4 * k + 2]; ee->cy = intr[4 * k + 3];
        ee->setVertex(0, vx[obs_pt[e]]);
        ee->setVertex(1, vp[k]);
        ee->setMeasurement(Eigen::Vector2d(obs_uv[2 * e], obs_uv[2 * e + 1]));
        ee->setInformation(Eigen::Matrix2d::Identity() * obs_invsigma[e]);
        auto* rk = new g2o::RobustKernelHuber();
        rk->setDelta(thHuber2D);
        ee->setRobustKernel(rk);
        opt.addEdge(ee);
        ed[e] = ee;
    }
    opt.initializeOptimization();
    iters_out[0] = opt.optimize(nIters, 1);
    for (int e = 0; e < E; e++) {
        if (ed[e]->chi2() > Chi2D || !ed[e]->depth_positive()) ed[e]->setLevel(1);
        ed[e]->setRobustKernel(0);
    }
    opt.initializeOptimization();
    iters_out[1] = opt.optimize(nIters * 2, 1);

    for (int k = 0; k < K; k++) {
        float* M = poses_out + 16 * k;
        if (fixed[k]) { std::memcpy(M, poses_f2g + 16 * k, 64); }
        else {
            Eigen::Matrix<double, 4, 4> H = vp[k]->estimate().to_homogeneous_matrix();
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M[i * 4 + j] = (float)H(i, j);
        }
        if (pose_state_out) {
            const g2o::SE3Quat& T = vp[k]->estimate();
            double* o = pose_state_out + 7 * k;
            o[0] = T.rotation().x(); o[1] = T.rotation().y(); o[2] = T.rotation().z(); o[3] = T.rotation().w();
            o[4] = T.translation()[0]; o[5] = T.translation()[1]; o[6] = T.translation()[2];
        }
    }
    for (int p = 0; p < P; p++) for (int a = 0; a < 3; a++) points_out[3 * p + a] = (float)vx[p]->estimate()[a];
    for (int e = 0; e < E; e++) {
        chi2_out[e] = ed[e]->chi2();
        bool bad = ed[e]->chi2() > Chi2D;
        if (!bad) {
            const float* M = poses_out + 16 * obs_kf[e];
            const float* X = points_out + 3 * obs_pt[e];
            if (M[8] * X[0] + M[9] * X[1] + M[10] * X[2] + M[11] < 0) bad = true;
        }
        bad_out[e] = bad;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ pose-only PnP
namespace {

class PoseOnlyE : public g2o::BaseUnaryEdge<2, Eigen::Vector2d, PoseV> {   // same parameterisation as typesg2o.h:590-650
   public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Eigen::Vector3d Xw;
    double fx = 1, fy = 1, cx = 0, cy = 0;
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
    Eigen::Vector3d in_camera() const { return static_cast<const PoseV*>(_vertices[0])->estimate().map(Xw); }
    void computeError() override {
        const Eigen::Vector3d c = in_camera();
        _error = _measurement - Eigen::Vector2d((c[0] / c[2]) * fx + cx, (c[1] / c[2]) * fy + cy);
    }
    void linearizeOplus() override {
        const Eigen::Vector3d c = in_camera();
        const double x = c[0], y = c[1], iz = 1.0 / c[2], iz2 = iz * iz;
        _jacobianOplusXi << x * y * iz2 * fx, -(1 + (x * x * iz2)) * fx, y * iz * fx, -iz * fx, 0, x * iz2 * fx,
            (1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, 0, -iz * fy, y * iz2 * fy;
    }
};

class WeightedHuber : public g2o::RobustKernel {   // the weight scales rho only (typesg2o.h:82-105)
   public:
    double W = 1, D = 1;
    void robustify(double e2, g2o::Vector3& rho) const override {
        const double dsqr = D * D;
        if (e2 <= dsqr) { rho[0] = W * e2; rho[1] = 1.; rho[2] = 0.; }
        else { const double sq = std::sqrt(e2); rho[0] = W * (2 * sq * D - dsqr); rho[1] = D / sq; rho[2] = -0.5 * rho[1] / e2; }
    }
};

}  // namespace

extern "C" int g2o_ref_pnp_solve(const float* pose_f2g, const float* intr4, int n, const float* p3d, const float* kp, const float* invsigma,
                                 const float* weight, float* pose_out, uint8_t* bad_out, int32_t* iters_out, double* state_out) {
    g2o::SparseOptimizer opt;
    auto linearSolver = g2o::make_unique<g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>>();
    opt.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<g2o::BlockSolver_6_3>(std::move(linearSolver))));
    auto toSE3 = [&]() {
        Eigen::Matrix3d R;
        R << pose_f2g[0], pose_f2g[1], pose_f2g[2], pose_f2g[4], pose_f2g[5], pose_f2g[6], pose_f2g[8], pose_f2g[9], pose_f2g[10];
        return g2o::SE3Quat(R, Eigen::Vector3d(pose_f2g[3], pose_f2g[7], pose_f2g[11]));
    };
    auto* cam = new PoseV();
    cam->setEstimate(toSE3());
    cam->setId(0);
    cam->setFixed(false);
    opt.addVertex(cam);
    const float Chi2D = 5.99f;
    const float thHuber2D = std::sqrt(5.99);
    std::vector<PoseOnlyE*> ed(n);
    for (int i = 0; i < n; i++) {
        auto* e = new PoseOnlyE();
        e->Xw = Eigen::Vector3d(p3d[3 * i], p3d[3 * i + 1], p3d[3 * i + 2]);
        e->fx = intr4[0]; e->fy = intr4[1]; e->cx = intr4[2]; e->cy = intr4[3];
        e->setVertex(0, cam);
        e->setMeasurement(Eigen::Vector2d(kp[2 * i], kp[2 * i + 1]));
        e->setInformation(Eigen::Matrix2d::Identity() * invsigma[i]);
        auto* rk = new WeightedHuber();
        rk->D = thHuber2D; rk->W = weight[i];
        e->setRobustKernel(rk);
        opt.addEdge(e);
        ed[i] = e;
    }
    std::vector<char> bad(n, 0);
    for (int it = 0; it < 4; it++) iters_out[it] = 0;
    if (n > 0)
        for (int it = 0; it < 4; it++) {
            cam->setEstimate(toSE3());
            opt.initializeOptimization(0);
            iters_out[it] = opt.optimize(10);
            int nGood = 0;
            for (int i = 0; i < n; i++) {
                if (bad[i]) ed[i]->computeError();
                bad[i] = ed[i]->chi2() > Chi2D;
                ed[i]->setLevel(bad[i] ? 1 : 0);
                if (it >= 2) ed[i]->setRobustKernel(nullptr);
                if (!bad[i]) nGood++;
            }
            if (nGood < 10) break;
        }
    Eigen::Matrix<double, 4, 4> Hm = cam->estimate().to_homogeneous_matrix();
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) pose_out[i * 4 + j] = (float)Hm(i, j);
    int good = 0;
    for (int i = 0; i < n; i++) { bad_out[i] = bad[i]; good += !bad[i]; }
    if (state_out) {
        const g2o::SE3Quat& T = cam->estimate();
        state_out[0] = T.rotation().x(); state_out[1] = T.rotation().y(); state_out[2] = T.rotation().z(); state_out[3] = T.rotation().w();
        state_out[4] = T.translation()[0]; state_out[5] = T.translation()[1]; state_out[6] = T.translation()[2];
    }
    return good;
}
