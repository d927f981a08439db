// TEST INFRASTRUCTURE ONLY — CPU oracle for the local/global bundle adjustment stage.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// Sequential fp64 restatement of what GlobalOptimizerG2O computes for monocular reprojection edges:
//   src/optimization/globaloptimizer_g2o.cpp:418-464   two-pass schedule (nIters with Huber, relabel, 2*nIters without)
//   src/optimization/globaloptimizer_g2o.cpp:466-537   getResults (float poses/points, bad associations)
//   src/optimization/typesg2o.h:249-323                EdgeSE3ProjectXYZ error + analytic Jacobians
//   src/optimization/typesg2o.h:51-55,76-79            oplus: X += d ; T <- exp(d) * T
//   3rdparty/g2o/g2o/core/sparse_optimizer.cpp:366-436 outer loop, stop when float(chi2 drop) <= minChi2BetweenIter
//   3rdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:58-175  LM trial loop, lambda init/update, scale
//   3rdparty/g2o/g2o/core/base_binary_edge.hpp:83-150  robustified quadratic form (rho' scaling, rho'' dropped)
//   3rdparty/g2o/g2o/core/block_solver.hpp:315-447     Schur complement, back substitution; :525-566 lambda on all diagonals
//   3rdparty/g2o/g2o/core/robust_kernel_impl.cpp:65-78 Huber
//   3rdparty/g2o/g2o/types/slam3d/se3quat.h:276-311    SE3 exponential, quaternion product + normalisation
// The reduced pose system is solved with a dense LDL^T (g2o: Eigen SimplicialLDLT — same factorisation up to ordering).
// Pinned against the real g2o build (oracle/_ref/libg2o_ref.so) by tests/test_ba.py (tolerance, not bit-exact:
// floating-point summation order differs).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Pose { double q[4]; double t[3]; };   // q = (x,y,z,w)

inline void quat_normalize_pos(double* q) {   // SE3Quat::normalizeRotation
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}

// Eigen::Quaternion(Matrix3) — the conversion g2o::SE3Quat(R,t) performs
void quat_from_R(const double R[9], double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

void quat_to_R(const double* q, double R[9]) {   // Eigen::Quaternion::toRotationMatrix
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

inline void pose_map(const Pose& T, const double R[9], const double X[3], double out[3]) {
    for (int r = 0; r < 3; r++) out[r] = R[r * 3] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2] + T.t[r];
}

// T <- exp(d) * T   (d = [omega, upsilon])
void pose_oplus(Pose& T, const double d[6]) {
    const double w[3] = {d[0], d[1], d[2]}, u[3] = {d[3], d[4], d[5]};
    const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
    double a, b, c1, c2;
    if (theta < 0.00001) { a = 1; b = 0.5; c1 = 0.5; c2 = 1.0 / 6.0; }
    else {
        a = std::sin(theta) / theta;
        b = (1 - std::cos(theta)) / (theta * theta);
        c1 = b;
        c2 = (theta - std::sin(theta)) / std::pow(theta, 3);
    }
    double R[9], V[9];
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + a * O[i] + b * O2[i];
        V[i] = I + c1 * O[i] + c2 * O2[i];
    }
    Pose E;
    quat_from_R(R, E.q);
    quat_normalize_pos(E.q);   // SE3Quat(Quaternion, t) normalises
    for (int r = 0; r < 3; r++) E.t[r] = V[r * 3] * u[0] + V[r * 3 + 1] * u[1] + V[r * 3 + 2] * u[2];
    // E * T : r = qE*qT, t = qE*tT + tE
    double RE[9];
    quat_to_R(E.q, RE);
    const double* a4 = E.q; const double* b4 = T.q;
    double q[4];
    q[3] = a4[3] * b4[3] - a4[0] * b4[0] - a4[1] * b4[1] - a4[2] * b4[2];
    q[0] = a4[3] * b4[0] + a4[0] * b4[3] + a4[1] * b4[2] - a4[2] * b4[1];
    q[1] = a4[3] * b4[1] + a4[1] * b4[3] + a4[2] * b4[0] - a4[0] * b4[2];
    q[2] = a4[3] * b4[2] + a4[2] * b4[3] + a4[0] * b4[1] - a4[1] * b4[0];
    double t[3];
    for (int r = 0; r < 3; r++) t[r] = RE[r * 3] * T.t[0] + RE[r * 3 + 1] * T.t[1] + RE[r * 3 + 2] * T.t[2] + E.t[r];
    std::memcpy(T.q, q, sizeof(q));
    std::memcpy(T.t, t, sizeof(t));
    quat_normalize_pos(T.q);
}

bool inv3(const double* M, double* Inv) {   // Eigen 3x3 inverse (cofactors / determinant)
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    const double id = 1.0 / det;
    Inv[0] = c00 * id; Inv[1] = (M[2] * M[7] - M[1] * M[8]) * id; Inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    Inv[3] = c01 * id; Inv[4] = (M[0] * M[8] - M[2] * M[6]) * id; Inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    Inv[6] = c02 * id; Inv[7] = (M[1] * M[6] - M[0] * M[7]) * id; Inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    return true;
}

// EdgeSE3ProjectXYZ::linearizeOplus (typesg2o.h:275-314): A = d e / d X (2x3), B = d e / d pose (2x6: rotation first, then
// translation) at camera-frame point pc = R X + t.  Checked against central differences of the error in tests/test_ba.py
// (oracle_ba_edge_eval below).
inline void edge_jacobians(const double* Rk, const double pc[3], double fx, double fy, double A[6], double B[12]) {
    const double x = pc[0], y = pc[1], z = pc[2], z2 = z * z;
    // Ji (2x3) = -1/z * [fx 0 -x/z fx; 0 fy -y/z fy] * R
    const double t0[3] = {fx, 0, -x / z * fx}, t1[3] = {0, fy, -y / z * fy};
    for (int c = 0; c < 3; c++) {
        A[c] = -1. / z * (t0[0] * Rk[c] + t0[1] * Rk[3 + c] + t0[2] * Rk[6 + c]);
        A[3 + c] = -1. / z * (t1[0] * Rk[c] + t1[1] * Rk[3 + c] + t1[2] * Rk[6 + c]);
    }
    B[0] = x * y / z2 * fx; B[1] = -(1 + (x * x / z2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z2 * fx;
    B[6] = (1 + y * y / z2) * fy; B[7] = -x * y / z2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z2 * fy;
}

struct BA {
    int K = 0, P = 0, E = 0;
    std::vector<Pose> pose;
    std::vector<int> fixed, slot;          // slot: index of a free frame in the reduced system, -1 if fixed
    std::vector<double> intr;              // fx fy cx cy per frame
    std::vector<double> pts;               // 3 per point
    std::vector<int> e_pt, e_kf;
    std::vector<double> e_uv, e_w;         // measurement, scalar information
    std::vector<char> e_active, e_robust;  // level 0 / has Huber kernel
    std::vector<double> e_err, e_chi2;     // last computed error (2) and chi2 per edge (stale for inactive edges)
    double delta = std::sqrt(5.99);
    int nfree = 0;
    // system
    std::vector<double> Hpp, bp, Hll, bl, Hpl, S, bs, xp, xl, Dinv;
    // LM state (persists across outer iterations of one pass)
    double lambda = -1, ni = 2;
    const volatile uint8_t* stop = nullptr;

    bool terminate() const { return stop && *stop; }

    void compute_errors() {   // SparseOptimizer::computeActiveErrors
        std::vector<double> R(9 * K);
        for (int k = 0; k < K; k++) quat_to_R(pose[k].q, &R[9 * k]);
        for (int e = 0; e < E; e++) {
            if (!e_active[e]) continue;
            const int k = e_kf[e];
            double pc[3];
            pose_map(pose[k], &R[9 * k], &pts[3 * e_pt[e]], pc);
            const double* in = &intr[4 * k];
            const double ex = e_uv[2 * e] - ((pc[0] / pc[2]) * in[0] + in[2]);
            const double ey = e_uv[2 * e + 1] - ((pc[1] / pc[2]) * in[1] + in[3]);
            e_err[2 * e] = ex; e_err[2 * e + 1] = ey;
            e_chi2[e] = e_w[e] * (ex * ex + ey * ey);
        }
    }
    double robust_chi2() const {   // SparseOptimizer::activeRobustChi2
        double s = 0;
        const double dsqr = delta * delta;
        for (int e = 0; e < E; e++) {
            if (!e_active[e]) continue;
            const double c = e_chi2[e];
            if (e_robust[e]) s += (c <= dsqr) ? c : 2 * std::sqrt(c) * delta - dsqr;
            else s += c;
        }
        return s;
    }
    void build_system() {   // BlockSolver::buildSystem + EdgeSE3ProjectXYZ::linearizeOplus + constructQuadraticForm
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
        std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
        std::fill(Hpl.begin(), Hpl.end(), 0.0);
        std::vector<double> R(9 * K);
        for (int k = 0; k < K; k++) quat_to_R(pose[k].q, &R[9 * k]);
        const double dsqr = delta * delta;
        for (int e = 0; e < E; e++) {
            if (!e_active[e]) continue;
            const int k = e_kf[e], p = e_pt[e];
            const double* Rk = &R[9 * k];
            double pc[3];
            pose_map(pose[k], Rk, &pts[3 * p], pc);
            double A[6], B[12];
            edge_jacobians(Rk, pc, intr[4 * k], intr[4 * k + 1], A, B);
            double w = e_w[e];
            double rho1 = 1.0;
            if (e_robust[e]) { const double c = e_chi2[e]; if (c > dsqr) rho1 = delta / std::sqrt(c); }
            const double ww = rho1 * w;                                   // robustInformation
            const double r0 = -w * e_err[2 * e] * rho1, r1 = -w * e_err[2 * e + 1] * rho1;   // omega_r
            double* Hl = &Hll[9 * p];
            for (int a = 0; a < 3; a++) {
                bl[3 * p + a] += A[a] * r0 + A[3 + a] * r1;
                for (int b = 0; b < 3; b++) Hl[a * 3 + b] += ww * (A[a] * A[b] + A[3 + a] * A[3 + b]);
            }
            const int s = slot[k];
            if (s >= 0) {
                double* Hp = &Hpp[36 * s];
                for (int a = 0; a < 6; a++) {
                    bp[6 * s + a] += B[a] * r0 + B[6 + a] * r1;
                    for (int b = 0; b < 6; b++) Hp[a * 6 + b] += ww * (B[a] * B[b] + B[6 + a] * B[6 + b]);
                }
                double* Hx = &Hpl[18 * e];   // 6x3 pose-landmark block of this edge
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 3; b++) Hx[a * 3 + b] += ww * (B[a] * A[b] + B[6 + a] * A[3 + b]);
            }
        }
    }
    double lambda_init() const {   // computeLambdaInit: tau * max |diag|
        double m = 0;
        for (int s = 0; s < nfree; s++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[36 * s + 7 * j]), m);
        for (int p = 0; p < P; p++) if (pt_active[p]) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[9 * p + 4 * j]), m);
        return 1e-5 * m;
    }
    std::vector<char> pt_active;
    std::vector<std::vector<int>> pt_edges;   // active edges per point, insertion order

    bool solve(double lam) {   // BlockSolver::solve with lambda on every diagonal
        const int n = 6 * nfree;
        std::fill(S.begin(), S.end(), 0.0);
        for (int s = 0; s < nfree; s++)
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) S[(6 * s + a) * n + 6 * s + b] = Hpp[36 * s + a * 6 + b] + (a == b ? lam : 0.0);
        std::vector<double> coeff(n, 0.0);
        for (int p = 0; p < P; p++) {
            if (!pt_active[p]) continue;
            double D[9];
            for (int i = 0; i < 9; i++) D[i] = Hll[9 * p + i] + (i % 4 == 0 ? lam : 0.0);
            double* Di = &Dinv[9 * p];
            inv3(D, Di);
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * bl[3 * p] + Di[a * 3 + 1] * bl[3 * p + 1] + Di[a * 3 + 2] * bl[3 * p + 2];
            const std::vector<int>& ed = pt_edges[p];
            for (size_t i = 0; i < ed.size(); i++) {
                const int s1 = slot[e_kf[ed[i]]];
                if (s1 < 0) continue;
                const double* B1 = &Hpl[18 * ed[i]];
                double BD[18];
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 3; b++) BD[a * 3 + b] = B1[a * 3] * Di[b] + B1[a * 3 + 1] * Di[3 + b] + B1[a * 3 + 2] * Di[6 + b];
                for (int a = 0; a < 6; a++) coeff[6 * s1 + a] += B1[a * 3] * db[0] + B1[a * 3 + 1] * db[1] + B1[a * 3 + 2] * db[2];
                for (size_t j = 0; j < ed.size(); j++) {
                    const int s2 = slot[e_kf[ed[j]]];
                    if (s2 < s1) continue;   // upper block triangle only (block_solver.hpp:380-390)
                    if (s2 == s1 && j != i) continue;   // one edge per (point, frame)
                    const double* B2 = &Hpl[18 * ed[j]];
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++)
                            S[(6 * s1 + a) * n + 6 * s2 + b] -= BD[a * 3] * B2[b * 3] + BD[a * 3 + 1] * B2[b * 3 + 1] + BD[a * 3 + 2] * B2[b * 3 + 2];
                }
            }
        }
        for (int i = 0; i < n; i++) bs[i] = bp[i] - coeff[i];
        // dense LDL^T on the upper triangle (mirror first)
        for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) S[i * n + j] = S[j * n + i];
        std::vector<double> Ld((size_t)n * n, 0.0), d(n, 0.0);
        for (int j = 0; j < n; j++) {
            double dj = S[j * n + j];
            for (int k = 0; k < j; k++) dj -= Ld[j * n + k] * Ld[j * n + k] * d[k];
            d[j] = dj;
            if (dj == 0.0 || !std::isfinite(dj)) return false;
            for (int i = j + 1; i < n; i++) {
                double v = S[i * n + j];
                for (int k = 0; k < j; k++) v -= Ld[i * n + k] * Ld[j * n + k] * d[k];
                Ld[i * n + j] = v / dj;
            }
        }
        for (int i = 0; i < n; i++) { double v = bs[i]; for (int k = 0; k < i; k++) v -= Ld[i * n + k] * xp[k]; xp[i] = v; }
        for (int i = 0; i < n; i++) xp[i] /= d[i];
        for (int i = n - 1; i >= 0; i--) { double v = xp[i]; for (int k = i + 1; k < n; k++) v -= Ld[k * n + i] * xp[k]; xp[i] = v; }
        // landmarks: xl = Dinv (bl - Hpl^T xp)
        for (int p = 0; p < P; p++) {
            if (!pt_active[p]) { xl[3 * p] = xl[3 * p + 1] = xl[3 * p + 2] = 0; continue; }
            double c[3] = {bl[3 * p], bl[3 * p + 1], bl[3 * p + 2]};
            for (int e : pt_edges[p]) {
                const int s = slot[e_kf[e]];
                if (s < 0) continue;
                const double* B1 = &Hpl[18 * e];
                for (int b = 0; b < 3; b++) for (int a = 0; a < 6; a++) c[b] -= B1[a * 3 + b] * xp[6 * s + a];
            }
            const double* Di = &Dinv[9 * p];
            for (int a = 0; a < 3; a++) xl[3 * p + a] = Di[a * 3] * c[0] + Di[a * 3 + 1] * c[1] + Di[a * 3 + 2] * c[2];
        }
        return true;
    }

    enum Result { OK, Terminate };
    Result lm_solve(int iteration) {   // OptimizationAlgorithmLevenberg::solve
        compute_errors();
        double currentChi = robust_chi2();
        double tempChi = currentChi;
        build_system();
        if (iteration == 0) { lambda = lambda_init(); ni = 2; }
        double rho = 0;
        int qmax = 0;
        do {
            std::vector<Pose> pose_bak = pose;          // push
            std::vector<double> pts_bak = pts;
            const bool ok2 = solve(lambda);
            if (ok2) {
                for (int k = 0; k < K; k++) if (slot[k] >= 0) pose_oplus(pose[k], &xp[6 * slot[k]]);
                for (int p = 0; p < P; p++) if (pt_active[p]) for (int a = 0; a < 3; a++) pts[3 * p + a] += xl[3 * p + a];
            } else {
                // g2o still calls update(x) with whatever x holds; x is unspecified after a failed factorisation, and the
                // step is rejected below in every case, so the state is simply left untouched here.
            }
            compute_errors();
            tempChi = robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int i = 0; i < 6 * nfree; i++) scale += xp[i] * (lambda * xp[i] + bp[i]);
            for (int p = 0; p < P; p++) if (pt_active[p]) for (int a = 0; a < 3; a++) scale += xl[3 * p + a] * (lambda * xl[3 * p + a] + bl[3 * p + a]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                pose = pose_bak;                        // pop
                pts = pts_bak;
                if (!std::isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !terminate());
        if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) return Terminate;
        return OK;
    }

    void initialize() {   // initializeOptimization(level 0)
        pt_active.assign(P, 0);
        pt_edges.assign(P, {});
        for (int e = 0; e < E; e++) if (e_active[e]) { pt_active[e_pt[e]] = 1; pt_edges[e_pt[e]].push_back(e); }
    }

    int optimize(int iterations, float minChi2BetweenIter) {   // SparseOptimizer::optimize
        float prevChi2 = std::numeric_limits<float>::max(), curChi2 = prevChi2, diff = prevChi2;
        bool ok = true;
        int done = 0;
        for (int i = 0; i < iterations && !terminate() && ok && diff > minChi2BetweenIter; i++) {
            std::swap(prevChi2, curChi2);
            ok = (lm_solve(i) == OK);
            curChi2 = (float)robust_chi2();
            diff = prevChi2 - curChi2;
            ++done;
        }
        return done;
    }
};

}  // namespace

extern "C" {

// Flat problem description shared with the HIP C ABI (include/ucoslam_hip.h, uh_ba_*):
//  poses_f2g  K x 16 float (row-major 4x4, the reference's cv::Mat pose_f2g), fixed K, intr K x 4 float (fx fy cx cy)
//  points P x 3 float; observations E: point index, frame index, undistorted pixel (2 float), inv scale factor (double)
//  outputs: poses K x 16 float, points P x 3 float, chi2 E double, bad E uint8, iters[2] outer iterations per pass
int oracle_ba_optimize(int K, int P, int E, const float* poses_f2g, const uint8_t* fixed, const float* intr,
                       const float* points, const int32_t* obs_pt, const int32_t* obs_kf, const float* obs_uv,
                       const double* obs_invsigma, int nIters, const uint8_t* stop_flag, float* poses_out, float* points_out,
                       double* chi2_out, uint8_t* bad_out, int32_t* iters_out, double* pose_state_out /* K x 7, may be NULL */) {
    BA ba;
    ba.K = K; ba.P = P; ba.E = E;
    ba.pose.resize(K); ba.fixed.assign(fixed, fixed + K); ba.slot.assign(K, -1); ba.intr.resize(4 * K);
    for (int k = 0; k < K; k++) {
        const float* M = poses_f2g + 16 * k;
        double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        quat_from_R(R, ba.pose[k].q);
        quat_normalize_pos(ba.pose[k].q);
        ba.pose[k].t[0] = M[3]; ba.pose[k].t[1] = M[7]; ba.pose[k].t[2] = M[11];
        if (!fixed[k]) ba.slot[k] = ba.nfree++;
        for (int j = 0; j < 4; j++) ba.intr[4 * k + j] = intr[4 * k + j];
    }
    ba.pts.resize(3 * P);
    for (int i = 0; i < 3 * P; i++) ba.pts[i] = points[i];
    ba.e_pt.assign(obs_pt, obs_pt + E); ba.e_kf.assign(obs_kf, obs_kf + E);
    ba.e_uv.resize(2 * E); ba.e_w.resize(E);
    for (int e = 0; e < E; e++) { ba.e_uv[2 * e] = obs_uv[2 * e]; ba.e_uv[2 * e + 1] = obs_uv[2 * e + 1]; ba.e_w[e] = obs_invsigma[e]; }
    ba.e_active.assign(E, 1); ba.e_robust.assign(E, 1);
    ba.e_err.assign(2 * E, 0.0); ba.e_chi2.assign(E, 0.0);
    const int n = 6 * ba.nfree;
    ba.Hpp.assign(36 * (size_t)ba.nfree, 0); ba.bp.assign(n, 0); ba.Hll.assign(9 * (size_t)P, 0); ba.bl.assign(3 * (size_t)P, 0);
    ba.Hpl.assign(18 * (size_t)E, 0); ba.S.assign((size_t)n * n, 0); ba.bs.assign(n, 0); ba.xp.assign(n, 0);
    ba.xl.assign(3 * (size_t)P, 0); ba.Dinv.assign(9 * (size_t)P, 0);
    ba.stop = stop_flag;
    // pass 1 (globaloptimizer_g2o.cpp:421-428)
    ba.initialize();
    iters_out[0] = ba.optimize(nIters, 1.f);
    iters_out[1] = 0;
    if (!(stop_flag && *stop_flag)) {
        std::vector<double> R(9);
        for (int e = 0; e < E; e++) {   // :434-449
            quat_to_R(ba.pose[ba.e_kf[e]].q, R.data());
            double pc[3];
            pose_map(ba.pose[ba.e_kf[e]], R.data(), &ba.pts[3 * ba.e_pt[e]], pc);
            if (ba.e_chi2[e] > 5.99 || !(pc[2] > 0.0)) ba.e_active[e] = 0;
            ba.e_robust[e] = 0;
        }
        ba.initialize();                // :457-458
        iters_out[1] = ba.optimize(nIters * 2, 1.f);
    }
    // getResults (:466-537)
    for (int k = 0; k < K; k++) {
        float* M = poses_out + 16 * k;
        if (fixed[k]) { std::memcpy(M, poses_f2g + 16 * k, 64); continue; }
        double R[9];
        quat_to_R(ba.pose[k].q, R);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = (float)R[r * 3 + c]; M[r * 4 + 3] = (float)ba.pose[k].t[r]; }
        M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
    }
    for (int i = 0; i < 3 * P; i++) points_out[i] = (float)ba.pts[i];
    for (int e = 0; e < E; e++) {
        chi2_out[e] = ba.e_chi2[e];
        bool bad = ba.e_chi2[e] > 5.99;
        if (!bad) {   // pincam = pose_f2g(float) * point(float); z < 0
            const float* M = poses_out + 16 * ba.e_kf[e];
            const float* X = points_out + 3 * ba.e_pt[e];
            const float z = M[8] * X[0] + M[9] * X[1] + M[10] * X[2] + M[11];
            if (z < 0) bad = true;
        }
        bad_out[e] = bad;
    }
    if (pose_state_out)
        for (int k = 0; k < K; k++) { std::memcpy(pose_state_out + 7 * k, ba.pose[k].q, 32); std::memcpy(pose_state_out + 7 * k + 4, ba.pose[k].t, 24); }
    return 0;
}

}  // extern "C"

// One EdgeSE3ProjectXYZ at the state (exp(dpose) * T, X + dX): error (computeError, typesg2o.h:260-273) and the analytic Jacobians the
// optimisation uses.  dpose / dX may be NULL (= zero).  pose7 = qx qy qz qw tx ty tz.  Test hook: central-difference Jacobian check.
extern "C" void oracle_ba_edge_eval(const double* pose7, const double* X, const double* intr4, const double* uv, const double* dpose,
                                    const double* dX, double* err2, double* A6, double* B12) {
    Pose T;
    for (int i = 0; i < 4; i++) T.q[i] = pose7[i];
    for (int i = 0; i < 3; i++) T.t[i] = pose7[4 + i];
    if (dpose) pose_oplus(T, dpose);
    double Xp[3] = {X[0], X[1], X[2]};
    if (dX) for (int i = 0; i < 3; i++) Xp[i] += dX[i];
    double R[9], pc[3];
    quat_to_R(T.q, R);
    pose_map(T, R, Xp, pc);
    err2[0] = uv[0] - ((pc[0] / pc[2]) * intr4[0] + intr4[2]);
    err2[1] = uv[1] - ((pc[1] / pc[2]) * intr4[1] + intr4[3]);
    if (A6 && B12) edge_jacobians(R, pc, intr4[0], intr4[1], A6, B12);
}
