// TEST INFRASTRUCTURE ONLY — CPU oracle for the per-frame pose-only optimisation (PnPSolver::solvePnp, mono, no markers).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// Restates  src/optimization/pnpsolver.cpp:116-409  (4 rounds x optimize(10) restarting from the input pose, chi2 > 5.99
//           reclassification after every round, robust kernels dropped from the third round on, early exit below 10 inliers),
//           src/optimization/typesg2o.h:590-650     EdgeSE3ProjectXYZOnlyPose (error, analytic 2x6 Jacobian),
//           src/optimization/typesg2o.h:82-105      WeightedHubberRobustKernel (the weight scales rho only),
//           3rdparty/g2o: BaseUnaryEdge::constructQuadraticForm (base_unary_edge.hpp:55-78), Levenberg (as ba_oracle.cpp),
//           SparseOptimizer::optimize(iterations, minChi2BetweenIter = 0)  (sparse_optimizer.h:120).
// Pinned against the real g2o (oracle/_ref/libg2o_ref.so, driver g2o_ref_pnp_solve) by tests/test_pnp.py.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Pose { double q[4]; double t[3]; };

void quat_norm_pos(double* q) {
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] = -q[i];
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
void quat_from_R(const double R[9], double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) { t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
void quat_to_R(const double* q, double R[9]) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
void pose_oplus(Pose& T, const double d[6]) {   // T <- exp(d) * T, se3quat.h:276-311
    const double w[3] = {d[0], d[1], d[2]}, u[3] = {d[3], d[4], d[5]};
    const double theta = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
    double a, b, c1, c2;
    if (theta < 0.00001) { a = 1; b = 0.5; c1 = 0.5; c2 = 1.0 / 6.0; }
    else { a = std::sin(theta) / theta; b = (1 - std::cos(theta)) / (theta * theta); c1 = b; c2 = (theta - std::sin(theta)) / std::pow(theta, 3); }
    double R[9], V[9];
    for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + a * O[i] + b * O2[i]; V[i] = I + c1 * O[i] + c2 * O2[i]; }
    Pose E;
    quat_from_R(R, E.q);
    quat_norm_pos(E.q);
    for (int r = 0; r < 3; r++) E.t[r] = V[r * 3] * u[0] + V[r * 3 + 1] * u[1] + V[r * 3 + 2] * u[2];
    double RE[9];
    quat_to_R(E.q, RE);
    const double* x = E.q; const double* y = T.q;
    double q[4] = {x[3] * y[0] + x[0] * y[3] + x[1] * y[2] - x[2] * y[1], x[3] * y[1] + x[1] * y[3] + x[2] * y[0] - x[0] * y[2],
                   x[3] * y[2] + x[2] * y[3] + x[0] * y[1] - x[1] * y[0], x[3] * y[3] - x[0] * y[0] - x[1] * y[1] - x[2] * y[2]};
    double t[3];
    for (int r = 0; r < 3; r++) t[r] = RE[r * 3] * T.t[0] + RE[r * 3 + 1] * T.t[1] + RE[r * 3 + 2] * T.t[2] + E.t[r];
    std::memcpy(T.q, q, sizeof(q));
    std::memcpy(T.t, t, sizeof(t));
    quat_norm_pos(T.q);
}

struct PnP {
    int n;
    const float* p3d; const float* kp; const float* invsig; const float* weight;
    double fx, fy, cx, cy, delta;
    Pose T;
    std::vector<char> active, robust;
    std::vector<double> err, chi2;
    double H[36], b[6], x[6], lambda = -1, ni = 2;

    void edge_error(int e, const double R[9], double& ex, double& ey, double pc[3]) const {
        const float* X = p3d + 3 * e;
        for (int r = 0; r < 3; r++) pc[r] = R[r * 3] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2] + T.t[r];
        ex = kp[2 * e] - ((pc[0] / pc[2]) * fx + cx);
        ey = kp[2 * e + 1] - ((pc[1] / pc[2]) * fy + cy);
    }
    void compute_errors() {
        double R[9]; quat_to_R(T.q, R);
        for (int e = 0; e < n; e++) {
            if (!active[e]) continue;
            double pc[3];
            edge_error(e, R, err[2 * e], err[2 * e + 1], pc);
            chi2[e] = (double)invsig[e] * (err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]);
        }
    }
    double robust_chi2() const {
        const double dsqr = delta * delta;
        double s = 0;
        for (int e = 0; e < n; e++) {
            if (!active[e]) continue;
            const double c = chi2[e];
            if (robust[e]) s += (c <= dsqr) ? (double)weight[e] * c : (double)weight[e] * (2 * std::sqrt(c) * delta - dsqr);
            else s += c;
        }
        return s;
    }
    void build() {
        std::memset(H, 0, sizeof(H)); std::memset(b, 0, sizeof(b));
        double R[9]; quat_to_R(T.q, R);
        const double dsqr = delta * delta;
        for (int e = 0; e < n; e++) {
            if (!active[e]) continue;
            double pc[3], ex, ey;
            edge_error(e, R, ex, ey, pc);
            const double X = pc[0], Y = pc[1], invz = 1.0 / pc[2], invz2 = invz * invz;
            const double J[12] = {X * Y * invz2 * fx, -(1 + (X * X * invz2)) * fx, Y * invz * fx, -invz * fx, 0, X * invz2 * fx,
                                  (1 + Y * Y * invz2) * fy, -X * Y * invz2 * fy, -X * invz * fy, 0, -invz * fy, Y * invz2 * fy};
            const double w = invsig[e];
            double rho1 = 1.0;
            if (robust[e] && chi2[e] > dsqr) rho1 = delta / std::sqrt(chi2[e]);
            for (int a = 0; a < 6; a++) {
                b[a] -= rho1 * (J[a] * w * err[2 * e] + J[6 + a] * w * err[2 * e + 1]);
                for (int c = 0; c < 6; c++) H[a * 6 + c] += (rho1 * w) * (J[a] * J[c] + J[6 + a] * J[6 + c]);
            }
        }
    }
    bool solve(double lam) {   // 6x6 LDL^T
        double M[36], L[36] = {0}, d[6];
        for (int i = 0; i < 36; i++) M[i] = H[i] + ((i % 7 == 0) ? lam : 0.0);
        for (int j = 0; j < 6; j++) {
            double dj = M[j * 6 + j];
            for (int k = 0; k < j; k++) dj -= L[j * 6 + k] * L[j * 6 + k] * d[k];
            d[j] = dj;
            if (dj == 0.0 || !std::isfinite(dj)) return false;
            for (int i = j + 1; i < 6; i++) { double v = M[i * 6 + j]; for (int k = 0; k < j; k++) v -= L[i * 6 + k] * L[j * 6 + k] * d[k]; L[i * 6 + j] = v / dj; }
        }
        for (int i = 0; i < 6; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[i * 6 + k] * x[k]; x[i] = v; }
        for (int i = 0; i < 6; i++) x[i] /= d[i];
        for (int i = 5; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < 6; k++) v -= L[k * 6 + i] * x[k]; x[i] = v; }
        return true;
    }
    bool lm_solve(int iteration) {   // true = OK, false = Terminate
        compute_errors();
        double currentChi = robust_chi2(), tempChi = currentChi;
        build();
        if (iteration == 0) { double m = 0; for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 7]), m); lambda = 1e-5 * m; ni = 2; }
        double rho = 0;
        int qmax = 0;
        do {
            const Pose bak = T;
            const bool ok2 = solve(lambda);
            if (ok2) pose_oplus(T, x);
            compute_errors();
            tempChi = robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = 0;
            for (int i = 0; i < 6; i++) scale += x[i] * (lambda * x[i] + b[i]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2; T = bak;
                if (!std::isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        return !(qmax == 10 || rho == 0 || !std::isfinite(lambda));
    }
    int optimize(int iterations) {
        float prev = std::numeric_limits<float>::max(), cur = prev, diff = prev;
        bool ok = true;
        int done = 0;
        for (int i = 0; i < iterations && ok && diff > 0.f; i++) {
            std::swap(prev, cur);
            ok = lm_solve(i);
            cur = (float)robust_chi2();
            diff = prev - cur;
            ++done;
        }
        return done;
    }
};

}  // namespace

extern "C" int oracle_pnp_solve(const float* pose_f2g, const float* intr4, int n, const float* p3d, const float* kp, const float* invsigma,
                                const float* weight, float* pose_out, uint8_t* bad_out, int32_t* iters_out /*4*/, double* state_out /*7*/) {
    PnP s;
    s.n = n; s.p3d = p3d; s.kp = kp; s.invsig = invsigma; s.weight = weight;
    s.fx = intr4[0]; s.fy = intr4[1]; s.cx = intr4[2]; s.cy = intr4[3];
    s.delta = (double)(float)std::sqrt(5.99);   // const float thHuber2D = sqrt(5.99)
    const double R0[9] = {pose_f2g[0], pose_f2g[1], pose_f2g[2], pose_f2g[4], pose_f2g[5], pose_f2g[6], pose_f2g[8], pose_f2g[9], pose_f2g[10]};
    Pose T0;
    quat_from_R(R0, T0.q);
    quat_norm_pos(T0.q);
    T0.t[0] = pose_f2g[3]; T0.t[1] = pose_f2g[7]; T0.t[2] = pose_f2g[11];
    s.T = T0;
    s.active.assign(n, 1); s.robust.assign(n, 1); s.err.assign(2 * (size_t)n, 0.0); s.chi2.assign(n, 0.0);
    std::vector<char> bad(n, 0);
    for (int it = 0; it < 4; it++) iters_out[it] = 0;
    const float Chi2D = 5.99f;
    if (n > 0)
        for (int it = 0; it < 4; it++) {
            s.T = T0;                                   // every round restarts from the input pose (:354)
            iters_out[it] = s.optimize(10);
            int nGood = 0;
            double R[9]; quat_to_R(s.T.q, R);
            for (int e = 0; e < n; e++) {
                if (bad[e]) {                           // excluded edges get a fresh error at the new pose (:364)
                    double pc[3];
                    s.edge_error(e, R, s.err[2 * e], s.err[2 * e + 1], pc);
                    s.chi2[e] = (double)invsigma[e] * (s.err[2 * e] * s.err[2 * e] + s.err[2 * e + 1] * s.err[2 * e + 1]);
                }
                bad[e] = s.chi2[e] > (double)Chi2D;
                s.active[e] = !bad[e];
                if (it >= 2) s.robust[e] = 0;
                if (!bad[e]) nGood++;
            }
            if (nGood < 10) break;
        }
    double R[9]; quat_to_R(s.T.q, R);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pose_out[r * 4 + c] = (float)R[r * 3 + c]; pose_out[r * 4 + 3] = (float)s.T.t[r]; }
    pose_out[12] = pose_out[13] = pose_out[14] = 0.f; pose_out[15] = 1.f;
    int good = 0;
    for (int e = 0; e < n; e++) { bad_out[e] = bad[e]; good += !bad[e]; }
    if (state_out) { std::memcpy(state_out, s.T.q, 32); std::memcpy(state_out + 4, s.T.t, 24); }
    return good;
}
