// Per-frame pose-only optimisation on MI355X (gfx950), fp64, behind the C ABI (uh_pnp_*).
//
// Semantic contract = PnPSolver::solvePnp for monocular matches without markers (reference file:line):
//   src/optimization/pnpsolver.cpp:116-409  four rounds; each restarts from the INPUT pose (:354), runs optimize(10)
//                                           (minChi2BetweenIter = 0), reclassifies every match with chi2 > 5.99 as outlier
//                                           (excluded edges get a fresh error first, :364), drops the robust kernels from the
//                                           third round on (:368) and stops early below 10 inliers (:379)
//   src/optimization/typesg2o.h:590-650     EdgeSE3ProjectXYZOnlyPose error and 2x6 Jacobian
//   src/optimization/typesg2o.h:82-105      WeightedHubberRobustKernel: the weight scales rho (the chi2 sums), not the Jacobian
//   3rdparty/g2o                            Levenberg loop, lambda init/update, SE3 exp — as in ba.hip
//
// MI355X design: the problem is one 6x6 system over a few hundred to a few thousand matches, i.e. pure latency.  The
// WHOLE solve — 4 rounds x <=10 iterations x <=10 trials, classification included — runs inside ONE persistent workgroup:
// edges are strided over 256 threads, the 28 sums of a linearisation go through one deterministic butterfly-transpose
// reduction, thread 0 does the 6x6 LDL^T / SE3 update / accept-reject between two barriers.  No kernel launches and no host
// synchronisation inside the solve; one launch, one result.
#include <cfloat>
#include <cmath>

#include "common.hpp"
#include "reduce.hpp"

namespace {

struct PnpArgs {
    const float* pose_in;    // 16
    const float* intr;       // fx fy cx cy
    int n;
    const float* p3d; const float* kp; const float* invsig; const float* weight;
    double* e_chi2;          // n
    unsigned char* flags;    // n x 3: active, robust, bad
    float* pose_out;         // 16
    unsigned char* bad_out;  // n
    int* result;             // [0] inliers, [1..4] outer iterations per round
    double* state_out;       // 7
};

struct PoseD { double q[4], t[3], Rt[12]; };

__device__ __forceinline__ void p_quat_to_R(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void p_quat_from_R(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {   // i = 0, j = 1, k = 2   (static indices: a runtime-indexed R[] would live in scratch)
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && !(R[8] > R[4])) {    // i = 1, j = 2, k = 0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[1] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
    } else {                                     // i = 2, j = 0, k = 1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[2] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
    }
}
__device__ __forceinline__ void p_quat_norm(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ void p_set_Rt(PoseD& T) { p_quat_to_R(T.q, T.Rt); T.Rt[9] = T.t[0]; T.Rt[10] = T.t[1]; T.Rt[11] = T.t[2]; }
__device__ void p_oplus(PoseD& T, const double* d) {   // T <- exp(d) * T
    const double w0 = d[0], w1 = d[1], w2 = d[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double O[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double O2[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
    double a, b, c1, c2;
    if (theta < 0.00001) { a = 1; b = 0.5; c1 = 0.5; c2 = 1.0 / 6.0; }
    else {
        double sn, cs;
        sincos(theta, &sn, &cs);
        a = sn / theta; b = (1 - cs) / (theta * theta); c1 = b; c2 = (theta - sn) / (theta * theta * theta);
    }
    double Rm[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; Rm[i] = I + a * O[i] + b * O2[i]; V[i] = I + c1 * O[i] + c2 * O2[i]; }
    double qe[4], te[3], RE[9];
    p_quat_from_R(Rm, qe);
    p_quat_norm(qe);
#pragma unroll
    for (int r = 0; r < 3; r++) te[r] = V[r * 3] * d[3] + V[r * 3 + 1] * d[4] + V[r * 3 + 2] * d[5];
    p_quat_to_R(qe, RE);
    const double* q = T.q;
    double qn[4] = {qe[3] * q[0] + qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1], qe[3] * q[1] + qe[1] * q[3] + qe[2] * q[0] - qe[0] * q[2],
                    qe[3] * q[2] + qe[2] * q[3] + qe[0] * q[1] - qe[1] * q[0], qe[3] * q[3] - qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2]};
    double tn[3];
#pragma unroll
    for (int r = 0; r < 3; r++) tn[r] = RE[r * 3] * T.t[0] + RE[r * 3 + 1] * T.t[1] + RE[r * 3 + 2] * T.t[2] + te[r];
    p_quat_norm(qn);
#pragma unroll
    for (int i = 0; i < 4; i++) T.q[i] = qn[i];
#pragma unroll
    for (int i = 0; i < 3; i++) T.t[i] = tn[i];
    p_set_Rt(T);
}
__device__ __forceinline__ bool p_solve6(const double* H, const double* b, double lam, double* x) {   // LDL^T, fails on a zero / non-finite pivot
    // fully unrolled with compile-time indices: the 6x6 system stays in registers (runtime-indexed arrays live in scratch,
    // one L2 round trip per access on the serial path of every LM trial)
    double M[6][6], L[6][6], d[6], id[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) { M[i][j] = H[i * 6 + j] + (i == j ? lam : 0.0); L[i][j] = 0; }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double dj = M[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k] * d[k];
        d[j] = dj;
        ok = ok && !(dj == 0.0 || !isfinite(dj));
        id[j] = 1.0 / dj;   // one division per pivot; the column and the substitution multiply by it
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = M[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * d[k];
            L[i][j] = v * id[j];
        }
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= id[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= L[k][i] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = y[i];
    return true;
}

// CACHED: the per-match inputs (map point, keypoint, 1/sigma, weight), chi2 and the three flag bytes live in LDS for the
// whole solve (44 bytes per match, n <= kPnpLdsMatches): the ~40 passes over the matches then cost LDS latency instead
// of an L2 round trip each.  Larger n runs the same code on the HBM arrays.
constexpr int kPnpLdsMatches = 3000;
// 8 waves = two per SIMD of the one CU this kernel lives on: the per-match fp64 chains (two divisions, a square root) of one wave
// fill the latency gaps of the other; 600 matches are then one or two per thread
constexpr int kPnpThreads = 512, kPnpWaves = kPnpThreads / 64;

template <bool CACHED>
__global__ __launch_bounds__(kPnpThreads) void pnp_solve_kernel(PnpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_cache[];
    __shared__ double s_part[kPnpWaves * 28], s_sum[28], s_red[kPnpWaves];
    __shared__ PoseD s_T, s_T0, s_bak;
    __shared__ double s_H[36], s_b[6], s_x[6];
    __shared__ double s_lambda, s_ni, s_currentChi, s_lastChiRaw, s_rho;
    __shared__ int s_ok2, s_again, s_ok, s_iter_cont, s_good;
    __shared__ float s_prev, s_cur;
    const int tid = threadIdx.x, n = A.n;
    const double fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];
    const double delta = (double)sqrtf(5.99f), dsqr = delta * delta;
    // one set of names for both instantiations; in each the pointers have a single provenance (LDS or HBM), so the
    // compiler emits ds_* or global_* accesses, never flat ones
    double* e_chi2; const float* p3d; const float* kpt; const float* invsig; const float* weight;
    unsigned char* active; unsigned char* robust; unsigned char* bad;
    if constexpr (CACHED) {
        e_chi2 = reinterpret_cast<double*>(s_cache);
        float* c_p3d = reinterpret_cast<float*>(s_cache + 8 * (size_t)n);
        float* c_kp = c_p3d + 3 * (size_t)n;
        float* c_is = c_kp + 2 * (size_t)n;
        float* c_w = c_is + n;
        active = reinterpret_cast<unsigned char*>(c_w + n);
        robust = active + n; bad = robust + n;
        for (int i = tid; i < 3 * n; i += kPnpThreads) c_p3d[i] = A.p3d[i];
        for (int i = tid; i < 2 * n; i += kPnpThreads) c_kp[i] = A.kp[i];
        for (int i = tid; i < n; i += kPnpThreads) { c_is[i] = A.invsig[i]; c_w[i] = A.weight[i]; }
        p3d = c_p3d; kpt = c_kp; invsig = c_is; weight = c_w;
    } else {
        e_chi2 = A.e_chi2; p3d = A.p3d; kpt = A.kp; invsig = A.invsig; weight = A.weight;
        active = A.flags; robust = A.flags + n; bad = A.flags + 2 * (size_t)n;
    }
    for (int e = tid; e < n; e += kPnpThreads) { active[e] = 1; robust[e] = 1; bad[e] = 0; e_chi2[e] = 0; }
    if (tid == 0) {
        const float* M = A.pose_in;
        const double R0[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        p_quat_from_R(R0, s_T0.q);
        p_quat_norm(s_T0.q);
        s_T0.t[0] = M[3]; s_T0.t[1] = M[7]; s_T0.t[2] = M[11];
        p_set_Rt(s_T0);
        s_T = s_T0;
        for (int i = 0; i < 5; i++) A.result[i] = 0;
    }
    __syncthreads();

    auto edge_err = [&](int e, const double* Rt, double& ex, double& ey, double* pc) {
        const double X0 = p3d[3 * e], X1 = p3d[3 * e + 1], X2 = p3d[3 * e + 2];
        pc[0] = Rt[0] * X0 + Rt[1] * X1 + Rt[2] * X2 + Rt[9];
        pc[1] = Rt[3] * X0 + Rt[4] * X1 + Rt[5] * X2 + Rt[10];
        pc[2] = Rt[6] * X0 + Rt[7] * X1 + Rt[8] * X2 + Rt[11];
        ex = (double)kpt[2 * e] - ((pc[0] / pc[2]) * fx + cx);
        ey = (double)kpt[2 * e + 1] - ((pc[1] / pc[2]) * fy + cy);
    };
    auto robchi = [&](int e, double c) -> double {
        if (!robust[e]) return c;
        const double w = weight[e];
        return (c <= dsqr) ? w * c : w * (2 * sqrt(c) * delta - dsqr);
    };

    for (int round = 0; round < 4 && n > 0; round++) {
        if (tid == 0) { s_T = s_T0; s_prev = FLT_MAX; s_cur = FLT_MAX; s_ok = 1; s_iter_cont = 1; }
        __syncthreads();
        int done = 0;
        for (int it = 0; it < 10; it++) {
            if (!s_iter_cont) break;     // uniform (read after a barrier)
            // ---- linearise at the current pose: errors, chi2, H, b
            double acc[28];
#pragma unroll
            for (int i = 0; i < 28; i++) acc[i] = 0;
            {
                double Rt[12];
#pragma unroll
                for (int i = 0; i < 12; i++) Rt[i] = s_T.Rt[i];
                for (int e = tid; e < n; e += kPnpThreads) {
                    if (!active[e]) continue;
                    double ex, ey, pc[3];
                    edge_err(e, Rt, ex, ey, pc);
                    const double w = invsig[e];
                    const double c = w * (ex * ex + ey * ey);
                    e_chi2[e] = c;
                    acc[27] += robchi(e, c);
                    const double X = pc[0], Y = pc[1], invz = 1.0 / pc[2], invz2 = invz * invz;
                    const double J[12] = {X * Y * invz2 * fx, -(1 + (X * X * invz2)) * fx, Y * invz * fx, -invz * fx, 0, X * invz2 * fx,
                                          (1 + Y * Y * invz2) * fy, -X * Y * invz2 * fy, -X * invz * fy, 0, -invz * fy, Y * invz2 * fy};
                    double rho1 = 1.0;
                    if (robust[e] && c > dsqr) rho1 = delta / sqrt(c);
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int cc = a; cc < 6; cc++) acc[q++] += (rho1 * w) * (J[a] * J[cc] + J[6 + a] * J[6 + cc]);
#pragma unroll
                    for (int a = 0; a < 6; a++) acc[21 + a] -= rho1 * (J[a] * w * ex + J[6 + a] * w * ey);
                }
            }
            block_sum_vec<28, kPnpWaves>(acc, s_part, s_sum);
            if (tid == 0) {
                int q = 0;
                for (int a = 0; a < 6; a++) for (int cc = a; cc < 6; cc++) { s_H[a * 6 + cc] = s_sum[q]; s_H[cc * 6 + a] = s_sum[q]; q++; }
                for (int a = 0; a < 6; a++) s_b[a] = s_sum[21 + a];
                s_currentChi = s_sum[27];
                if (it == 0) { double m = 0; for (int j = 0; j < 6; j++) m = fmax(fabs(s_H[j * 7]), m); s_lambda = 1e-5 * m; s_ni = 2; }
                const float t = s_prev; s_prev = s_cur; s_cur = t;   // swap(prevChi2, curChi2) at loop entry
            }
            __syncthreads();
            // ---- Levenberg trial loop
            int qmax = 0;
            double rho = 0;
            for (;;) {
                if (tid == 0) {
                    s_bak = s_T;
                    s_ok2 = p_solve6(s_H, s_b, s_lambda, s_x) ? 1 : 0;
                    if (s_ok2) p_oplus(s_T, s_x);
                }
                __syncthreads();
                double part = 0;
                {
                    double Rt[12];
#pragma unroll
                    for (int i = 0; i < 12; i++) Rt[i] = s_T.Rt[i];
                    for (int e = tid; e < n; e += kPnpThreads) {
                        if (!active[e]) continue;
                        double ex, ey, pc[3];
                        edge_err(e, Rt, ex, ey, pc);
                        const double c = (double)invsig[e] * (ex * ex + ey * ey);
                        e_chi2[e] = c;
                        part += robchi(e, c);
                    }
                }
                const double tempRaw = block_sum<kPnpWaves>(part, s_red);   // valid in every thread
                if (tid == 0) {
                    s_lastChiRaw = tempRaw;
                    double tempChi = s_ok2 ? tempRaw : DBL_MAX;
                    double r = s_currentChi - tempChi, scale = 0;
                    for (int i = 0; i < 6; i++) scale += s_x[i] * (s_lambda * s_x[i] + s_b[i]);
                    scale += 1e-3;
                    r /= scale;
                    bool lam_finite = true;
                    if (r > 0 && isfinite(tempChi)) {
                        const double t3 = 2 * r - 1;
                        double alpha = 1. - t3 * t3 * t3;
                        alpha = fmin(alpha, 2. / 3.);
                        s_lambda *= fmax(1. / 3., alpha);
                        s_ni = 2;
                        s_currentChi = tempChi;
                    } else {
                        s_lambda *= s_ni; s_ni *= 2; s_T = s_bak;
                        if (!isfinite(s_lambda)) lam_finite = false;
                    }
                    s_x[0] = s_x[0];
                    s_rho = r;                       // publish rho
                    s_again = lam_finite ? 1 : 0;    // 0 -> break before qmax++
                }
                __syncthreads();
                rho = s_rho;
                const int lam_ok = s_again;
                __syncthreads();
                if (!lam_ok) break;
                qmax++;
                if (!(rho < 0 && qmax < 10)) break;
            }
            done++;
            if (tid == 0) {
                const bool terminate = (qmax == 10 || rho == 0 || !isfinite(s_lambda));
                s_cur = (float)s_lastChiRaw;
                const float diff = s_prev - s_cur;
                s_iter_cont = (!terminate && diff > 0.f) ? 1 : 0;
            }
            __syncthreads();
        }
        // ---- classification (:358-371)
        int good = 0;
        {
            double Rt[12];
#pragma unroll
            for (int i = 0; i < 12; i++) Rt[i] = s_T.Rt[i];
            for (int e = tid; e < n; e += kPnpThreads) {
                double c = e_chi2[e];
                if (bad[e]) {
                    double ex, ey, pc[3];
                    edge_err(e, Rt, ex, ey, pc);
                    c = (double)invsig[e] * (ex * ex + ey * ey);
                    e_chi2[e] = c;
                }
                const bool b = c > (double)5.99f;
                bad[e] = b; active[e] = !b;
                if (round >= 2) robust[e] = 0;
                good += !b;
            }
        }
        const double gsum = block_sum<kPnpWaves>((double)good, s_red);
        if (tid == 0) { A.result[1 + round] = done; s_good = (int)gsum; }
        __syncthreads();
        if (s_good < 10) break;
    }
    __syncthreads();
    int good = 0;
    for (int e = tid; e < n; e += kPnpThreads) { A.bad_out[e] = bad[e]; good += !bad[e]; }
    const double gsum = block_sum<kPnpWaves>((double)good, s_red);
    if (tid == 0) {
        A.result[0] = (int)gsum;
        float* M = A.pose_out;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = (float)s_T.Rt[r * 3 + c]; M[r * 4 + 3] = (float)s_T.t[r]; }
        M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
        if (A.state_out) { for (int i = 0; i < 4; i++) A.state_out[i] = s_T.q[i]; for (int i = 0; i < 3; i++) A.state_out[4 + i] = s_T.t[i]; }
    }
}

}  // namespace

struct uh_pnp {
    uh_ctx* ctx = nullptr;
    uh::DevBuf d_in, d_work, d_out;
    bool attr_set = false;
};

extern "C" {

int uh_pnp_create(uh_ctx* ctx, uh_pnp** out) {
    UH_REQUIRE(ctx && out, "uh_pnp_create: NULL argument");
    uh_pnp* p = new uh_pnp();
    p->ctx = ctx;
    *out = p;
    return UH_OK;
}
void uh_pnp_destroy(uh_pnp* p) { delete p; }

// Everything resident in HBM; asynchronous on the context stream.  d_work: n*8 + n*3 bytes of scratch (8-byte aligned).
int uh_pnp_solve_dev(uh_pnp* p, const float* d_pose_f2g, const float* d_intr4, int n, const float* d_p3d, const float* d_kp,
                     const float* d_inv_sigma, const float* d_weight, void* d_work, float* d_pose_out, uint8_t* d_bad_out,
                     int32_t* d_result5, double* d_state7) {
    UH_REQUIRE(p && d_pose_f2g && d_intr4 && d_pose_out && d_result5, "uh_pnp_solve_dev: NULL argument");
    UH_REQUIRE(n >= 0, "uh_pnp_solve_dev: negative match count");
    if (n > 0) UH_REQUIRE(d_p3d && d_kp && d_inv_sigma && d_weight && d_work && d_bad_out, "uh_pnp_solve_dev: NULL match arrays");
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    PnpArgs A;
    A.pose_in = d_pose_f2g; A.intr = d_intr4; A.n = n; A.p3d = d_p3d; A.kp = d_kp; A.invsig = d_inv_sigma; A.weight = d_weight;
    A.e_chi2 = reinterpret_cast<double*>(d_work);
    A.flags = reinterpret_cast<unsigned char*>(d_work) + (size_t)n * 8;
    A.pose_out = d_pose_out; A.bad_out = d_bad_out; A.result = d_result5; A.state_out = d_state7;
    if (n <= kPnpLdsMatches) {
        const size_t lds = (size_t)n * 44 + 16;
        if (!p->attr_set) {
            UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pnp_solve_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kPnpLdsMatches * 44 + 16));
            p->attr_set = true;
        }
        UH_LAUNCH(p->ctx, pnp_solve_kernel<true>, dim3(1), dim3(kPnpThreads), lds, A);
    } else {
        UH_LAUNCH(p->ctx, pnp_solve_kernel<false>, dim3(1), dim3(kPnpThreads), 0, A);
    }
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

// Host-pointer form: PnPSolver::solvePnp(frame, map, matches, pose): returns the inlier count (>= 0) or a negative error.
int uh_pnp_solve(uh_pnp* p, const float* pose_f2g, const float* intr4, int n, const float* p3d, const float* kp, const float* inv_sigma,
                 const float* weight, float* pose_out, uint8_t* bad_out, int32_t* iters_out4, double* state_out7) {
    UH_REQUIRE(p && pose_f2g && intr4 && pose_out, "uh_pnp_solve: NULL argument");
    UH_REQUIRE(n >= 0, "uh_pnp_solve: negative match count");
    if (n == 0) { memcpy(pose_out, pose_f2g, 64); if (iters_out4) memset(iters_out4, 0, 16); return 0; }   // pnpsolver.cpp:149-150
    UH_REQUIRE(p3d && kp && inv_sigma && weight && bad_out, "uh_pnp_solve: NULL match arrays");
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    const size_t nf = (size_t)n;
    const size_t o_pose = 0, o_intr = 64, o_p3d = 128, o_kp = o_p3d + nf * 12, o_is = o_kp + nf * 8, o_w = o_is + nf * 4, in_bytes = o_w + nf * 4;
    int rc;
    if ((rc = p->d_in.reserve(in_bytes))) return rc;
    if ((rc = p->d_work.reserve(nf * 11 + 64))) return rc;
    const size_t o_pout = 0, o_res = 64, o_state = 96, o_bad = 160, out_bytes = o_bad + nf;
    if ((rc = p->d_out.reserve(out_bytes))) return rc;
    char* din = p->d_in.as<char>();
    UH_HIP_CHECK(hipMemcpyAsync(din + o_pose, pose_f2g, 64, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(din + o_intr, intr4, 16, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(din + o_p3d, p3d, nf * 12, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(din + o_kp, kp, nf * 8, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(din + o_is, inv_sigma, nf * 4, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(din + o_w, weight, nf * 4, hipMemcpyHostToDevice, st));
    char* dout = p->d_out.as<char>();
    rc = uh_pnp_solve_dev(p, (float*)(din + o_pose), (float*)(din + o_intr), n, (float*)(din + o_p3d), (float*)(din + o_kp), (float*)(din + o_is),
                          (float*)(din + o_w), p->d_work.p, (float*)(dout + o_pout), (uint8_t*)(dout + o_bad), (int32_t*)(dout + o_res),
                          (double*)(dout + o_state));
    if (rc) return rc;
    int32_t res[5];
    UH_HIP_CHECK(hipMemcpyAsync(pose_out, dout + o_pout, 64, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(res, dout + o_res, 20, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(bad_out, dout + o_bad, nf, hipMemcpyDeviceToHost, st));
    if (state_out7) UH_HIP_CHECK(hipMemcpyAsync(state_out7, dout + o_state, 56, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    if (iters_out4) memcpy(iters_out4, res + 1, 16);
    return res[0];
}

}  // extern "C"
