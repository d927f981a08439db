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
// MI355X design: one 6x6 system over a few hundred to a few thousand matches, ~20 dependent Levenberg trials — pure latency, so
// the WHOLE solve runs inside ONE workgroup of one launch and every trial costs ONE pass over the matches and ONE barrier:
//   * a pass evaluates a pose: per match the error, chi2, robust weight AND the Jacobian products, i.e. the trial's chi2 and the
//     normal equations of the NEXT linearisation come out of the same pass (g2o re-linearises at the pose it has just accepted and
//     gets exactly these numbers; after a rejection H and b of the old pose are still valid) — one pass per trial instead of two;
//     the classification between two rounds is folded into the first pass of the next round;
//   * the 29 sums (21 + 6 + chi2 + inlier count) go through one butterfly-transpose reduction per wave, one LDS exchange and one
//     barrier (double-buffered by pass parity); every wave then adds the eight partials in wave order and holds the totals as
//     wave-uniform values;
//   * every wave solves the 6x6 system, applies the SE3 update and takes the accept / reject decision REDUNDANTLY from those
//     identical totals: no wave waits for another one between two passes, no single-thread section, no second barrier;
//   * the matches live in LDS for the whole solve (37 bytes each, n <= kPnpLdsMatches; larger n runs the same code on HBM arrays);
//   * host form: the inputs are read straight from a pinned, device-visible staging block, the results go straight into pinned
//     memory behind a completion word the host polls — one launch, no copy engine, no stream synchronisation.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>

#include "common.hpp"
#include "reduce.hpp"

namespace {

struct PnpArgs {
    const float* pose_in;    // 16
    const float* intr;       // fx fy cx cy
    int n;
    const int* n_dev;        // NULL, or the match count in device memory (written by an earlier launch of the stream; n is then its upper bound)
    const float* p3d; const float* kp; const float* invsig; const float* weight;
    void* work;              // n x 32 bytes of scratch (only used when the matches do not fit LDS)
    float* pose_out;         // 16
    unsigned char* bad_out;  // n
    int* result;             // [0] inliers, [1..4] outer iterations per round
    double* state_out;       // 7 or NULL
    unsigned long long* host_done;   // NULL, or a word in pinned host memory that receives done_word after everything else
    unsigned long long done_word;
    long long* clk;          // NULL, or 8 timestamps (s_memtime) for scripts/time_pnp.py
    uh::PnpDecide dec;       // dyn17 != NULL: the tracker's decision rides on this solve (common.hpp)
};

// one thread: the decision of system.cpp:6762-6881 from this solve's inlier count; M = the pose it returns (16 floats)
__device__ void pnp_decide(const PnpArgs& A, int inliers, const float* M) {
    const int tracked = inliers >= A.dec.min_inliers ? 1 : 0;
    const float* T = tracked ? M : A.pose_in;
    float* d = A.dec.dyn17;
    for (int i = 0; i < 12; i++) d[i] = T[i];
    // camCenter = pose_f2g.inv() * (0,0,0), se3transform.h:89-113 — the float expressions of projmatch.hip's match_enqueue
    const float m0 = T[0], m1 = T[4], m2 = T[8], m4 = T[1], m5 = T[5], m6 = T[9], m8 = T[2], m9 = T[6], m10 = T[10];
    const float m3 = -(T[3] * m0 + T[7] * m1 + T[11] * m2), m7 = -(T[3] * m4 + T[7] * m5 + T[11] * m6), m11 = -(T[3] * m8 + T[7] * m9 + T[11] * m10);
    d[12] = m0 * 0.f + m1 * 0.f + m2 * 0.f + m3;
    d[13] = m4 * 0.f + m5 * 0.f + m6 * 0.f + m7;
    d[14] = m8 * 0.f + m9 * 0.f + m10 * 0.f + m11;
    d[15] = tracked ? A.dec.r_tracked : A.dec.r_lost;
    d[16] = 0.f;
    for (int i = 0; i < 16; i++) A.dec.pose_map[i] = T[i];
    *A.dec.tracked = tracked;
}

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
__device__ __forceinline__ void p_set_Rt(PoseD& T) { p_quat_to_R(T.q, T.Rt); T.Rt[9] = T.t[0]; T.Rt[10] = T.t[1]; T.Rt[11] = T.t[2]; }
// ---- fp64 primitives of the serial path: v_rcp_f64 / v_rsq_f64 + two Newton steps (<= 1 ulp) instead of the IEEE division /
// square-root expansions (a dozen instructions each, on a path where every instruction is latency)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__device__ __forceinline__ double rsq_nr(double x) {   // x > 0
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = fma(y, fma(-h * y, y, 0.5), y);
    y = fma(y, fma(-h * y, y, 0.5), y);
    return y;
}

// Hu = the 21 upper-triangle entries row by row (H[a][c], a <= c, at index tri(a, c))
__device__ __forceinline__ constexpr int tri(int a, int c) { return a * 6 - a * (a - 1) / 2 + (c - a); }
__device__ __forceinline__ bool p_solve6(const double* Hu, const double* b, double lam, double* x) {   // LDL^T, fails on a zero / non-finite pivot
    // Right-looking (every pivot's column updates the trailing block at once) with the right-hand side carried along, then a
    // column-oriented back substitution: the dependent chain per pivot is reciprocal -> scale -> one fma instead of a dot product
    // of growing length — this routine sits on the serial path of every trial.  Fully unrolled, compile-time indices: the system
    // stays in registers.  x is left alone on failure.
    double a[6][6], L[6][6], id[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        y[i] = b[i];
#pragma unroll
        for (int j = 0; j <= i; j++) a[i][j] = Hu[tri(j, i)] + (i == j ? lam : 0.0);
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double dj = a[j][j];
        ok = ok && !(dj == 0.0 || !isfinite(dj));
        id[j] = rcp_nr(dj);
#pragma unroll
        for (int i = j + 1; i < 6; i++) L[i][j] = a[i][j] * id[j];
#pragma unroll
        for (int k = j + 1; k < 6; k++)
#pragma unroll
            for (int i = k; i < 6; i++) a[i][k] = fma(-L[i][j], a[k][j], a[i][k]);
#pragma unroll
        for (int i = j + 1; i < 6; i++) y[i] = fma(-L[i][j], y[j], y[i]);
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= id[i];
#pragma unroll
    for (int j = 5; j >= 0; j--) {
#pragma unroll
        for (int i = 0; i < j; i++) y[i] = fma(-L[j][i], y[j], y[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = y[i];
    return true;
}

// T <- exp(d) * T on the 3x4 matrix (g2o: SE3Quat::exp(update) * estimate, se3quat.h:276-311 — the same Rodrigues / V matrices; g2o
// goes through a unit quaternion after every product, which changes the result by rounding only).  For |w| < 0.5 the three
// coefficients sin(t)/t, (1 - cos t)/t^2, (t - sin t)/t^3 come from their power series in t^2 (nine terms: truncation < 1e-19):
// no square root, no division, no sincos on the path of every trial.
__device__ __forceinline__ void p_oplus_rt(double (&Rt)[12], const double* d) {
    const double w0 = d[0], w1 = d[1], w2 = d[2];
    const double z = w0 * w0 + w1 * w1 + w2 * w2;   // theta^2
    double a, b, c2;
    if (z < 1e-10) { a = 1; b = 0.5; c2 = 1.0 / 6.0; }      // theta < 0.00001 (se3quat.h:290)
    else if (z < 0.25) {
        // 1/(2k+1)!, 1/(2k+2)!, 1/(2k+3)! with alternating signs, Horner in z
        a = 1.0 / 355687428096000.0; b = 1.0 / 6402373705728000.0; c2 = 1.0 / 121645100408832000.0;
        a = fma(a, z, -1.0 / 1307674368000.0); b = fma(b, z, -1.0 / 20922789888000.0); c2 = fma(c2, z, -1.0 / 355687428096000.0);
        a = fma(a, z, 1.0 / 6227020800.0);     b = fma(b, z, 1.0 / 87178291200.0);     c2 = fma(c2, z, 1.0 / 1307674368000.0);
        a = fma(a, z, -1.0 / 39916800.0);      b = fma(b, z, -1.0 / 479001600.0);      c2 = fma(c2, z, -1.0 / 6227020800.0);
        a = fma(a, z, 1.0 / 362880.0);         b = fma(b, z, 1.0 / 3628800.0);         c2 = fma(c2, z, 1.0 / 39916800.0);
        a = fma(a, z, -1.0 / 5040.0);          b = fma(b, z, -1.0 / 40320.0);          c2 = fma(c2, z, -1.0 / 362880.0);
        a = fma(a, z, 1.0 / 120.0);            b = fma(b, z, 1.0 / 720.0);             c2 = fma(c2, z, 1.0 / 5040.0);
        a = fma(a, z, -1.0 / 6.0);             b = fma(b, z, -1.0 / 24.0);             c2 = fma(c2, z, -1.0 / 120.0);
        a = fma(a, z, 1.0);                    b = fma(b, z, 0.5);                     c2 = fma(c2, z, 1.0 / 6.0);
    } else {
        const double theta = sqrt(z);
        double sn, cs;
        sincos(theta, &sn, &cs);
        a = sn / theta; b = (1 - cs) / z; c2 = (theta - sn) / (z * theta);
    }
    // O = [w]x, O2 = O*O = w w^T - z I;  Rm = I + a O + b O2,  V = I + b O + c2 O2
    const double O[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    const double O2[9] = {w0 * w0 - z, w0 * w1, w0 * w2, w0 * w1, w1 * w1 - z, w1 * w2, w0 * w2, w1 * w2, w2 * w2 - z};
    double Rm[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        Rm[i] = fma(b, O2[i], fma(a, O[i], I));
        V[i] = fma(c2, O2[i], fma(b, O[i], I));
    }
    double n[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) n[r * 3 + c] = fma(Rm[r * 3 + 2], Rt[6 + c], fma(Rm[r * 3 + 1], Rt[3 + c], Rm[r * 3] * Rt[c]));
        const double rt = fma(Rm[r * 3 + 2], Rt[11], fma(Rm[r * 3 + 1], Rt[10], Rm[r * 3] * Rt[9]));
        const double vu = fma(V[r * 3 + 2], d[5], fma(V[r * 3 + 1], d[4], V[r * 3] * d[3]));
        n[9 + r] = rt + vu;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) Rt[i] = n[i];
}

// The matches live in LDS for the whole solve as 32-byte records {X, Y, Z, u, v, 1/sigma, weight, flags} (n <= kPnpLdsMatches:
// 96 KB): a pass reads a match with two ds_read_b128.  Larger n runs the same code on records packed into the caller's scratch.
constexpr int kPnpLdsMatches = 3000;
// 8 waves = two per SIMD of the one CU this kernel lives on: the per-match fp64 chains of one wave fill the issue gaps of the other.
// Wave 0 owns the Levenberg state and is alone on its SIMD during the serial steps (the other waves wait at the barrier).
constexpr int kPnpThreads = 512, kPnpWaves = kPnpThreads / 64;
constexpr int kNS = 29;               // sums per pass: H (21), b (6), robust chi2, inlier count
constexpr unsigned kActive = 1, kRobust = 2, kBad = 4;
enum : int { kModeEval = 0, kModeClassify = 1, kModeExit = 2, kModeLadder = 3 };
constexpr int kLadderMax = 8;         // candidates per ladder pass (one per wave; Levenberg's inner loop tries at most ten damping factors per iteration)

struct __attribute__((aligned(16))) MatchRec { float X, Y, Z, u, v, invsig, weight; unsigned flags; };
static_assert(sizeof(MatchRec) == 32, "match record");

template <bool CACHED>
__global__ __launch_bounds__(kPnpThreads) void pnp_solve_kernel(PnpArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_cache[];
    __shared__ __attribute__((aligned(16))) double s_part[kPnpWaves * 32];
    __shared__ __attribute__((aligned(16))) double s_tot[32];
    __shared__ __attribute__((aligned(16))) double s_pose[4][12];   // [0] pose to evaluate + accumulate, [1] pose of the excluded matches' fresh chi2 (end of the last
                                                                    // round), [2] pose the kept chi2 belong to (the last one evaluated), [3] the input pose
    __shared__ __attribute__((aligned(16))) double s_H[28];         // the normal equations of the current pose (21 + 6): wave 0's, parked here between solves
    __shared__ __attribute__((aligned(16))) double s_T[12], s_x[8]; // wave 0's: the current (last accepted) pose and the last solved step
    __shared__ int s_ctl[4];                                        // mode, classify, drop_robust, ladder length
    __shared__ __attribute__((aligned(16))) double s_lad[2];        // ladder request: lambda and ni of its first candidate
    __shared__ __attribute__((aligned(16))) double s_cand[kLadderMax][20];   // per candidate: trial pose (12), step (6), factorisation ok (1)
    const int tid = threadIdx.x, lane = tid & 63;
    int n = A.n;
    if (A.n_dev) {   // (uh_track_pose: the matches were chosen on the device)
        const int nd = __builtin_amdgcn_readfirstlane(*A.n_dev);
        n = nd < n ? nd : n;
        if (n <= 0) {   // pnpsolver.cpp:149-150: without matches the pose comes back as it went in
            if (tid < 16) A.pose_out[tid] = A.pose_in[tid];
            if (tid < 5) A.result[tid] = 0;
            if (A.dec.dyn17 && tid == 0) pnp_decide(A, 0, A.pose_in);
            return;
        }
    }
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave index as a scalar: the role split below is an s_cbranch
    if (A.clk && tid == 0) A.clk[0] = __builtin_readcyclecounter();
    const double fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];
    const double delta = (double)sqrtf(5.99f), dsqr = delta * delta;
    // one name for both instantiations; in each the pointer has a single provenance (LDS or HBM): ds_* or global_* accesses, never flat
    MatchRec* rec;
    if constexpr (CACHED) rec = reinterpret_cast<MatchRec*>(s_cache);
    else rec = reinterpret_cast<MatchRec*>(A.work);
    for (int e = tid; e < n; e += kPnpThreads) {
        MatchRec r;
        r.X = A.p3d[3 * e]; r.Y = A.p3d[3 * e + 1]; r.Z = A.p3d[3 * e + 2]; r.u = A.kp[2 * e]; r.v = A.kp[2 * e + 1];
        r.invsig = A.invsig[e]; r.weight = A.weight[e]; r.flags = kActive | kRobust;
        rec[e] = r;
    }
    {
        PoseD P;
        const float* M = A.pose_in;
        const double R0[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        p_quat_from_R(R0, P.q);       // g2o::SE3Quat(R, t): the rotation the optimisation starts from is the one of the NORMALISED quaternion
        p_quat_norm(P.q);
        P.t[0] = M[3]; P.t[1] = M[7]; P.t[2] = M[11];
        p_set_Rt(P);
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) { s_pose[3][i] = P.Rt[i]; s_pose[1][i] = P.Rt[i]; s_pose[2][i] = P.Rt[i]; }
        }
    }
    __syncthreads();
    if (A.clk && tid == 0) A.clk[1] = __builtin_readcyclecounter();

    // chi2 of a match at the pose Rt; xz, yz, invz for the Jacobian
    auto project = [&](const MatchRec& m, const double* Rt, double& ex, double& ey, double& xz, double& yz, double& invz) -> double {
        const double X0 = m.X, X1 = m.Y, X2 = m.Z;
        const double p0 = fma(Rt[2], X2, fma(Rt[1], X1, fma(Rt[0], X0, Rt[9])));
        const double p1 = fma(Rt[5], X2, fma(Rt[4], X1, fma(Rt[3], X0, Rt[10])));
        const double p2 = fma(Rt[8], X2, fma(Rt[7], X1, fma(Rt[6], X0, Rt[11])));
        invz = rcp_nr(p2);
        xz = p0 * invz; yz = p1 * invz;
        ex = (double)m.u - fma(xz, fx, cx);
        ey = (double)m.v - fma(yz, fy, cy);
        return (double)m.invsig * fma(ex, ex, ey * ey);
    };

    // One pass over this thread's matches (every wave).  classify: first the reclassification that ends a round
    // (pnpsolver.cpp:358-371): a match that was excluded gets a fresh chi2 at RtC, the others the chi2 of the LAST pose the optimiser
    // evaluated them at (RtK: g2o keeps the edge errors of its last trial, accepted or not) — recomputed here instead of stored, same
    // bits; chi2 > 5.99 -> outlier; drop_robust as the reference from its third round on.  Then (RtA != nullptr) every active match:
    // error, chi2, robust weight and Jacobian at RtA, accumulated into the 29 sums; the wave's totals go to s_part.
    auto pass = [&](bool accumulate, const double* RtA, bool classify, bool drop_robust) {
        double acc[kNS];
#pragma unroll
        for (int i = 0; i < kNS; i++) acc[i] = 0;
        for (int e = tid; e < n; e += kPnpThreads) {
            const MatchRec m = rec[e];
            unsigned f = m.flags;
            double ex, ey, xz, yz, invz;
            if (classify) {
                double Rs[12];
#pragma unroll
                for (int i = 0; i < 12; i++) Rs[i] = s_pose[(f & kBad) ? 1 : 2][i];   // (read per match: four classifying passes per solve)
                const double c = project(m, Rs, ex, ey, xz, yz, invz);
                const bool b = c > (double)5.99f;
                f = (b ? kBad : kActive) | (drop_robust ? 0u : (f & kRobust));
                rec[e].flags = f;
                acc[28] += b ? 0.0 : 1.0;
            }
            if (!(f & kActive) || !accumulate) continue;
            const double c = project(m, RtA, ex, ey, xz, yz, invz);
            const double w = m.invsig;
            double rho1 = 1.0, rc = c;
            if (f & kRobust) {
                const double wt = m.weight;
                if (c <= dsqr) rc = wt * c;
                else { const double rs = rsq_nr(c); rc = wt * fma(2 * (c * rs), delta, -dsqr); rho1 = delta * rs; }
            }
            acc[27] += rc;
            // 2x6 Jacobian rows (typesg2o.h:614-650): j = d ex / d xi, k = d ey / d xi; j[4] = k[3] = 0
            const double zf = invz * fx, zg = invz * fy, xf = xz * fx, yg = yz * fy;
            const double j0 = xf * yz, j1 = -fma(xf, xz, fx), j2 = yz * fx, j3 = -zf, j5 = xz * zf;
            const double k0 = fma(yg, yz, fy), k1 = -(xz * yg), k2 = -(xz * fy), k4 = -zg, k5 = yz * zg;
            const double s = rho1 * w;
            const double J[6] = {j0, j1, j2, j3, 0.0, j5}, K[6] = {k0, k1, k2, 0.0, k4, k5};
            const double sJ[6] = {s * j0, s * j1, s * j2, s * j3, 0.0, s * j5}, sK[6] = {s * k0, s * k1, s * k2, 0.0, s * k4, s * k5};
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int cc = a; cc < 6; cc++) {   // (the structural zeros are compile-time: their products are never formed)
                    double v = acc[tri(a, cc)];
                    if (a != 4 && cc != 4) v = fma(sJ[a], J[cc], v);
                    if (a != 3 && cc != 3) v = fma(sK[a], K[cc], v);
                    acc[tri(a, cc)] = v;
                }
#pragma unroll
            for (int a = 0; a < 6; a++) {
                double v = acc[21 + a];
                if (a != 4) v = fma(-sJ[a], ex, v);
                if (a != 3) v = fma(-sK[a], ey, v);
                acc[21 + a] = v;
            }
        }
        int off = 0, real = kNS;
        WaveTransposeValu<kNS, 32>::run(acc, lane, off, real);   // lane l ends with the wave total of one value index
        if (real >= 1) s_part[wv * 32 + off] = acc[0];
    };
    // what a pass does is published by wave 0 through s_ctl / s_pose before barrier A
    auto run_published_pass = [&]() -> int {
        const int mode = s_ctl[0];
        if (mode == kModeExit) return mode;
        double RtA[12];
#pragma unroll
        for (int i = 0; i < 12; i++) RtA[i] = s_pose[0][i];
        pass(mode != kModeClassify, RtA, s_ctl[1] != 0, s_ctl[2] != 0);
        return mode;
    };

    // ---- the damping ladder.  When a trial is rejected, Levenberg multiplies lambda by ni, doubles ni and tries again from the SAME
    // pose and normal equations, up to ten times — at convergence (rho < 0 by rounding noise) it walks the whole ladder, which as
    // dependent passes is most of a solve.  The candidates do not depend on each other's outcome, only the DECISION is sequential:
    // wave w solves and applies candidates w, w + 8 (every wave the same bits as the sequential loop would produce), one pass
    // evaluates the robust chi2 of all of them, wave 0 then walks the decisions in order and stops where the loop would have.
    auto solve_candidates = [&]() {
        const int K = s_ctl[3];
        for (int k = wv; k < K; k += kPnpWaves) {
            double lam = s_lad[0], nn = s_lad[1];
            for (int j = 0; j < k; j++) { lam *= nn; nn *= 2; }
            double Hu[21], b[6], Tt[12], xs[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 21; i++) Hu[i] = s_H[i];
#pragma unroll
            for (int i = 0; i < 6; i++) b[i] = s_H[21 + i];
#pragma unroll
            for (int i = 0; i < 12; i++) Tt[i] = s_pose[0][i];
            const bool ok = p_solve6(Hu, b, lam, xs);
            if (ok) p_oplus_rt(Tt, xs);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 12; i++) s_cand[k][i] = Tt[i];
#pragma unroll
                for (int i = 0; i < 6; i++) s_cand[k][12 + i] = xs[i];
                s_cand[k][18] = ok ? 1.0 : 0.0;
            }
        }
    };
    auto ladder_pass = [&]() {
        const int K = s_ctl[3];
        double chi[kLadderMax];
#pragma unroll
        for (int k = 0; k < kLadderMax; k++) {
            chi[k] = 0;
            if (k < K) {
                double Rt[12];
#pragma unroll
                for (int i = 0; i < 12; i++) Rt[i] = s_cand[k][i];
                for (int e = tid; e < n; e += kPnpThreads) {
                    const MatchRec m = rec[e];
                    if (!(m.flags & kActive)) continue;
                    double ex, ey, xz, yz, invz;
                    const double c = project(m, Rt, ex, ey, xz, yz, invz);
                    double rc = c;
                    if (m.flags & kRobust) {
                        const double wt = m.weight;
                        if (c <= dsqr) rc = wt * c;
                        else rc = wt * fma(2 * (c * rsq_nr(c)), delta, -dsqr);
                    }
                    chi[k] += rc;
                }
            }
        }
        int off = 0, real = kLadderMax;
        WaveTransposeValu<kLadderMax, 32>::run(chi, lane, off, real);
        if (real >= 1) s_part[wv * 32 + off] = chi[0];
    };

    if (wv != 0) {
        // ---- the seven worker waves: evaluate what wave 0 publishes until it says stop
        for (;;) {
            __syncthreads();                    // A: the request is in LDS
            if (s_ctl[0] == kModeLadder) {
                solve_candidates();
                __syncthreads();                // A2: the candidates' poses are in LDS
                ladder_pass();
            } else {
                const int mode = run_published_pass();
                if (mode == kModeExit) break;
            }
            __syncthreads();                    // B: the partial sums are in LDS
        }
    } else {
        // ---- wave 0: g2o's SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve as a state machine whose every
        // transition is ONE pass (or one damping-ladder pass) over the matches: a single call site for the pass keeps the code and
        // the register allocation of this wave small.  The control flow is the reference's, statement for statement:
        //   for round < 4 { T = T0; [classify previous round]; linearise;                         -> ST_INIT
        //     for it < 10 { swap(prev, cur); qmax = 0;
        //        do { solve; oplus; chi2 of the trial; accept / reject; } while (rho < 0 && ++qmax < 10)   -> ST_TRIAL, then ST_LADDER
        //        (a pose accepted inside a ladder pass is linearised by ST_RELIN before the next solve)
        //        stop on terminate or when the float chi2 no longer decreases }
        //   } classify the last round                                                               -> ST_FINAL
        enum : int { ST_INIT, ST_RELIN, ST_TRIAL, ST_LADDER, ST_FINAL, ST_DONE };
        int st = n > 0 ? ST_INIT : ST_DONE;
        // measurement (A.clk != NULL only): cycles of wave 0 by section of a plain pass — [16] request prepared (solve + update), [17] barrier A,
        // [18] its share of the matches + butterfly, [19] barrier B, [20] totals + decision; [21] ladder passes as a whole
        long long ph[6] = {0, 0, 0, 0, 0, 0}, tp = A.clk ? (long long)__builtin_readcyclecounter() : 0;
        auto stamp = [&](int i) { if (A.clk) { const long long t = (long long)__builtin_readcyclecounter(); ph[i] += t - tp; tp = t; } };
        int n_pass = 0, round = 0, it = 0, qmax = 0, done = 0, last_round = -1, good = n;
        bool stale_H = false, lam_finite = true, ok2 = false;
        double lambda = 0, ni = 2, currentChi = 0, lastChiRaw = 0, rho = 0, scale = 1;
        float prevChi = FLT_MAX, curChi = FLT_MAX;
        // (the pose and the last solved step live in LDS, s_T / s_x: nothing but scalars stays in registers across a pass.  A failed
        // factorisation leaves the step as it was — g2o's solution vector does the same)
        if (lane < 8) s_x[lane] = 0;

        auto iter_begin = [&]() {
            const float t = prevChi; prevChi = curChi; curChi = t;   // swap(prevChi2, curChi2) at loop entry
            qmax = 0; rho = 0; lam_finite = true;
            st = stale_H ? ST_RELIN : ST_TRIAL;
        };
        auto round_end = [&]() {
            if (lane < 12) s_pose[1][lane] = s_T[lane];
            if (lane == 0) A.result[1 + round] = done;
            last_round = round;
            ++round;
            st = round < 4 ? ST_INIT : ST_FINAL;
        };
        auto after_trials = [&]() {          // the do-while of Levenberg's solve() has just evaluated a trial
            if (lam_finite && rho < 0 && qmax < 10) { st = ST_LADDER; return; }
            done++;
            const bool terminate = (qmax == 10 || rho == 0 || !isfinite(lambda));
            curChi = (float)lastChiRaw;
            const float diff = prevChi - curChi;
            if (terminate || !(diff > 0.f) || it + 1 >= 10) round_end();
            else { ++it; iter_begin(); }
        };
        auto decide = [&](bool ok, double chi, const double* pose_lds) -> bool {   // one accept / reject decision (lambda, ni, currentChi, T)
            lastChiRaw = chi;
            const double tempChi = ok ? chi : DBL_MAX;
            const double r = (currentChi - tempChi) / scale;
            const bool acc = r > 0 && isfinite(tempChi);
            if (acc) {
                const double t3 = 2 * r - 1;
                double alpha = 1. - t3 * t3 * t3;
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                if (lane < 12) s_T[lane] = pose_lds[lane];
            } else {
                lambda *= ni; ni *= 2;                          // T and the normal equations stay those of the last accepted pose
                if (!isfinite(lambda)) lam_finite = false;
            }
            rho = r;
            return acc;
        };

        while (st != ST_DONE) {
            if (st == ST_LADDER) {
                // ---- rejected: the rest of the damping ladder in one pass
                const int K = 10 - qmax < kLadderMax ? 10 - qmax : kLadderMax;
                if (lane < 12) s_pose[0][lane] = s_T[lane];
                if (lane == 0) {
                    s_ctl[0] = kModeLadder; s_ctl[3] = K;
                    s_lad[0] = lambda; s_lad[1] = ni;
                }
                __syncthreads();                // A
                solve_candidates();
                __syncthreads();                // A2
                ladder_pass();
                __syncthreads();                // B
                ++n_pass;
                if (lane < kLadderMax) {
                    double mine = s_part[lane];
#pragma unroll
                    for (int w2 = 1; w2 < kPnpWaves; w2++) mine += s_part[w2 * 32 + lane];   // wave order
                    s_tot[lane] = mine;
                }
                double bb[6], x[6];
#pragma unroll
                for (int i = 0; i < 6; i++) { bb[i] = s_H[21 + i]; x[i] = s_x[i]; }
                int k_last = 0;
                for (int k = 0; k < K; k++) {   // the decisions, in the order the sequential loop takes them
                    const bool okk = s_cand[k][18] != 0.0;
                    if (okk) {
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] = s_cand[k][12 + i];
                    }
                    double sc = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) sc += x[i] * (lambda * x[i] + bb[i]);
                    scale = 1e-3 + sc;
                    k_last = k;
                    if (decide(okk, s_tot[k], s_cand[k])) stale_H = true;   // (a ladder pass does not linearise)
                    if (!lam_finite) break;   // the reference leaves its loop before the increment when lambda overflows
                    qmax++;
                    if (!(rho < 0 && qmax < 10)) break;
                }
                if (lane < 12) s_pose[2][lane] = s_cand[k_last][lane];   // the edges keep the errors of the last trial the loop evaluated
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 6; i++) s_x[i] = x[i];
                }
                after_trials();
                stamp(5);
                continue;
            }
            // ---- the request of this state
            int mode = kModeEval;
            bool classify = false, drop = false;
            if (st == ST_INIT) {
                if (lane < 12) { s_T[lane] = s_pose[3][lane]; s_pose[0][lane] = s_pose[3][lane]; }   // every round restarts from the input pose (:354)
                classify = round > 0; drop = round - 1 >= 2;
            } else if (st == ST_RELIN) {
                if (lane < 12) s_pose[0][lane] = s_T[lane];
            } else if (st == ST_TRIAL) {
                double Tt[12], x[6], Hu[21], b[6];
#pragma unroll
                for (int i = 0; i < 12; i++) Tt[i] = s_T[i];
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = s_x[i];
#pragma unroll
                for (int i = 0; i < 21; i++) Hu[i] = s_H[i];
#pragma unroll
                for (int i = 0; i < 6; i++) b[i] = s_H[21 + i];
                ok2 = p_solve6(Hu, b, lambda, x);
                if (ok2) p_oplus_rt(Tt, x);
                double sc = 0;
#pragma unroll
                for (int i = 0; i < 6; i++) sc += x[i] * (lambda * x[i] + b[i]);
                scale = 1e-3 + sc;
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 12; i++) s_pose[0][i] = Tt[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) s_x[i] = x[i];
                }
            } else if (st == ST_FINAL) {
                mode = kModeClassify; classify = true; drop = last_round >= 2;
            }
            if (lane == 0) { s_ctl[0] = mode; s_ctl[1] = classify ? 1 : 0; s_ctl[2] = drop ? 1 : 0; }
            stamp(0);
            __syncthreads();                    // A
            stamp(1);
            run_published_pass();
            stamp(2);
            __syncthreads();                    // B
            stamp(3);
            ++n_pass;
            if (lane < kNS) {
                double mine = s_part[lane];
#pragma unroll
                for (int w2 = 1; w2 < kPnpWaves; w2++) mine += s_part[w2 * 32 + lane];   // wave order
                s_tot[lane] = mine;
            }
            if (mode == kModeEval && lane < 12) s_pose[2][lane] = s_pose[0][lane];   // the pose the edges' errors now belong to
            // (the same wave wrote s_tot: LDS operations of one wave complete in order)
            const double chi_sum = s_tot[27], good_sum = s_tot[28];
            if (st == ST_INIT) {
                if (round > 0) {
                    good = (int)good_sum;
                    if (good < 10) { st = ST_DONE; continue; }
                }
                if (lane < 27) s_H[lane] = s_tot[lane];          // the totals of the pass are the normal equations of the pose it evaluated
                currentChi = chi_sum; lastChiRaw = chi_sum; ni = 2;
                {
                    double m = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) m = fmax(fabs(s_H[tri(j, j)]), m);
                    lambda = 1e-5 * m;
                }
                prevChi = FLT_MAX; curChi = FLT_MAX; done = 0; stale_H = false; it = 0;
                iter_begin();
            } else if (st == ST_RELIN) {
                if (lane < 27) s_H[lane] = s_tot[lane];
                stale_H = false;
                st = ST_TRIAL;
            } else if (st == ST_TRIAL) {
                if (decide(ok2, chi_sum, s_pose[0])) { if (lane < 27) s_H[lane] = s_tot[lane]; }   // the pass has linearised at the accepted pose already
                if (lam_finite) qmax++;   // (the reference leaves its loop before the increment when lambda overflows)
                after_trials();
            } else {   // ST_FINAL: the classification that ends the last round
                good = (int)good_sum;
                st = ST_DONE;
            }
            stamp(4);
        }
        if (A.clk && lane == 0) { for (int i = 0; i < 6; i++) A.clk[16 + i] = ph[i]; }
        if (A.clk && lane == 0) A.clk[2] = __builtin_readcyclecounter();
        if (lane == 0) {
            s_ctl[0] = kModeExit;
            for (int r = last_round + 1; r < 4; r++) A.result[1 + r] = 0;
            A.result[0] = n > 0 ? good : 0;
            double Tend[12];
#pragma unroll
            for (int i = 0; i < 12; i++) Tend[i] = s_pose[1][i];
            float* M = A.pose_out;
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = (float)Tend[r * 3 + c]; M[r * 4 + 3] = (float)Tend[9 + r]; }
            M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
            if (A.dec.dyn17) {
                float Mv[16];
                for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Mv[r * 4 + c] = (float)Tend[r * 3 + c]; Mv[r * 4 + 3] = (float)Tend[9 + r]; }
                Mv[12] = Mv[13] = Mv[14] = 0.f; Mv[15] = 1.f;
                pnp_decide(A, n > 0 ? good : 0, Mv);
            }
            if (A.state_out) {
                double q[4];
                p_quat_from_R(Tend, q);
                p_quat_norm(q);
                for (int i = 0; i < 4; i++) A.state_out[i] = q[i];
                for (int i = 0; i < 3; i++) A.state_out[4 + i] = Tend[9 + i];
            }
            if (A.clk) A.clk[4] = n_pass;
        }
        __syncthreads();                        // A of the exit request
    }
    for (int e = tid; e < n; e += kPnpThreads) A.bad_out[e] = (rec[e].flags & kBad) ? 1 : 0;
    if (A.host_done) {   // results went to pinned host memory: make them visible, then post the completion word
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(A.host_done, A.done_word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (A.clk && tid == 0) A.clk[3] = __builtin_readcyclecounter();
}

}  // namespace

struct uh_pnp {
    uh_ctx* ctx = nullptr;
    uh::DevBuf d_work;
    uh::MappedBuf h_io;      // pinned, device-visible: [inputs | results | completion word]
    unsigned long long seq = 0;
    bool attr_set = false;
    long long* d_clk = nullptr;   // measurement hook (uh_pnp_debug_clocks)
    ~uh_pnp() { if (d_clk) (void)hipFree(d_clk); }
};

namespace {

int launch(uh_pnp* p, PnpArgs& A) {
    const int n = A.n;
    A.clk = p->d_clk;
    if (n <= kPnpLdsMatches) {
        const size_t lds = (size_t)std::max(n, 1) * 32;
        if (!p->attr_set) {
            UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pnp_solve_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kPnpLdsMatches * 32));
            p->attr_set = true;
        }
        UH_LAUNCH(p->ctx, pnp_solve_kernel<true>, dim3(1), dim3(kPnpThreads), lds, A);
    } else {
        UH_LAUNCH(p->ctx, pnp_solve_kernel<false>, dim3(1), dim3(kPnpThreads), 0, A);
    }
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

}  // namespace

namespace uh {
uh_ctx* pnp_ctx(uh_pnp* p) { return p->ctx; }
// the solve behind uh_track_pose: everything resident, the match count decided by an earlier launch of the same stream
int pnp_enqueue_dev(uh_pnp* p, const float* d_pose, const float* d_intr4, int n_cap, const int* d_n, const float* d_p3d, const float* d_kp, const float* d_inv_sigma,
                    const float* d_weight, float* d_pose_out, unsigned char* d_bad_out, int* d_result5, const PnpDecide* dec) {
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    int rc;
    if (n_cap > kPnpLdsMatches && (rc = p->d_work.reserve((size_t)n_cap * 32))) return rc;
    PnpArgs A{};
    A.pose_in = d_pose; A.intr = d_intr4; A.n = n_cap; A.n_dev = d_n; A.p3d = d_p3d; A.kp = d_kp; A.invsig = d_inv_sigma; A.weight = d_weight;
    A.work = p->d_work.p;
    A.pose_out = d_pose_out; A.bad_out = d_bad_out; A.result = d_result5; A.state_out = nullptr;
    A.host_done = nullptr; A.done_word = 0;
    if (dec) A.dec = *dec;
    return launch(p, A);
}
}  // namespace uh

extern "C" {

int uh_pnp_create(uh_ctx* ctx, uh_pnp** out) {
    UH_REQUIRE(ctx && out, "uh_pnp_create: NULL argument");
    uh_pnp* p = new uh_pnp();
    p->ctx = ctx;
    *out = p;
    return UH_OK;
}
void uh_pnp_destroy(uh_pnp* p) { delete p; }

// Everything resident in HBM; asynchronous on the context stream.  d_work: n * 32 bytes of scratch (16-byte aligned), only used
// beyond kPnpLdsMatches matches.
int uh_pnp_solve_dev(uh_pnp* p, const float* d_pose_f2g, const float* d_intr4, int n, const float* d_p3d, const float* d_kp,
                     const float* d_inv_sigma, const float* d_weight, void* d_work, float* d_pose_out, uint8_t* d_bad_out,
                     int32_t* d_result5, double* d_state7) {
    UH_REQUIRE(p && d_pose_f2g && d_intr4 && d_pose_out && d_result5, "uh_pnp_solve_dev: NULL argument");
    UH_REQUIRE(n >= 0, "uh_pnp_solve_dev: negative match count");
    if (n > 0) UH_REQUIRE(d_p3d && d_kp && d_inv_sigma && d_weight && d_work && d_bad_out, "uh_pnp_solve_dev: NULL match arrays");
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    PnpArgs A{};
    A.pose_in = d_pose_f2g; A.intr = d_intr4; A.n = n; A.p3d = d_p3d; A.kp = d_kp; A.invsig = d_inv_sigma; A.weight = d_weight;
    A.work = d_work;
    A.pose_out = d_pose_out; A.bad_out = d_bad_out; A.result = d_result5; A.state_out = d_state7;
    A.host_done = nullptr; A.done_word = 0;
    return launch(p, A);
}

// Host-pointer form: PnPSolver::solvePnp(frame, map, matches, pose): returns the inlier count (>= 0) or a negative error.
// The caller's arrays are packed into the object's pinned staging block (a few KB), the kernel reads them from there and writes
// pose / flags / counters back into the same block; the host polls the completion word the kernel posts last.
int uh_pnp_solve(uh_pnp* p, const float* pose_f2g, const float* intr4, int n, const float* p3d, const float* kp, const float* inv_sigma,
                 const float* weight, float* pose_out, uint8_t* bad_out, int32_t* iters_out4, double* state_out7) {
    UH_REQUIRE(p && pose_f2g && intr4 && pose_out, "uh_pnp_solve: NULL argument");
    UH_REQUIRE(n >= 0, "uh_pnp_solve: negative match count");
    if (n == 0) { memcpy(pose_out, pose_f2g, 64); if (iters_out4) memset(iters_out4, 0, 16); return 0; }   // pnpsolver.cpp:149-150
    UH_REQUIRE(p3d && kp && inv_sigma && weight && bad_out, "uh_pnp_solve: NULL match arrays");
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    const size_t nf = (size_t)n;
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t o_done = 0, o_pout = 64, o_res = 128, o_state = 192, o_pose = 256, o_intr = 320, o_p3d = 384, o_kp = al(o_p3d + nf * 12), o_is = al(o_kp + nf * 8),
                 o_w = al(o_is + nf * 4), o_bad = al(o_w + nf * 4), total = al(o_bad + nf);
    int rc;
    if ((rc = p->h_io.reserve(total))) return rc;
    if (n > kPnpLdsMatches && (rc = p->d_work.reserve(nf * 32))) return rc;
    char* h = p->h_io.host<char>();
    char* d = p->h_io.dev<char>();
    memcpy(h + o_pose, pose_f2g, 64);
    memcpy(h + o_intr, intr4, 16);
    memcpy(h + o_p3d, p3d, nf * 12);
    memcpy(h + o_kp, kp, nf * 8);
    memcpy(h + o_is, inv_sigma, nf * 4);
    memcpy(h + o_w, weight, nf * 4);
    PnpArgs A{};
    A.pose_in = (const float*)(d + o_pose); A.intr = (const float*)(d + o_intr); A.n = n;
    A.p3d = (const float*)(d + o_p3d); A.kp = (const float*)(d + o_kp); A.invsig = (const float*)(d + o_is); A.weight = (const float*)(d + o_w);
    A.work = p->d_work.p;
    A.pose_out = (float*)(d + o_pout); A.bad_out = (unsigned char*)(d + o_bad); A.result = (int*)(d + o_res); A.state_out = (double*)(d + o_state);
    A.host_done = (unsigned long long*)(d + o_done);
    A.done_word = ++p->seq;
    std::atomic_thread_fence(std::memory_order_release);
    if ((rc = launch(p, A))) return rc;
    if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(h + o_done), A.done_word, p->ctx->stream, "uh_pnp_solve"))) return rc;
    int32_t res[5];
    memcpy(res, h + o_res, 20);
    memcpy(pose_out, h + o_pout, 64);
    memcpy(bad_out, h + o_bad, nf);
    if (state_out7) memcpy(state_out7, h + o_state, 56);
    if (iters_out4) memcpy(iters_out4, res + 1, 16);
    return res[0];
}

// measurement hook (scripts/time_pnp.py): shader-clock timestamps of the next solves — [0] kernel entry, [1] inputs staged,
// [2] rounds done, [3] results posted (s_memtime ticks), [4] number of passes.  on = 0 switches it off again.  out512: 512 entries
// (the whole 4096-byte stamp block is copied).
int uh_pnp_debug_clocks(uh_pnp* p, int on, long long* out512) {
    UH_REQUIRE(p, "uh_pnp_debug_clocks: NULL");
    UH_HIP_CHECK(hipSetDevice(p->ctx->device));
    if (on && !p->d_clk) { UH_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->d_clk), 4096)); UH_HIP_CHECK(hipMemset(p->d_clk, 0, 4096)); }
    if (out512 && p->d_clk) { UH_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); UH_HIP_CHECK(hipMemcpy(out512, p->d_clk, 4096, hipMemcpyDeviceToHost)); }
    if (!on && p->d_clk) { (void)hipFree(p->d_clk); p->d_clk = nullptr; }
    return UH_OK;
}

}  // extern "C"
