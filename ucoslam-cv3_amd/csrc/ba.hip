// Levenberg–Marquardt / Schur-complement bundle adjustment on MI355X (gfx950), fp64, behind the C ABI (uh_ba_*).
//
// Semantic contract = what GlobalOptimizerG2O does with monocular reprojection edges (reference file:line):
//   globaloptimizer_g2o.cpp:418-464  optimize(): optimize(nIters,1) -> edges with chi2>5.99 or depth<=0 to level 1, all
//                                     kernels dropped -> optimize(2*nIters,1)
//   globaloptimizer_g2o.cpp:466-537  getResults(): float poses/points + bad associations
//   typesg2o.h:249-323               EdgeSE3ProjectXYZ error/Jacobians;  :51-55,:76-79 oplus
//   g2o/core/sparse_optimizer.cpp:366-436                outer loop (stop when float chi2 drop <= minChi2BetweenIter)
//   g2o/core/optimization_algorithm_levenberg.cpp:58-175 LM trial loop, lambda init 1e-5*max|diag|, rho, lambda update
//   g2o/core/base_binary_edge.hpp:83-150                 Huber-weighted quadratic form (rho'' term dropped)
//   g2o/core/block_solver.hpp:315-447,525-566            lambda on ALL diagonals, Schur complement, back substitution
//
// MI355X design.  The problem is tiny for this chip (~26k edges, 3000 landmarks, <=64 free poses), so the enemy is latency,
// not bandwidth: the whole LM control flow lives in a device-resident state machine (BAState) and the host only enqueues a
// fixed sequence of "steps" (one LM trial each) — no host synchronisation inside a pass.  Every kernel starts by reading the
// state and returns immediately once the pass is done.  One step = TWO launches (+ lin in front of a pass, decide behind a
// round of steps); the state lives in two slots, step s reads slot s&1 and leaves the state it ran with in the other one:
//   lin     (first trial of a pass only)  8 lanes per landmark: errors, Huber weights, Hll, bl, per-edge Hpl blocks;
//                                          8 workgroups per free camera: Hpp, bp partials (deterministic tree reduction)
//   schur   prologue in EVERY workgroup: apply the accept/reject decision of the previous trial (rho, flip current<->trial
//           or lambda *= nu, iteration/termination logic) to the state — same sums, same code, same result everywhere;
//           workgroup 0 publishes it.  Then one workgroup per (camera i1 <= i2, landmark chunk): partial Hpl D^-1 Hpl^T
//           blocks of the reduced system; plus the camera workgroups (Hpp, bp) on every trial but the first
//   backsub prologue in EVERY workgroup (reduced system in LDS, n <= 126): S = Hpp + lambda I - sum of partials, blocked
//           look-ahead LDL^T, substitution, pose update T <- exp(dx) T into the trial buffer — 94 identical solves in
//           parallel instead of a one-workgroup launch.  Then 8 lanes per landmark: dx_l = D^-1 (b_l - Hpl^T dx_p), trial
//           point, trial errors, partial chi2 / scale sums, and speculatively the linearisation AT THE TRIAL estimate into
//           the trial half of the double-buffered Hll/bl/Hpl: accepting a trial flips estimate and linearisation together
//   solve   stand-alone one-workgroup form of the solve for systems that need the HBM workspace (126 < n <= 384)
//   decide  stand-alone form of the decision, closes a round of enqueued steps (one wave)
// (Folding solve and decide into the LAST-FINISHING workgroup of their producer launch — the threadfence-reduction pattern —
// was measured and rejected: the agent-scope fences write back / invalidate the per-XCD L2s once per workgroup and cost
// 15 us per step, five times the kernel boundary they save.  Redundant execution needs no fence.)
// All reductions run in a fixed order, so results are run-to-run deterministic.  MFMA is not used: the only dense algebra is
// 6x3·3x3·3x6 products per landmark pair (fp64) — far below any matrix-core tile; see DESIGN.md.
#include <cfloat>
#include <functional>
#include <cstdio>
#include <emmintrin.h>
#include <immintrin.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <atomic>

#include "common.hpp"
#include "reduce.hpp"

namespace {

constexpr int kMaxFree = 64;          // free (non-fixed) poses in the reduced system
constexpr int kThreads = 256;
constexpr int kSolveThreads = 256;                      // solve: one workgroup of 4 waves (one per SIMD)
constexpr int kHbmThreads = 512;                        // the stand-alone solve in HBM (33-64 free cameras): a thread per row of up to 385
constexpr int kFusedThreads = 512;                      // the chain's fused solve (17-21 free cameras): wave 0 the panels, seven waves the trailing update (wave 0 waited half the factorisation for three)
constexpr int kPackedThreads = 512;                     // its workgroup: 8 waves, two per SIMD — the trailing update's tiles are latency-bound on one
constexpr int kPackedFree = 32;                          // stand-alone solve on a packed triangle in LDS: up to 32 free cameras (n = 192: 148 KB)
constexpr int kMaxSplit = 12;                           // schur: landmark chunks per camera pair (partials the solve adds)
constexpr int kTrailU = 4;                              // solve: trailing-update elements in flight per thread
constexpr int kCamChunks = 8;                            // lin: workgroups per free camera
constexpr int kLanesPerPoint = 8;                       // lin / backsub: lanes cooperating on one landmark
constexpr int kPointsPerBlock = kThreads / kLanesPerPoint;

struct BAState {
    int phase;        // 0: next step linearises then runs a trial, 1: next step retries a trial (after a reject), 2: pass done
    int iteration;    // outer iteration index i of SparseOptimizer::optimize
    int max_iters;
    int qmax;         // trials done inside the current lm solve
    int cur;          // which of the two state buffers holds the current estimate
    int solve_ok;
    int iters_done;
    int first_trial;  // first trial of the pass: the lin kernel linearises; afterwards the backsub kernel does it speculatively
    int stopped;      // force-stop flag observed
    int pending;      // a trial has been computed whose accept/reject decision has not been applied to this state yet
    int stop_seen;    // force-stop flag as sampled by the backsub kernel of the pending trial
    int gate;         // set between the passes: pass 1 was complete (and not stopped) when the pre-enqueued pass 2 began
    double lambda, ni, currentChi, lastChiRaw;
    float prevChi2, curChi2, minChi2;
    int iters_pass1;  // outer iterations of pass 1 (iters_done is reset when pass 2 begins)
};
// Keep it small: measured on MI355X (A/B on the same box) a 96- or 104-byte state costs 0.79 ms per local BA, 112 bytes 1.05 ms
// and 128 bytes 1.22 ms — every one of the ~40 launches got ~6-10 us slower.  The state is read at the top of every kernel
// and copied through LDS in the schur prologue; past 104 bytes something in that path falls off a cliff (not investigated
// further).  gate / iters_pass1 therefore sit in what used to be padding.
static_assert(sizeof(BAState) == 96, "BAState layout");

struct BADims {
    int K, P, E, nfree, n;   // n = 6*nfree
    int nPointBlocks;        // ceil(P / kPointsPerBlock)
    double delta, dsqr, chi2_th;
};

struct BAPtrs {
    // problem (constant during optimize)
    const int* pt_ptr;        // P+1, CSR of edges per point (edge ids in input order)
    const int* pt_edges;      // E
    const int* cam_ptr;       // nfree+1, CSR of edges per free camera
    const int* cam_edges;     // edges of free cameras
    const int* e_pt; const int* e_kf;
    const double* e_uv; const double* e_w;
    const int* slot;          // K: reduced-system slot or -1
    const int* free_kf;       // nfree: frame index of slot s
    const double* intr;       // K x 4
    const int* edge_of;       // P x nfree: edge id of (point, slot) or -1
    // state (two buffers each)
    double* pose[2];          // K x 7 (qx qy qz qw tx ty tz)
    double* poseR[2];         // K x 12 (R row-major 9, t 3)
    double* pts[2];           // P x 3
    unsigned char* e_active; unsigned char* e_robust;
    double* e_err;            // E x 2
    double* e_chi2;           // E
    // system
    double* Hll[2]; double* bl[2];   // P x 9, P x 3: linearisation at the estimate in state buffer 0 / 1
    double* Hpl[2];           // E x 18 (6x3 row-major), per state buffer
    double* HppPart; double* bp;   // nfree x kCamChunks x 27 partial (21 upper Hpp entries + 6 of bp); summed bp nfree x 6
    double* S;                // (n x (n+1)) HBM workspace of the factorisation when it does not fit in LDS
    double* Spart;            // nsplit x npairs x 42: schur partials (6x6 block + 6-vector)
    double* dpart; double* dbpart;   // dense Schur form (ba_schur_dense_kernel): [G][tiles][256] product partials in MFMA C layout, [G][16 ntt] b_schur partials
    double* xp;               // n
    double* part_lin_chi;     // nPointBlocks
    double* part_maxdiag;     // nPointBlocks + nfree
    double* part_chi; double* part_scale;   // nPointBlocks
    BAState* st;              // two slots: step s reads slot s&1 and leaves the state it ran with in the other one
    const volatile unsigned char* stop;     // pinned host flag (may be NULL)
    long long* clk;           // 64 phase timestamps (100 MHz s_memrealtime) of the latest step, read by uh_ba_debug_clocks
};

#define UH_BA_CLK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) p.clk[i] = wall_clock64(); } while (0)
#define UH_BA_CLKL(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { p.clk[i] = wall_clock64(); p.clk[32 + i] = clock64(); } } while (0)   // single-workgroup kernels

// The BA launches are tiny and latency-bound, and they share the chip with the tracking stream's wide, VALU-saturating launches
// (8000 kNN waves, tens of thousands of FAST tiles).  Raising the issue priority of their waves (s_setprio) keeps those
// neighbours from taking eight ninths of a SIMD's issue slots away from a wave on the critical path.
__device__ __forceinline__ void uh_latency_critical() { __builtin_amdgcn_s_setprio(3); }

// ------------------------------------------------------------------------------------------------ small fp64 helpers
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void quat_from_R(const double* R, double* q) {   // Eigen::Quaternion(Matrix3)
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
__device__ __forceinline__ void quat_norm_pos(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void inv3(const double* M, double* I) {
    const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    const double id = 1.0 / (M[0] * c00 + M[1] * c01 + M[2] * c02);
    I[0] = c00 * id; I[1] = (M[2] * M[7] - M[1] * M[8]) * id; I[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    I[3] = c01 * id; I[4] = (M[0] * M[8] - M[2] * M[6]) * id; I[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    I[6] = c02 * id; I[7] = (M[1] * M[6] - M[0] * M[7]) * id; I[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// 1/d by v_rcp_f64 + two Newton steps (about 1 ulp): the pivots of the LDL^T sit on its longest dependent chain, where the
// IEEE division sequence (scale, rcp, 5 fma, fmas, fixup) costs twice as much.  Zero / non-finite d is tested by the caller.
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {   // lane index must be wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct EdgeLin {          // one edge's linearisation at a given state
    double ex, ey, chi2, rho1, ww, r0, r1;
    double A[6], B[12];
    double robchi;
};

// error + Huber weight (+ Jacobians) of edge e with camera pose (R,t) and point X
// value form (observation, information scalar and intrinsics passed in): shared by the legacy kernels and the persistent kernel
template <bool JAC>
__device__ __forceinline__ void edge_eval_v(double u, double v, double w, double fx, double fy, double cx, double cy, double delta, double dsqr,
                                            const double* Rt, const double* X, bool robust, EdgeLin& o) {
    const double x = Rt[0] * X[0] + Rt[1] * X[1] + Rt[2] * X[2] + Rt[9];
    const double y = Rt[3] * X[0] + Rt[4] * X[1] + Rt[5] * X[2] + Rt[10];
    const double z = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
    o.ex = u - ((x / z) * fx + cx);
    o.ey = v - ((y / z) * fy + cy);
    o.chi2 = w * (o.ex * o.ex + o.ey * o.ey);
    o.rho1 = 1.0;
    o.robchi = o.chi2;
    if (robust && o.chi2 > dsqr) {
        const double sq = sqrt(o.chi2);
        o.rho1 = delta / sq;
        o.robchi = 2 * sq * delta - dsqr;
    }
    if (JAC) {
        o.ww = o.rho1 * w;
        o.r0 = -w * o.ex * o.rho1;
        o.r1 = -w * o.ey * o.rho1;
        const double z2 = z * z;
        const double t02 = -x / z * fx, t12 = -y / z * fy, iz = -1. / z;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o.A[c] = iz * (fx * Rt[c] + t02 * Rt[6 + c]);
            o.A[3 + c] = iz * (fy * Rt[3 + c] + t12 * Rt[6 + c]);
        }
        o.B[0] = x * y / z2 * fx; o.B[1] = -(1 + (x * x / z2)) * fx; o.B[2] = y / z * fx; o.B[3] = -1. / z * fx; o.B[4] = 0; o.B[5] = x / z2 * fx;
        o.B[6] = (1 + y * y / z2) * fy; o.B[7] = -x * y / z2 * fy; o.B[8] = -x / z * fy; o.B[9] = 0; o.B[10] = -1. / z * fy; o.B[11] = y / z2 * fy;
    }
}

template <bool JAC>
__device__ __forceinline__ void edge_eval(const BAPtrs& p, const BADims& d, int e, int k, const double* Rt, const double* X,
                                          bool robust, EdgeLin& o) {
    edge_eval_v<JAC>(p.e_uv[2 * e], p.e_uv[2 * e + 1], p.e_w[e], p.intr[4 * k], p.intr[4 * k + 1], p.intr[4 * k + 2], p.intr[4 * k + 3],
                     d.delta, d.dsqr, Rt, X, robust, o);
}

// Linearisation of one landmark by its 8 lanes (one observation each per round; 32 landmarks per workgroup): errors, Huber
// weights, Hll, bl and the per-edge Hpl blocks at pose set poseR / point X, written to linearisation buffer `buf`.  The 8 partial sums are added with a fixed xor
// butterfly, so the result is deterministic and identical in all 8 lanes.  Lane gl==0 returns the landmark's robust chi2
// and max |diag Hll| (others 0).  Used by the lin kernel (current estimate) and by the backsub kernel (trial estimate).
__device__ __forceinline__ void linearize_point(const BAPtrs& p, const BADims& d, int buf, int pt, int gl, bool live,
                                                const double* poseR, const double* X, double& chi_part, double& maxd) {
    double acc[10];   // Hll upper (6), bl (3), robust chi2 (1)
#pragma unroll
    for (int i = 0; i < 10; i++) acc[i] = 0;
    bool any = false;
    if (live) {
        for (int i = p.pt_ptr[pt] + gl; i < p.pt_ptr[pt + 1]; i += kLanesPerPoint) {
            const int e = p.pt_edges[i];
            if (!p.e_active[e]) continue;
            any = true;
            const int k = p.e_kf[e];
            EdgeLin L;
            edge_eval<true>(p, d, e, k, poseR + 12 * k, X, p.e_robust[e] != 0, L);
            p.e_err[2 * e] = L.ex; p.e_err[2 * e + 1] = L.ey;
            p.e_chi2[e] = L.chi2;
            acc[9] += L.robchi;
            acc[0] += L.ww * (L.A[0] * L.A[0] + L.A[3] * L.A[3]); acc[1] += L.ww * (L.A[0] * L.A[1] + L.A[3] * L.A[4]);
            acc[2] += L.ww * (L.A[0] * L.A[2] + L.A[3] * L.A[5]); acc[3] += L.ww * (L.A[1] * L.A[1] + L.A[4] * L.A[4]);
            acc[4] += L.ww * (L.A[1] * L.A[2] + L.A[4] * L.A[5]); acc[5] += L.ww * (L.A[2] * L.A[2] + L.A[5] * L.A[5]);
            acc[6] += L.A[0] * L.r0 + L.A[3] * L.r1; acc[7] += L.A[1] * L.r0 + L.A[4] * L.r1; acc[8] += L.A[2] * L.r0 + L.A[5] * L.r1;
            if (p.slot[k] >= 0) {
                double* Hx = p.Hpl[buf] + 18 * (size_t)e;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 3; c++) Hx[a * 3 + c] = L.ww * (L.B[a] * L.A[c] + L.B[6 + a] * L.A[3 + c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; i++) {
#pragma unroll
        for (int o = kLanesPerPoint / 2; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
    }
    const unsigned long long anym = __ballot(any);
    const bool any_pt = ((anym >> ((threadIdx.x & 63) & ~(kLanesPerPoint - 1))) & 0xFFull) != 0;
    chi_part = 0; maxd = 0;
    if (live && gl == 0) {
        double* Hl = p.Hll[buf] + 9 * (size_t)pt;
        Hl[0] = acc[0]; Hl[1] = acc[1]; Hl[2] = acc[2]; Hl[3] = acc[1]; Hl[4] = acc[3]; Hl[5] = acc[4]; Hl[6] = acc[2]; Hl[7] = acc[4]; Hl[8] = acc[5];
        double* bo = p.bl[buf] + 3 * (size_t)pt;
        bo[0] = acc[6]; bo[1] = acc[7]; bo[2] = acc[8];
        chi_part = acc[9];
        if (any_pt) maxd = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
    }
}

// One (free camera, chunk of its observations) workgroup: partial Hpp (21 upper entries) and bp (6) at the CURRENT estimate;
// the chunks are added in order by the consumers (lambda init in the schur kernel, assembly in the solve kernel).  Runs in
// the lin launch when that kernel linearises, else as extra workgroups of the schur launch.
__device__ __forceinline__ void camera_block(const BAPtrs& p, const BADims& d, int cb, const double* poseR, const double* pts) {
    const int s = cb / kCamChunks, chunk = cb - s * kCamChunks;
    const int k = p.free_kf[s];
    double Rt[12];
#pragma unroll
    for (int i = 0; i < 12; i++) Rt[i] = poseR[12 * k + i];
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    for (int i = p.cam_ptr[s] + chunk * kThreads + threadIdx.x; i < p.cam_ptr[s + 1]; i += kCamChunks * kThreads) {
        const int e = p.cam_edges[i];
        if (!p.e_active[e]) continue;
        const int pt = p.e_pt[e];
        const double X[3] = {pts[3 * pt], pts[3 * pt + 1], pts[3 * pt + 2]};
        EdgeLin L;
        edge_eval<true>(p, d, e, k, Rt, X, p.e_robust[e] != 0, L);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = a; c < 6; c++) acc[q++] += L.ww * (L.B[a] * L.B[c] + L.B[6 + a] * L.B[6 + c]);
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += L.B[a] * L.r0 + L.B[6 + a] * L.r1;
    }
    __shared__ double s_part[4 * 27];
    __shared__ double s_out27[27];
    block_sum_vec<27>(acc, s_part, s_out27);
    if (threadIdx.x < 27) p.HppPart[(size_t)cb * 27 + threadIdx.x] = s_out27[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ lin
// grid = nPointBlocks + nfree*kCamChunks.  Launched once per pass (first trial); later linearisations come from backsub.
__global__ __launch_bounds__(kThreads) void ba_lin_kernel(BAPtrs p, BADims d, int slot) {
    uh_latency_critical();
    __shared__ double s_red[kThreads];
    const BAState st = p.st[slot];
    if (st.phase == 2 || !st.first_trial) return;
    UH_BA_CLK(0);
    const int cur = st.cur;
    const double* poseR = p.poseR[cur];
    const double* pts = p.pts[cur];
    if ((int)blockIdx.x < d.nPointBlocks) {
        const int gl = threadIdx.x & (kLanesPerPoint - 1);
        const int pt = blockIdx.x * kPointsPerBlock + (threadIdx.x >> 3);
        const bool live = pt < d.P;
        const int ptc = live ? pt : 0;
        const double X[3] = {pts[3 * ptc], pts[3 * ptc + 1], pts[3 * ptc + 2]};
        double chi_part, maxd;
        linearize_point(p, d, cur, pt, gl, live, poseR, X, chi_part, maxd);
        const double cs = block_sum(chi_part, s_red);
        const double mx = block_max(maxd, s_red);
        if (threadIdx.x == 0) { p.part_lin_chi[blockIdx.x] = cs; p.part_maxdiag[blockIdx.x] = mx; }
        UH_BA_CLK(1);
    } else {
        camera_block(p, d, blockIdx.x - d.nPointBlocks, poseR, pts);
    }
}

// ------------------------------------------------------------------------------------------------ decide
// One wave: the tail of OptimizationAlgorithmLevenberg::solve's do-while body plus SparseOptimizer::optimize's loop header.
__device__ __forceinline__ double wave_sum_fixed(double v) {   // xor butterfly: fixed order, identical in all lanes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct DecideSums { double lin, chi, scale, xs; };

// The four sums a decision needs, gathered by ONE wave (all 64 lanes call this): each lane pre-adds its strided share in a
// fixed order, the 64 lane sums are added by a fixed butterfly -> identical in every lane and in every workgroup that runs it.
// All loads are issued before the first use so that the caller pays one memory latency.
__device__ __forceinline__ DecideSums decide_sums(const BAPtrs& p, const BADims& d, int lane, double lambda0) {
    const int nb = d.nPointBlocks;
    double l0 = 0, l1 = 0, l2 = 0;
    for (int i = lane; i < nb; i += 64) { l0 += p.part_lin_chi[i]; l1 += p.part_chi[i]; l2 += p.part_scale[i]; }
    double xv[6 * kMaxFree / 64], bv[6 * kMaxFree / 64];
#pragma unroll
    for (int k = 0; k < 6 * kMaxFree / 64; k++) {
        const int i = lane + 64 * k;
        xv[k] = i < d.n ? p.xp[i] : 0.0;
        bv[k] = i < d.n ? p.bp[i] : 0.0;
    }
    double xs = 0;
#pragma unroll
    for (int k = 0; k < 6 * kMaxFree / 64; k++) xs += xv[k] * (lambda0 * xv[k] + bv[k]);
    for (int i = lane + 6 * kMaxFree; i < d.n; i += 64) { const double x = p.xp[i]; xs += x * (lambda0 * x + p.bp[i]); }   // wide problems
    DecideSums r;
    r.lin = wave_sum_fixed(l0); r.chi = wave_sum_fixed(l1); r.scale = wave_sum_fixed(l2); r.xs = wave_sum_fixed(xs);
    return r;
}

// The tail of OptimizationAlgorithmLevenberg::solve's do-while body plus SparseOptimizer::optimize's loop header, as a pure
// function of the state the trial started from and the trial's sums.
__device__ __forceinline__ BAState apply_decision(const BAState& st0, const DecideSums& sm, bool stop_flag) {
    const double sum_lin = sm.lin, sum_chi = sm.chi, sum_scale = sm.scale, sum_xs = sm.xs;
    BAState st = st0;
    if (st.first_trial) st.currentChi = sum_lin;   // activeRobustChi2 at the pass's initial estimate (lin kernel); later the accepted tempChi
    double tempChi = sum_chi, scale = sum_scale;
    st.lastChiRaw = tempChi;
    if (st.solve_ok) scale += sum_xs;
    if (!st.solve_ok) tempChi = DBL_MAX;
    double rho = st.currentChi - tempChi;
    scale += 1e-3;
    rho /= scale;
    bool lambda_finite = true;
    if (rho > 0 && isfinite(tempChi)) {
        const double t3 = 2 * rho - 1;
        double alpha = 1. - t3 * t3 * t3;
        alpha = fmin(alpha, 2. / 3.);
        const double sf = fmax(1. / 3., alpha);
        st.lambda *= sf;
        st.ni = 2;
        st.currentChi = tempChi;
        st.cur ^= 1;   // discardTop: the trial buffers (estimate AND its linearisation, see backsub) become the current ones
    } else {
        st.lambda *= st.ni;
        st.ni *= 2;    // pop: the current buffers stay
        if (!isfinite(st.lambda)) lambda_finite = false;
    }
    const bool stop = stop_flag;
    if (stop) st.stopped = 1;
    bool again = false;
    if (lambda_finite) {
        st.qmax++;
        again = (rho < 0 && st.qmax < 10 && !stop);
    }
    if (again) {
        st.phase = 1;
    } else {
        const bool terminate = (st.qmax == 10 || rho == 0 || !lambda_finite);
        const bool ok = !terminate;
        // SparseOptimizer::optimize: curChi2 = activeRobustChi2() of the LAST computed errors; Chi2Diff = prev - cur (float)
        st.curChi2 = (float)st.lastChiRaw;
        const float diff = st.prevChi2 - st.curChi2;
        st.iters_done++;
        st.iteration++;
        const bool cont = st.iteration < st.max_iters && !stop && ok && diff > st.minChi2;
        if (cont) {
            const float t = st.prevChi2; st.prevChi2 = st.curChi2; st.curChi2 = t;   // swap at the next loop entry
            st.phase = 0;
            st.qmax = 0;
        } else {
            st.phase = 2;
        }
    }
    st.first_trial = 0;
    st.pending = 0;
    return st;
}

// ------------------------------------------------------------------------------------------------ schur
// grid = npairs * nsplit.  Block (pair, chunk) accumulates the landmarks pt = chunk*256 + tid (+ nsplit*256 ...) of the
// (i1 <= i2) block and writes a PARTIAL 6x6 (and, on diagonal pairs, a partial 6-vector); the solve kernel adds the
// partials in chunk order.  The first block also publishes lambda at iteration 0 (computeLambdaInit).
__global__ __launch_bounds__(kThreads) void ba_schur_kernel(BAPtrs p, BADims d, int nsplit, int slot) {
    uh_latency_critical();
    __shared__ double s_part[4 * 42];
    __shared__ double s_out[42];
    // workgroup role and, for pair workgroups, the first landmark's edge ids and activity flags: none of it depends on the LM
    // state, so these loads are in flight together with the state load instead of behind it (each dependent HBM round trip
    // is ~1 us of this kernel's ~8)
    const int npairs = d.nfree * (d.nfree + 1) / 2;
    const int npairblocks = (npairs > 0 ? npairs : 1) * nsplit;
    const bool cam_role = (int)blockIdx.x >= npairblocks;
    const int pair = blockIdx.x / nsplit, chunk = blockIdx.x - pair * nsplit;
    const bool have_pair = !cam_role && pair < npairs;   // structure-only BA: the single workgroup only publishes lambda
    int s1 = 0, rem = have_pair ? pair : 0;
    while (have_pair && rem >= d.nfree - s1) { rem -= d.nfree - s1; ++s1; }
    const int s2 = s1 + rem;
    const bool diag = s1 == s2;
    const int pt0 = chunk * kThreads + threadIdx.x;
    int e1n = -1, e2n = -1;
    if (have_pair && pt0 < d.P) {
        e1n = p.edge_of[(size_t)pt0 * d.nfree + s1];
        e2n = diag ? e1n : p.edge_of[(size_t)pt0 * d.nfree + s2];
    }
    bool actn = e1n >= 0 && e2n >= 0;
    if (actn) actn = (p.e_active[e1n] != 0) & (p.e_active[e2n] != 0);
    // The decision about the PREVIOUS trial is taken here, by every workgroup for itself (same inputs, same code, same
    // result): the state the previous step left in slot `slot` plus that trial's sums give the state this step runs with.
    // Workgroup 0 publishes it in the other slot, which this step's later kernels and the next step's schur kernel read.
    __shared__ BAState s_state;
    if (threadIdx.x < 64) {
        const BAState st0 = p.st[slot];
        const DecideSums sm = decide_sums(p, d, threadIdx.x, st0.lambda);
        if (threadIdx.x == 0) s_state = (st0.phase != 2 && st0.pending) ? apply_decision(st0, sm, st0.stop_seen != 0) : st0;
    }
    __syncthreads();
    BAState st = s_state;
    if (st.phase == 2) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st.pending = 0; p.st[slot ^ 1] = st; }
        return;
    }
    UH_BA_CLK(4);
    double lambda = st.lambda;
    if (st.iteration == 0 && st.qmax == 0) {   // tau * max |H_jj| over poses and landmarks
        double m = 0;
        for (int i = 0; i < d.nPointBlocks; i++) m = fmax(m, p.part_maxdiag[i]);
        for (int s = 0; s < d.nfree; s++) {
            int q = 0;
            for (int a = 0; a < 6; a++) {   // diagonal entries of the 21-entry upper triangle: q = 0, 6, 11, 15, 18, 20
                double v = 0;
                for (int c = 0; c < kCamChunks; c++) v += p.HppPart[((size_t)s * kCamChunks + c) * 27 + q];
                m = fmax(m, fabs(v));
                q += 6 - a;
            }
        }
        lambda = 1e-5 * m;
        st.lambda = lambda; st.ni = 2;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { BAState pub = st; pub.pending = 1; pub.stop_seen = 0; p.st[slot ^ 1] = pub; }
    if (cam_role) {   // camera workgroups: Hpp / bp partials (the lin kernel has them at the first trial)
        if (!st.first_trial) camera_block(p, d, blockIdx.x - npairblocks, p.poseR[st.cur], p.pts[st.cur]);
        return;
    }
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0;
    for (int pt = pt0; have_pair && pt < d.P; pt += nsplit * kThreads) {
        int e1 = e1n, e2 = e2n;
        bool act = actn;
        if (pt != pt0) {   // later rounds (P > nsplit*256 landmarks per pair)
            e1 = p.edge_of[(size_t)pt * d.nfree + s1];
            e2 = diag ? e1 : p.edge_of[(size_t)pt * d.nfree + s2];
            act = e1 >= 0 && e2 >= 0;
            if (act) act = (p.e_active[e1] != 0) & (p.e_active[e2] != 0);
        }
        if (!act) continue;
        double D[9], Di[9];
#pragma unroll
        for (int i = 0; i < 9; i++) D[i] = p.Hll[st.cur][9 * (size_t)pt + i];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, Di);
        double b1[18], b2[18];
        const double* B1 = p.Hpl[st.cur] + 18 * (size_t)e1;
        const double* B2 = p.Hpl[st.cur] + 18 * (size_t)e2;
#pragma unroll
        for (int i = 0; i < 18; i++) { b1[i] = B1[i]; b2[i] = B2[i]; }
        double l0 = 0, l1 = 0, l2 = 0;
        if (diag) { const double* bi = p.bl[st.cur] + 3 * (size_t)pt; l0 = bi[0]; l1 = bi[1]; l2 = bi[2]; }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double y0 = b1[a * 3] * Di[0] + b1[a * 3 + 1] * Di[3] + b1[a * 3 + 2] * Di[6];
            const double y1 = b1[a * 3] * Di[1] + b1[a * 3 + 1] * Di[4] + b1[a * 3 + 2] * Di[7];
            const double y2 = b1[a * 3] * Di[2] + b1[a * 3 + 1] * Di[5] + b1[a * 3 + 2] * Di[8];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[a * 6 + c] += y0 * b2[c * 3] + y1 * b2[c * 3 + 1] + y2 * b2[c * 3 + 2];
            acc[36 + a] += y0 * l0 + y1 * l1 + y2 * l2;   // B1 * (Dinv * bl), only meaningful on diagonal pairs
        }
    }
    block_sum_vec<42>(acc, s_part, s_out);
    if (have_pair && threadIdx.x < 42) p.Spart[((size_t)chunk * npairs + pair) * 42 + threadIdx.x] = s_out[threadIdx.x];
    UH_BA_CLK(5);
}

// ------------------------------------------------------------------------------------------------ schur, dense form (17-32 free cameras)
// The pair form above recomputes a landmark's D^-1 for every camera pair and re-reads both Hpl blocks per pair: 528 pairs x 3000
// landmarks x 360 B = 570 MB of L2 traffic per launch at 32 free cameras (106 us); local-BA windows are dense (a landmark is seen by
// most of the window), so the product is a dense SYRK.  Here workgroup g owns the landmarks [g * lpw, (g + 1) * lpw), 16 at a time:
//   * panel: one thread per (landmark, camera): Hll + lambda I = L L^T (3 x 3), Y = Hpl L^-T (6 x 3) into the LDS panel
//     Yt[3 l + k][6 c + a] (zeros where the camera does not see the landmark), z = L^-1 b_l; every Hpl block is read ONCE;
//   * product: S_g += Yt^T Yt as v_mfma_f64_16x16x4_f64 on the lower 16 x 16 tiles (K = 48 rows per chunk = 12 MFMAs per tile), the
//     tiles dealt to the four waves, accumulators resident for the whole launch; b_schur_g += Yt^T z by the first n threads;
//   * the G partials are written in MFMA C layout (2 KB per tile, fully coalesced) and summed IN WORKGROUP ORDER by
//     ba_schur_reduce_kernel into the pair layout the solve's assembly reads (as ONE chunk): run-to-run deterministic.
// The prologue (the previous trial's decision, lambda at iteration 0, the published state) and the camera workgroups are those of
// ba_schur_kernel.  Reference: g2o/core/block_solver.hpp:341-392 (Hschur -= Hpl D^-1 Hpl^T, b -= Hpl D^-1 b_l).
struct SchurDense { int G, cpw, ntt, ys, T, SP, nown; };   // landmark groups (= partials), chunks per group, tiles per dimension, LDS row stride, lower tiles; wide form: workgroups per group, tiles per wave

__device__ __forceinline__ double schur_rsqrt(double x) {   // v_rsq_f64 + two Newton steps; x <= 0 / NaN gives NaN / inf (the solve then reports failure)
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x, r * r, 1.5);
    r = r * fma(-0.5 * x, r * r, 1.5);
    return r;
}

// tile t = ti (ti + 1) / 2 + tj of the lower triangle, as compile-time functions: a wave's tile list is STATIC (wave w owns the tiles
// w, w + 4, ...), so an MFMA's operands are picked from a register array of the 12 column groups by constant indices — one LDS read
// per column group and k-step (12) instead of two per tile (40)
__host__ __device__ constexpr int schur_tile_row(int t) { int ti = 0, t0 = 0; while (t0 + ti + 1 <= t) { t0 += ti + 1; ++ti; } return ti; }
__host__ __device__ constexpr int schur_tile_col(int t) { int ti = 0, t0 = 0; while (t0 + ti + 1 <= t) { t0 += ti + 1; ++ti; } return t - t0; }

template <int WV, int NTT>
__device__ __forceinline__ void schur_dense_wave(const BAPtrs& p, const BADims& d, const SchurDense& sd, const BAState& st, double lambda, double* Yt, double* s_z) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = blockIdx.x - d.nfree * kCamChunks, n = d.n, YS = sd.ys, i16 = lane & 15, kq = lane >> 4;   // (the camera workgroups come first)
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    constexpr int TT = NTT * (NTT + 1) / 2, NOWN = (TT - WV + 3) / 4;   // this wave's tiles: WV, WV + 4, ... < TT — everything below is unrolled over them
    f64x4 acc[NOWN];
#pragma unroll
    for (int u = 0; u < NOWN; u++) acc[u] = f64x4{0, 0, 0, 0};
    double bacc = 0;
    const double* Hll = p.Hll[st.cur];
    const double* Hpl = p.Hpl[st.cur];
    const double* bl = p.bl[st.cur];
    const int nslot = 16 * d.nfree;
    for (int c = 0; c < sd.cpw; c++) {
        const int l0 = (g * sd.cpw + c) * 16;
        if (l0 >= d.P) break;   // (uniform)
        if (c > 0) __syncthreads();   // the previous chunk's product has read the panel
        // panel, two (landmark, camera) slots per thread with their loads issued together (the rounds are three dependent memory
        // round trips each: edge id -> activity flag -> the blocks)
        constexpr int RND = 2;
        int e[RND], l[RND], sc[RND], pt[RND]; bool act[RND], live[RND];
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const int sl = tid + r * kThreads;
            live[r] = sl < nslot;
            l[r] = live[r] ? sl / d.nfree : 0; sc[r] = live[r] ? sl - l[r] * d.nfree : 0; pt[r] = l0 + l[r];
            live[r] = live[r] && pt[r] < d.P;
            e[r] = live[r] ? p.edge_of[(size_t)pt[r] * d.nfree + sc[r]] : -1;
        }
#pragma unroll
        for (int r = 0; r < RND; r++) act[r] = e[r] >= 0 && p.e_active[e[r] >= 0 ? e[r] : 0] != 0;
        double D[RND][6], bb[RND][18], bi[RND][3];
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const double* Dp = Hll + 9 * (size_t)(live[r] ? pt[r] : 0);
            D[r][0] = Dp[0]; D[r][1] = Dp[3]; D[r][2] = Dp[6]; D[r][3] = Dp[4]; D[r][4] = Dp[7]; D[r][5] = Dp[8];
            const double* Bp = Hpl + 18 * (size_t)(act[r] ? e[r] : 0);
#pragma unroll
            for (int i = 0; i < 18; i++) bb[r][i] = Bp[i];
            const double* bp = bl + 3 * (size_t)(live[r] ? pt[r] : 0);
            bi[r][0] = bp[0]; bi[r][1] = bp[1]; bi[r][2] = bp[2];
        }
#pragma unroll
        for (int r = 0; r < RND; r++) {
            if (tid + r * kThreads >= nslot) continue;
            const double d00 = D[r][0] + lambda, d10 = D[r][1], d20 = D[r][2], d11 = D[r][3] + lambda, d21 = D[r][4], d22 = D[r][5] + lambda;
            const double r0 = schur_rsqrt(d00), l10 = d10 * r0, l20 = d20 * r0;
            const double r1 = schur_rsqrt(fma(-l10, l10, d11)), l21 = fma(-l20, l10, d21) * r1;
            const double r2 = schur_rsqrt(fma(-l21, l21, fma(-l20, l20, d22)));
            double* yo = Yt + 3 * l[r] * YS + 6 * sc[r];
#pragma unroll
            for (int a = 0; a < 6; a++) {   // Y_a L^T = B_a (zeros where the camera does not see the landmark)
                const double y0 = bb[r][3 * a] * r0, y1 = fma(-y0, l10, bb[r][3 * a + 1]) * r1, y2 = fma(-y1, l21, fma(-y0, l20, bb[r][3 * a + 2])) * r2;
                yo[a] = act[r] ? y0 : 0.0; yo[YS + a] = act[r] ? y1 : 0.0; yo[2 * YS + a] = act[r] ? y2 : 0.0;
            }
            if (sc[r] == 0) {   // z = L^-1 b_l
                const double z0 = bi[r][0] * r0, z1 = fma(-l10, z0, bi[r][1]) * r1, z2 = fma(-l21, z1, fma(-l20, z0, bi[r][2])) * r2;
                s_z[3 * l[r]] = live[r] ? z0 : 0.0; s_z[3 * l[r] + 1] = live[r] ? z1 : 0.0; s_z[3 * l[r] + 2] = live[r] ? z2 : 0.0;
            }
        }
        __syncthreads();
        if (c == 0 && g == 0 && tid == 0) p.clk[6] = wall_clock64();
        if (tid < n) {
#pragma unroll 8
            for (int k = 0; k < 48; k++) bacc = fma(Yt[k * YS + tid], s_z[k], bacc);
        }
        // product: the 12 column groups of a k-step in registers (the next step's are fetched behind this step's MFMAs)
        double cur[NTT], nxt[NTT];
        const double* row0 = Yt + kq * YS + i16;
#pragma unroll
        for (int cg = 0; cg < NTT; cg++) cur[cg] = row0[16 * cg];
        for (int k4 = 0; k4 < 12; k4++) {
            const double* rown = Yt + (4 * (k4 + 1 < 12 ? k4 + 1 : k4) + kq) * YS + i16;
#pragma unroll
            for (int cg = 0; cg < NTT; cg++) nxt[cg] = rown[16 * cg];
#pragma unroll
            for (int u = 0; u < NOWN; u++)
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[schur_tile_row(WV + 4 * u)], cur[schur_tile_col(WV + 4 * u)], acc[u], 0, 0, 0);
#pragma unroll
            for (int cg = 0; cg < NTT; cg++) cur[cg] = nxt[cg];
        }
    }
    if (g == 0 && tid == 0) p.clk[7] = wall_clock64();
    double* out = p.dpart + (size_t)g * TT * 256;
#pragma unroll
    for (int u = 0; u < NOWN; u++) {
        double* o = out + (size_t)(WV + 4 * u) * 256 + lane * 4;
        *reinterpret_cast<double2*>(o) = double2{acc[u][0], acc[u][1]};
        *reinterpret_cast<double2*>(o + 2) = double2{acc[u][2], acc[u][3]};
    }
    if (tid < n) p.dbpart[(size_t)g * 16 * sd.ntt + tid] = bacc;
}

__global__ __launch_bounds__(kThreads) void ba_schur_dense_kernel(BAPtrs p, BADims d, SchurDense sd, int slot) {
    uh_latency_critical();
    extern __shared__ __attribute__((aligned(16))) double s_dense[];   // Yt[48][ys], z[48]
    __shared__ BAState s_state;
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncam = d.nfree * kCamChunks;
    const bool cam_role = (int)blockIdx.x < ncam;   // (the camera workgroups go first: they are the shorter ones and were the launch's tail behind the product workgroups)
#ifdef UH_BA_DENSE_CLK   // whole-launch stamps (scripts/ba_chain_clocks.py): entry of workgroup 0, the last product / camera workgroup's end
    if ((int)blockIdx.x == ncam && tid == 0) { p.clk[15] = wall_clock64(); p.clk[16] = 0; p.clk[17] = 0; }
#define UH_DENSE_END(i) do { if (tid == 0) atomicMax((unsigned long long*)&p.clk[i], (unsigned long long)wall_clock64()); } while (0)
#else
#define UH_DENSE_END(i)
#endif
    if (tid < 64) {
        const BAState st0 = p.st[slot];
        const DecideSums sm = decide_sums(p, d, tid, st0.lambda);
        if (tid == 0) s_state = (st0.phase != 2 && st0.pending) ? apply_decision(st0, sm, st0.stop_seen != 0) : st0;
    }
    double* const Yt = s_dense;
    double* const s_z = s_dense + 48 * sd.ys;
    if (!cam_role) for (int i = tid; i < 48 * sd.ys; i += kThreads) Yt[i] = 0.0;   // (the padding columns stay zero for the whole launch)
    __syncthreads();
    BAState st = s_state;
    if (st.phase == 2) {
        if (blockIdx.x == 0 && tid == 0) { st.pending = 0; p.st[slot ^ 1] = st; }
        return;
    }
    if ((int)blockIdx.x == ncam && tid == 0) p.clk[4] = wall_clock64();
    double lambda = st.lambda;
    if (st.iteration == 0 && st.qmax == 0) {   // tau * max |H_jj| over poses and landmarks (as in ba_schur_kernel)
        double m = 0;
        for (int i = 0; i < d.nPointBlocks; i++) m = fmax(m, p.part_maxdiag[i]);
        for (int s = 0; s < d.nfree; s++) {
            int q = 0;
            for (int a = 0; a < 6; a++) {
                double v = 0;
                for (int c = 0; c < kCamChunks; c++) v += p.HppPart[((size_t)s * kCamChunks + c) * 27 + q];
                m = fmax(m, fabs(v));
                q += 6 - a;
            }
        }
        lambda = 1e-5 * m;
        st.lambda = lambda; st.ni = 2;
    }
    if (blockIdx.x == 0 && tid == 0) { BAState pub = st; pub.pending = 1; pub.stop_seen = 0; p.st[slot ^ 1] = pub; }
    if (cam_role) {
        if (!st.first_trial) camera_block(p, d, blockIdx.x, p.poseR[st.cur], p.pts[st.cur]);
        UH_DENSE_END(17);
        return;
    }
    // each wave runs its own instantiation (the same barriers, a static tile list), one set per number of tile rows
#define UH_SCHUR_DENSE_NTT(N) case N: switch (wv) { \
        case 0: schur_dense_wave<0, N>(p, d, sd, st, lambda, Yt, s_z); break; case 1: schur_dense_wave<1, N>(p, d, sd, st, lambda, Yt, s_z); break; \
        case 2: schur_dense_wave<2, N>(p, d, sd, st, lambda, Yt, s_z); break; default: schur_dense_wave<3, N>(p, d, sd, st, lambda, Yt, s_z); break; } break;
    switch (sd.ntt) {
        UH_SCHUR_DENSE_NTT(7) UH_SCHUR_DENSE_NTT(8) UH_SCHUR_DENSE_NTT(9) UH_SCHUR_DENSE_NTT(10) UH_SCHUR_DENSE_NTT(11) UH_SCHUR_DENSE_NTT(12)
        default: break;   // (the host launches this kernel for 7..12 tile rows only: 17..32 free cameras)
    }
#undef UH_SCHUR_DENSE_NTT
    if ((int)blockIdx.x == ncam && tid == 0) p.clk[5] = wall_clock64();
    UH_DENSE_END(16);
}

// The dense form for 33-64 free cameras (13-24 tile rows, up to 300 lower tiles: their accumulators do not fit one workgroup).  SP
// workgroups share a landmark group; each builds the group's panels itself (8 landmarks = 24 rows at a time: the panel of 24 tile
// columns is 77 KB) and owns a contiguous range of the tile list, NOWN tiles per wave.  A wave's tile list is computed at run time
// (the operands' column offsets are wave-uniform scalars), every k-step reads its 2 NOWN operands from LDS and issues NOWN MFMAs —
// no predicate: a slot past the end of the list works on tile 0 and is not stored.
template <int NOWN>
__global__ __launch_bounds__(kThreads) void ba_schur_dense_wide_kernel(BAPtrs p, BADims d, SchurDense sd, int slot) {
    uh_latency_critical();
    extern __shared__ __attribute__((aligned(16))) double s_dense[];   // Yt[24][ys], z[24]
    __shared__ BAState s_state;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncam = d.nfree * kCamChunks;
    const bool cam_role = (int)blockIdx.x < ncam;
    if (tid < 64) {
        const BAState st0 = p.st[slot];
        const DecideSums sm = decide_sums(p, d, tid, st0.lambda);
        if (tid == 0) s_state = (st0.phase != 2 && st0.pending) ? apply_decision(st0, sm, st0.stop_seen != 0) : st0;
    }
    constexpr int ROWS = 24;
    double* const Yt = s_dense;
    double* const s_z = s_dense + ROWS * sd.ys;
    if (!cam_role) for (int i = tid; i < ROWS * sd.ys; i += kThreads) Yt[i] = 0.0;
    __syncthreads();
    BAState st = s_state;
    if (st.phase == 2) {
        if (blockIdx.x == 0 && tid == 0) { st.pending = 0; p.st[slot ^ 1] = st; }
        return;
    }
    if ((int)blockIdx.x == ncam && tid == 0) p.clk[4] = wall_clock64();
    double lambda = st.lambda;
    if (st.iteration == 0 && st.qmax == 0) {   // tau * max |H_jj| over poses and landmarks (as in ba_schur_kernel)
        double m = 0;
        for (int i = 0; i < d.nPointBlocks; i++) m = fmax(m, p.part_maxdiag[i]);
        for (int s = 0; s < d.nfree; s++) {
            int q = 0;
            for (int a = 0; a < 6; a++) {
                double v = 0;
                for (int c = 0; c < kCamChunks; c++) v += p.HppPart[((size_t)s * kCamChunks + c) * 27 + q];
                m = fmax(m, fabs(v));
                q += 6 - a;
            }
        }
        lambda = 1e-5 * m;
        st.lambda = lambda; st.ni = 2;
    }
    if (blockIdx.x == 0 && tid == 0) { BAState pub = st; pub.pending = 1; pub.stop_seen = 0; p.st[slot ^ 1] = pub; }
    if (cam_role) {
        if (!st.first_trial) camera_block(p, d, blockIdx.x, p.poseR[st.cur], p.pts[st.cur]);
        return;
    }
    const int wg = blockIdx.x - ncam, g = wg / sd.SP, sp = wg - g * sd.SP;
    const int n = d.n, YS = sd.ys, i16 = lane & 15, kq = lane >> 4;
    const int tps = (sd.T + sd.SP - 1) / sd.SP, tbeg = sp * tps, tend = min(sd.T, tbeg + tps);   // this workgroup's tiles
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    f64x4 acc[NOWN];
    int offA[NOWN], offB[NOWN], tix[NOWN];
#pragma unroll
    for (int u = 0; u < NOWN; u++) {
        acc[u] = f64x4{0, 0, 0, 0};
        const int t = tbeg + wv + 4 * u;
        tix[u] = t < tend ? t : -1;
        const int tc = t < tend ? t : 0;
        offA[u] = 16 * schur_tile_row(tc); offB[u] = 16 * schur_tile_col(tc);
    }
    double bacc[2] = {0, 0};
    const double* Hll = p.Hll[st.cur];
    const double* Hpl = p.Hpl[st.cur];
    const double* bl = p.bl[st.cur];
    const int nslot = 8 * d.nfree;   // (<= 512: two slots per thread)
    for (int c = 0; c < sd.cpw; c++) {
        const int l0 = (g * sd.cpw + c) * 8;
        if (l0 >= d.P) break;   // (uniform)
        if (c > 0) __syncthreads();
        constexpr int RND = 2;
        int e[RND], l[RND], sc[RND], pt[RND]; bool act[RND], live[RND];
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const int sl = tid + r * kThreads;
            live[r] = sl < nslot;
            l[r] = live[r] ? sl / d.nfree : 0; sc[r] = live[r] ? sl - l[r] * d.nfree : 0; pt[r] = l0 + l[r];
            live[r] = live[r] && pt[r] < d.P;
            e[r] = live[r] ? p.edge_of[(size_t)pt[r] * d.nfree + sc[r]] : -1;
        }
#pragma unroll
        for (int r = 0; r < RND; r++) act[r] = e[r] >= 0 && p.e_active[e[r] >= 0 ? e[r] : 0] != 0;
        double D[RND][6], bb[RND][18], bi[RND][3];
#pragma unroll
        for (int r = 0; r < RND; r++) {
            const double* Dp = Hll + 9 * (size_t)(live[r] ? pt[r] : 0);
            D[r][0] = Dp[0]; D[r][1] = Dp[3]; D[r][2] = Dp[6]; D[r][3] = Dp[4]; D[r][4] = Dp[7]; D[r][5] = Dp[8];
            const double* Bp = Hpl + 18 * (size_t)(act[r] ? e[r] : 0);
#pragma unroll
            for (int i = 0; i < 18; i++) bb[r][i] = Bp[i];
            const double* bp = bl + 3 * (size_t)(live[r] ? pt[r] : 0);
            bi[r][0] = bp[0]; bi[r][1] = bp[1]; bi[r][2] = bp[2];
        }
#pragma unroll
        for (int r = 0; r < RND; r++) {
            if (tid + r * kThreads >= nslot) continue;
            const double d00 = D[r][0] + lambda, d10 = D[r][1], d20 = D[r][2], d11 = D[r][3] + lambda, d21 = D[r][4], d22 = D[r][5] + lambda;
            const double r0 = schur_rsqrt(d00), l10 = d10 * r0, l20 = d20 * r0;
            const double r1 = schur_rsqrt(fma(-l10, l10, d11)), l21 = fma(-l20, l10, d21) * r1;
            const double r2 = schur_rsqrt(fma(-l21, l21, fma(-l20, l20, d22)));
            double* yo = Yt + 3 * l[r] * YS + 6 * sc[r];
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double y0 = bb[r][3 * a] * r0, y1 = fma(-y0, l10, bb[r][3 * a + 1]) * r1, y2 = fma(-y1, l21, fma(-y0, l20, bb[r][3 * a + 2])) * r2;
                yo[a] = act[r] ? y0 : 0.0; yo[YS + a] = act[r] ? y1 : 0.0; yo[2 * YS + a] = act[r] ? y2 : 0.0;
            }
            if (sc[r] == 0) {
                const double z0 = bi[r][0] * r0, z1 = fma(-l10, z0, bi[r][1]) * r1, z2 = fma(-l21, z1, fma(-l20, z0, bi[r][2])) * r2;
                s_z[3 * l[r]] = live[r] ? z0 : 0.0; s_z[3 * l[r] + 1] = live[r] ? z1 : 0.0; s_z[3 * l[r] + 2] = live[r] ? z2 : 0.0;
            }
        }
        __syncthreads();
        if (sp == 0) {   // (b_schur once per landmark group)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int col = tid + h * kThreads;
                if (col < n) {
#pragma unroll 8
                    for (int k = 0; k < ROWS; k++) bacc[h] = fma(Yt[k * YS + col], s_z[k], bacc[h]);
                }
            }
        }
        for (int k4 = 0; k4 < ROWS / 4; k4++) {
            const double* row = Yt + (4 * k4 + kq) * YS + i16;
            double a[NOWN], b[NOWN];
#pragma unroll
            for (int u = 0; u < NOWN; u++) { a[u] = row[offA[u]]; b[u] = row[offB[u]]; }
#pragma unroll
            for (int u = 0; u < NOWN; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc[u], 0, 0, 0);
        }
    }
    double* out = p.dpart + (size_t)g * sd.T * 256;
#pragma unroll
    for (int u = 0; u < NOWN; u++) {
        if (tix[u] >= 0) {
            double* o = out + (size_t)tix[u] * 256 + lane * 4;
            *reinterpret_cast<double2*>(o) = double2{acc[u][0], acc[u][1]};
            *reinterpret_cast<double2*>(o + 2) = double2{acc[u][2], acc[u][3]};
        }
    }
    if (sp == 0) {
#pragma unroll
        for (int h = 0; h < 2; h++) { const int col = tid + h * kThreads; if (col < n) p.dbpart[(size_t)g * 16 * sd.ntt + col] = bacc[h]; }
    }
    if ((int)blockIdx.x == ncam && tid == 0) p.clk[5] = wall_clock64();
}

// Sum of the G dense partials, in workgroup order, into the pair layout the solve's assembly reads (Spart as ONE chunk).  Workgroup t
// sums tile t: thread e reads word e of the tile in every partial (consecutive threads, consecutive words), finds its element (i, j)
// from the MFMA C layout (row = (lane >> 4) + 4 reg, col = lane & 15) and writes entry (a, c) of pair (j / 6, i / 6) — both orders
// inside a diagonal camera block.  Workgroup T sums b_schur.
constexpr int kReduceGroups = 4;   // the partials are cut into four consecutive ranges, summed by four thread groups and added in range order
// mode 0: into Spart (the pair layout, for a solve that assembles itself); mode 1 / 2: the FINISHED system into p.S — camera sums and
// lambda added on the diagonal blocks exactly as the assembly does ((-sum) + (h + lambda)), row n = b_p - b_schur, b_p stored for the
// decision — with row stride n + 1 (1: the fused solve) or as a packed lower triangle (2: the stand-alone solve): the solve copies it.
__global__ __launch_bounds__(kThreads* kReduceGroups) void ba_schur_reduce_kernel(BAPtrs p, BADims d, SchurDense sd, int slot, int mode) {
    uh_latency_critical();
    __shared__ double s_sum[kReduceGroups][kThreads];
    const BAState stp = p.st[slot];
    if (stp.phase == 2) return;   // (the pass had finished: the schur launch wrote nothing)
    const int tid = threadIdx.x & (kThreads - 1), grp = threadIdx.x / kThreads, t = blockIdx.x;
    const double* src; size_t stride;
    const bool bvec = t >= sd.T;                    // workgroups T, T + 1: b_schur, 256 columns each
    const int bcol = (t - sd.T) * kThreads + tid;
    if (bvec) { src = p.dbpart + (bcol < d.n ? bcol : 0); stride = (size_t)16 * sd.ntt; }
    else { src = p.dpart + (size_t)t * 256 + tid; stride = (size_t)sd.T * 256; }
    const int g0 = (int)((long long)sd.G * grp / kReduceGroups), g1 = (int)((long long)sd.G * (grp + 1) / kReduceGroups);
    double v = 0;
    int gidx = g0;
    for (; gidx + 16 <= g1; gidx += 16) {   // (a partial was written by another XCD a moment ago: every batch is a memory round trip)
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; u++) w[u] = src[(size_t)(gidx + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; u++) v += w[u];
    }
    {
        double w[16];
#pragma unroll
        for (int u = 0; u < 16; u++) w[u] = gidx + u < g1 ? src[(size_t)(gidx + u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) v += w[u];
    }
    s_sum[grp][tid] = v;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int k = 1; k < kReduceGroups; k++) v += s_sum[k][tid];
    auto pair_of = [&](int s1, int s2) { return s1 * d.nfree - s1 * (s1 - 1) / 2 + (s2 - s1); };
    const int n = d.n, ld = n + 1;
    auto SX = [&](int r, int c) -> size_t { return mode == 2 ? (size_t)r * (r + 1) / 2 + c : (size_t)r * ld + c; };
    if (bvec) {
        if (bcol >= n) return;
        const int sc = bcol / 6, a = bcol - 6 * sc;
        if (mode == 0) { p.Spart[(size_t)pair_of(sc, sc) * 42 + 36 + a] = v; return; }
        double h = 0;
#pragma unroll
        for (int cch = 0; cch < kCamChunks; cch++) h += p.HppPart[((size_t)sc * kCamChunks + cch) * 27 + 21 + a];
        p.bp[bcol] = h;
        p.S[SX(n, bcol)] = h - v;
        return;
    }
    const int ti = schur_tile_row(t), tj = schur_tile_col(t);
    const int lane = tid >> 2, reg = tid & 3;
    const int i = 16 * ti + (lane >> 4) + 4 * reg, j = 16 * tj + (lane & 15);
    if (i >= n || j > i) return;
    const int si = i / 6, ai = i - 6 * si, sj = j / 6, aj = j - 6 * sj;
    if (mode == 0) {
        double* o = p.Spart + (size_t)pair_of(sj, si) * 42;
        o[aj * 6 + ai] = v;
        if (si == sj && i != j) o[ai * 6 + aj] = v;
        return;
    }
    v = -v;
    if (si == sj) {
        const int lo = aj, hi = ai;   // (j <= i inside the block: aj <= ai)
        const int hq = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
        double h = 0;
#pragma unroll
        for (int cch = 0; cch < kCamChunks; cch++) h += p.HppPart[((size_t)si * kCamChunks + cch) * 27 + hq];
        v += h + (i == j ? stp.lambda : 0.0);
    }
    p.S[SX(i, j)] = v;
}

// The reduced system of the pair form, finished, into p.S (row stride n + 1, lower triangle, row n = b_p - b_schur; b_p stored for the
// decision) — for the solve in HBM (33-64 free cameras), which then needs no assembly of its own: one workgroup summing 2080 pairs x 42
// words was 185 us of a 64-camera trial.  Thread per (pair, q); the arithmetic and its order are those of the solve's assembly:
// sum of the chunks, negated, + (camera sums + lambda) on the diagonal blocks.
__global__ __launch_bounds__(kThreads) void ba_assemble_pairs_kernel(BAPtrs p, BADims d, int nsplit, int slot) {
    uh_latency_critical();
    const BAState st = p.st[slot];
    if (st.phase == 2) return;
    const int npairs = d.nfree * (d.nfree + 1) / 2;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= npairs * 42) return;
    const int pair = idx / 42, q = idx - 42 * pair;
    int s1 = 0, rem = pair;
    while (rem >= d.nfree - s1) { rem -= d.nfree - s1; ++s1; }
    const int s2 = s1 + rem;
    const bool diag = s1 == s2;
    if (q >= 36 && !diag) return;
    double v = 0;
    for (int k = 0; k < nsplit; k++) v += p.Spart[((size_t)k * npairs + pair) * 42 + q];
    const int n = d.n, ld = n + 1;
    if (q >= 36) {
        const int a = q - 36;
        double h = 0;
#pragma unroll
        for (int cch = 0; cch < kCamChunks; cch++) h += p.HppPart[((size_t)s1 * kCamChunks + cch) * 27 + 21 + a];
        p.bp[6 * s1 + a] = h;
        p.S[(size_t)n * ld + 6 * s1 + a] = h - v;
        return;
    }
    const int a = q / 6, c = q - 6 * a;
    v = -v;
    if (diag) {
        if (c > a) return;   // (the lower triangle of a diagonal block)
        const int lo = c, hi = a;
        const int hq = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
        double h = 0;
#pragma unroll
        for (int cch = 0; cch < kCamChunks; cch++) h += p.HppPart[((size_t)s1 * kCamChunks + cch) * 27 + hq];
        v += h + (a == c ? st.lambda : 0.0);
        p.S[(size_t)(6 * s1 + a) * ld + 6 * s1 + c] = v;
    } else {
        p.S[(size_t)(6 * s2 + c) * ld + 6 * s1 + a] = v;   // upper-block entry (6 s1 + a, 6 s2 + c), stored mirrored
    }
}

// Pose update of free pose `s` into the trial buffer: T_trial = exp(dx) * T_cur (SE3Quat::exp, VertexSE3Expmap::oplusImpl);
// with ok == 0 (the solve failed) the trial pose is the current one.  x: the solved increment (LDS or HBM).
// T <- exp(dx) * T on (unit quaternion q, translation t): SE3Quat::exp (se3quat.h:276) and VertexSE3Expmap::oplusImpl
__device__ __forceinline__ void se3_left_update(double (&q)[4], double (&t)[3], const double* dx) {
    const double w0 = dx[0], w1 = dx[1], w2 = dx[2];
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
    double qe[4];
    quat_from_R(Rm, qe);
    quat_norm_pos(qe);
    double te[3];
#pragma unroll
    for (int r = 0; r < 3; r++) te[r] = V[r * 3] * dx[3] + V[r * 3 + 1] * dx[4] + V[r * 3 + 2] * dx[5];
    double RE[9];
    quat_to_R(qe, RE);
    double qn[4];
    qn[3] = qe[3] * q[3] - qe[0] * q[0] - qe[1] * q[1] - qe[2] * q[2];
    qn[0] = qe[3] * q[0] + qe[0] * q[3] + qe[1] * q[2] - qe[2] * q[1];
    qn[1] = qe[3] * q[1] + qe[1] * q[3] + qe[2] * q[0] - qe[0] * q[2];
    qn[2] = qe[3] * q[2] + qe[2] * q[3] + qe[0] * q[1] - qe[1] * q[0];
    double tn[3];
#pragma unroll
    for (int r = 0; r < 3; r++) tn[r] = RE[r * 3] * t[0] + RE[r * 3 + 1] * t[1] + RE[r * 3 + 2] * t[2] + te[r];
    quat_norm_pos(qn);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = qn[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = tn[i];

}

__device__ __forceinline__ void pose_update_one(const BAPtrs& p, int s, int cur, int ok, const double* x) {
    const int trial = cur ^ 1;
    const int k = p.free_kf[s];
    const double* T = p.pose[cur] + 7 * k;
    double q[4] = {T[0], T[1], T[2], T[3]}, t[3] = {T[4], T[5], T[6]};
    if (ok) se3_left_update(q, t, x + 6 * s);
    double* To = p.pose[trial] + 7 * k;
    To[0] = q[0]; To[1] = q[1]; To[2] = q[2]; To[3] = q[3]; To[4] = t[0]; To[5] = t[1]; To[6] = t[2];
    double* Ro = p.poseR[trial] + 12 * k;
    quat_to_R(q, Ro);
    Ro[9] = t[0]; Ro[10] = t[1]; Ro[11] = t[2];
}

#ifndef UH_LDLT_LEGACY
#define UH_LDLT_LEGACY 0   // 1: keep the element-per-thread factorisation for every size (A/B in scripts/micro/ldlt_time.hip)
#endif
#ifndef UH_LDLT_CLK
#define UH_LDLT_CLK(i)   // scripts/micro/ldlt_time.hip stamps the phases of the factorisation through this hook
#endif
// (The row-per-lane factorisation for n + 1 <= 64 rows and its back substitution live in ldlt_rowlane_v2.hpp since round 5; rounds 3-4's
// ldlt_rowlane_lds — 6.4 us for eight free keyframes against 4.9 — is in the history.)
// The same factorisation for 64 < n + 1 <= 128 rows: TWO rows per lane of wave 0 (rows lane and lane + 64), the panel buffer
// s_y[2][6][128].  A pivot row lives in the low or the high register set of its lane — which one is uniform per column, so the
// v_readlane source is chosen by a scalar select.  Used by the persistent kernel's 9-16 free keyframe instantiation (n <= 96).
__device__ __forceinline__ double readlane2_f64(double lo, double hi, int row_uniform) {
    const double a = readlane_f64(lo, row_uniform & 63), b = readlane_f64(hi, row_uniform & 63);
    return row_uniform < 64 ? a : b;
}
__device__ __forceinline__ bool ldlt_rowlane2_lds(double* M, int n, int ld, int nfree, int npairs, const short (*s_pair)[2], double* s_y_raw) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = n / 6, nrow = n + 1;
    double* const s_y = s_y_raw + ((reinterpret_cast<uintptr_t>(s_y_raw) >> 3) & 1);   // [2][6][128], 16-byte aligned
    bool failed = false;
    double lp0[6] = {0, 0, 0, 0, 0, 0}, lp1[6] = {0, 0, 0, 0, 0, 0};
    auto wave0_step = [&](int kb) {
        const int k0 = 6 * kb;
        const int r0 = lane, r1 = lane + 64 < nrow ? lane + 64 : nrow - 1;   // (nrow > 64: every low row exists; idle high lanes shadow the last row)
        double a0[6], a1[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { a0[j] = M[r0 * ld + k0 + j]; a1[j] = M[r1 * ld + k0 + j]; }
        if (kb > 0) {
            const double* yb = s_y + ((kb - 1) & 1) * 768 + k0;
            double2 yv[6][3];
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int h = 0; h < 3; h++) yv[t][h] = *reinterpret_cast<const double2*>(yb + t * 128 + 2 * h);
#pragma unroll
            for (int t = 0; t < 6; t++)
                asm volatile("" : "+v"(yv[t][0].x), "+v"(yv[t][0].y), "+v"(yv[t][1].x), "+v"(yv[t][1].y), "+v"(yv[t][2].x), "+v"(yv[t][2].y));
#pragma unroll
            for (int t = 0; t < 6; t++) {
                a0[0] = fma(-lp0[t], yv[t][0].x, a0[0]); a0[1] = fma(-lp0[t], yv[t][0].y, a0[1]); a0[2] = fma(-lp0[t], yv[t][1].x, a0[2]);
                a0[3] = fma(-lp0[t], yv[t][1].y, a0[3]); a0[4] = fma(-lp0[t], yv[t][2].x, a0[4]); a0[5] = fma(-lp0[t], yv[t][2].y, a0[5]);
                a1[0] = fma(-lp1[t], yv[t][0].x, a1[0]); a1[1] = fma(-lp1[t], yv[t][0].y, a1[1]); a1[2] = fma(-lp1[t], yv[t][1].x, a1[2]);
                a1[3] = fma(-lp1[t], yv[t][1].y, a1[3]); a1[4] = fma(-lp1[t], yv[t][2].x, a1[4]); a1[5] = fma(-lp1[t], yv[t][2].y, a1[5]);
            }
        }
        asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]));
        asm volatile("" : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]), "+v"(a1[4]), "+v"(a1[5]));
        double dj[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            dj[j] = readlane2_f64(a0[j], a1[j], k0 + j);
            failed = failed || dj[j] == 0.0 || !isfinite(dj[j]);
            const double rr = __builtin_amdgcn_rcp(dj[j]);
            const double e = fma(-dj[j], rr, 1.0);
            const double ikj = fma(fma(e, e, e), rr, rr);
            lp0[j] = a0[j] * ikj; lp1[j] = a1[j] * ikj;
            if (j + 1 < 6) {
                const double y = readlane2_f64(a0[j], a1[j], k0 + j + 1);
                a0[j + 1] = fma(-(a0[j] * y), ikj, a0[j + 1]); a1[j + 1] = fma(-(a1[j] * y), ikj, a1[j + 1]);
            }
#pragma unroll
            for (int c = j + 2; c < 6; c++) {
                const double y = readlane2_f64(a0[j], a1[j], k0 + c);
                a0[c] = fma(-lp0[j], y, a0[c]); a1[c] = fma(-lp1[j], y, a1[c]);
            }
        }
        // every lane stores both rows (rows above the block land in the unused upper triangle; idle high lanes repeat the last row)
        double* yo = s_y + (kb & 1) * 768;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            M[r0 * ld + k0 + j] = lp0[j];
            M[r1 * ld + k0 + j] = lp1[j];
            yo[j * 128 + r0] = a0[j];
            yo[j * 128 + r1] = a1[j];
        }
    };
    // Block column kb of M from the diagonal upwards <- 0 (what wave 0's unmasked stores left there): the back substitution (backsolve2_lds)
    // then needs no mask — a lane whose row is not below the column multiplies by zero.  Done by waves 1..3 behind their tiles; with two
    // rows per lane on wave 0 they have the slack (one row per lane, n <= 63: they do not — there the factorisation got 0.9 us slower for
    // 0.56 us less back substitution, and ldlt_rowlane_lds / backsolve_lds keep the masked form).  D is not kept: nobody needs it.
    auto zero_upper = [&](int kb, int first, int step) {
        const int k0 = 6 * kb;
        for (int u = first; u < 6 * (k0 + 6); u += step) {
            const int r = u / 6, j = u - 6 * r;
            if (r <= k0 + j) M[r * ld + k0 + j] = 0.0;
        }
    };
    // Waves 1..3: the rank-6 update of panel kb on everything behind block column kb + 1 (rows c0 .. n, the right-hand-side row
    // included), as v_mfma_f64_16x16x4_f64 on 32 x 16 macro tiles with K padded from 6 to 8 (the lanes of k = 6, 7 feed zeros).  The
    // element-per-thread form it replaces (one (row, 6-column tile) per thread: 48 LDS operand reads per 36 FMAs) kept wave 0 waiting
    // at the barrier for half of a 17-camera factorisation (scripts/micro/ldlt_time.hip: 35 k of 70 k clocks).  The row-stride layout
    // has room above the diagonal (block column kb' is zeroed from the diagonal upwards by zero_upper(kb') AFTER every update that
    // reaches it), so a tile that straddles the diagonal simply updates those words too: only a tile past the last row or the last
    // column redirects its missing elements to a dump word (the unused word (0, n)).
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    auto trailing = [&](int kb) {
        const int k0 = 6 * kb, c0 = 6 * (kb + 2);
        if (c0 >= nrow) { zero_upper(kb, tid - 64, (int)blockDim.x - 64); return; }
        const double* yb = s_y + (kb & 1) * 768;
        const int w = __builtin_amdgcn_readfirstlane(wv) - 1, nwt = ((int)blockDim.x >> 6) - 1, i16 = lane & 15, kq = lane >> 4;
        const int TC = (n - c0 + 15) >> 4;   // tile columns (the border row has no column of its own)
        const bool k2 = kq < 2;   // this lane's second k-step exists (k = 4 + kq < 6)
        // a tile column's macro tiles are anchored at the LAST row (the top one may start above the diagonal — those words are the
        // free upper triangle — so none is ragged at the bottom: a ragged tile costs three times an interior one)
        int tj = 0, m0 = 0;
        for (int m = w;; m += nwt) {
            while (tj < TC && m0 + ((nrow - c0 - 16 * tj + 31) >> 5) <= m) { m0 += (nrow - c0 - 16 * tj + 31) >> 5; ++tj; }
            if (tj >= TC) break;
            const int cb = c0 + 16 * tj, rb = nrow - 32 * (m - m0 + 1), cc = cb + i16, rw0 = rb + kq;
            const double b0 = yb[kq * 128 + (cc < 128 ? cc : 127)], b1r = yb[(k2 ? 4 + kq : 0) * 128 + (cc < 128 ? cc : 127)];
            const double b1 = k2 ? b1r : 0.0;
            f64x4 acc0, acc1;
            if (rb >= 0 && cb + 16 <= ld) {   // (uniform) no ragged edge
                const int ab0 = (rb + i16) * ld + k0 + kq, ab1 = ab0 + 16 * ld;
                const double a00 = M[ab0], a01r = M[ab0 + (k2 ? 4 : 0)], a10 = M[ab1], a11r = M[ab1 + (k2 ? 4 : 0)];
                const int i0 = rw0 * ld + cc;
#pragma unroll
                for (int v = 0; v < 4; v++) { acc0[v] = M[i0 + 4 * v * ld]; acc1[v] = M[i0 + (16 + 4 * v) * ld]; }
                const double a01 = k2 ? -a01r : 0.0, a11 = k2 ? -a11r : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a00, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a10, b0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, b1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, b1, acc1, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) { M[i0 + 4 * v * ld] = acc0[v]; M[i0 + (16 + 4 * v) * ld] = acc1[v]; }
            } else {
                const int ar0 = rb + i16 > 0 ? rb + i16 : 0, ar1 = rb + 16 + i16 > 0 ? rb + 16 + i16 : 0;   // (a tile that starts before row 0 or ends past column n)
                const int ab0 = ar0 * ld + k0 + kq, ab1 = ar1 * ld + k0 + kq;
                const double a00 = M[ab0], a01r = M[ab0 + (k2 ? 4 : 0)], a10 = M[ab1], a11r = M[ab1 + (k2 ? 4 : 0)];
                int iv[8];
#pragma unroll
                for (int v = 0; v < 8; v++) iv[v] = (rw0 + 4 * v >= 0 && cc < ld) ? (rw0 + 4 * v) * ld + cc : n;   // (word (0, n): the dump)
#pragma unroll
                for (int v = 0; v < 4; v++) { acc0[v] = M[iv[v]]; acc1[v] = M[iv[4 + v]]; }
                const double a01 = k2 ? -a01r : 0.0, a11 = k2 ? -a11r : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a00, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a10, b0, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, b1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, b1, acc1, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; v++) { M[iv[v]] = acc0[v]; M[iv[4 + v]] = acc1[v]; }
            }
        }
        zero_upper(kb, tid - 64, (int)blockDim.x - 64);   // (after the tiles: they are what wave 0 waits for)
    };
    if (wv == 0 && nb > 0) wave0_step(0);
    for (int kb = 0; kb < nb; kb++) {
        UH_LDLT_CLK(100);   // (scripts/micro/ldlt_time.hip: wave 0's and wave 1's waits at the barrier tell which side the factorisation is bound by)
        __syncthreads();
        UH_LDLT_CLK(101);
        if (kb == nb - 1) break;
        if (wv == 0) wave0_step(kb + 1);
        else trailing(kb);
    }
    if (nb > 0) zero_upper(nb - 1, tid, (int)blockDim.x);
    if (wv != 0) failed = false;
    return failed;
}

// L^T x = z for 64 < n <= 128 (z = row n of M after the factorisation): wave 0, two unknowns per lane, one broadcast per column
__device__ __forceinline__ void backsolve2_lds(const double* M, int n, int ld, double* s_x) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (wv != 0) return;
    // (unmasked: ldlt_rowlane2_lds leaves zeros on and above the diagonal.  Two sweeps, so that the register a column's value is read from is fixed per sweep: columns
    // n-1 .. 64 in pairs (n is even), then 63 .. 0 in fours.)
    double x0 = M[(size_t)n * ld + lane], x1 = lane + 64 < n ? M[(size_t)n * ld + lane + 64] : 0.0;
    const int r1 = lane + 64 < n ? lane + 64 : n - 1;
    {
        double l0[2], l1[2], n0[2], n1[2];
#pragma unroll
        for (int t = 0; t < 2; t++) { const int jj = n - 1 - t; l0[t] = M[(size_t)jj * ld + lane]; l1[t] = M[(size_t)jj * ld + r1]; }
        for (int j0 = n - 1; j0 >= 64; j0 -= 2) {
#pragma unroll
            for (int t = 0; t < 2; t++) { const int jj = j0 - 2 - t >= 64 ? j0 - 2 - t : 64; n0[t] = M[(size_t)jj * ld + lane]; n1[t] = M[(size_t)jj * ld + r1]; }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const double xj = readlane_f64(x1, j0 - t - 64);   // (>= 0: n - 64 is even)
                x0 = fma(-l0[t], xj, x0);
                x1 = fma(-l1[t], xj, x1);
            }
#pragma unroll
            for (int t = 0; t < 2; t++) { l0[t] = n0[t]; l1[t] = n1[t]; }
        }
    }
    {
        double l0[4], n0[4];
#pragma unroll
        for (int t = 0; t < 4; t++) l0[t] = M[(size_t)(63 - t) * ld + lane];
        for (int j0 = 63; j0 >= 0; j0 -= 4) {
            if (j0 >= 4) {
#pragma unroll
                for (int t = 0; t < 4; t++) n0[t] = M[(size_t)(j0 - 4 - t) * ld + lane];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) x0 = fma(-l0[t], readlane_f64(x0, j0 - t), x0);
#pragma unroll
            for (int t = 0; t < 4; t++) l0[t] = n0[t];
        }
    }
    s_x[lane] = x0;
    if (lane + 64 < n) s_x[lane + 64] = x1;
}

#include "ldlt_rowlane_v2.hpp"
#include "ldlt_mfma.hpp"

// Blocked look-ahead LDL^T of the bordered system [S b; b^T .] held as a lower triangle in LDS (row stride ld = n + 1, odd), shared by the
// fused legacy solve and the persistent kernel.  Every thread of the workgroup calls it (it contains barriers); the first
// kSolveThreads threads do the work.  Returns (in wave 0) whether a zero / non-finite pivot was met.
__device__ __forceinline__ bool ldlt_bordered_lds(double* M, int n, int ld, int nfree, int npairs, const short (*s_pair)[2],
                                                  double (*s_w)[121][6]) {
    if (n + 1 <= 64 && !UH_LDLT_LEGACY) return ldlt_rowlane_v2<true>(M, n, ld, nfree, npairs, s_pair, &s_w[0][0][0]);   // (uniform)
    // 65..128 rows (11-21 free cameras): two rows per lane; the caller's panel buffer must hold 2 x 6 x 128 + 2 doubles (s_w[2][129][6])
    if (n + 1 <= 128 && !UH_LDLT_LEGACY) return ldlt_rowlane2_lds(M, n, ld, nfree, npairs, s_pair, &s_w[0][0][0]);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool active = tid < kSolveThreads;
    bool failed = false;
    {
        // LOOK-AHEAD form: panel kb is first applied to block column kb+1 alone, by all four waves; then wave 0 factorises that
        // diagonal block and solves ITS panel while waves 1..3 apply panel kb to the tiles behind — the serial chain of
        // reciprocals no longer waits for the bulk of the trailing update.  The L*D panel is double-buffered (s_w[kb & 1]).
        // The right-hand side rides along as row n of the matrix ([S b; b^T .] = bordered system): its L entries are
        // D^-1 L^-1 b, i.e. the forward substitution and the scaling by D^-1 come out of the panel solves and trailing
        // updates the factorisation performs anyway (one more row among idle lanes); only L^T x = z is left afterwards.
        const int nb = n / 6;
        const int nrow = n + 1;
        auto diag_and_panel = [&](int kb) {   // wave 0 only: block column kb is fully updated
            const int k0 = 6 * kb;
            double a[6][6], dk[6], ik[6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c <= i; c++) a[i][c] = M[(k0 + i) * ld + k0 + c];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                dk[j] = a[j][j];
                failed = failed || dk[j] == 0.0 || !isfinite(dk[j]);
                ik[j] = fast_rcp(dk[j]);
                double lcol[6];
#pragma unroll
                for (int i = j + 1; i < 6; i++) lcol[i] = a[i][j] * ik[j];
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int c = j + 1; c <= i; c++) a[i][c] = fma(-lcol[i], a[c][j], a[i][c]);
#pragma unroll
                for (int i = j + 1; i < 6; i++) a[i][j] = lcol[i];
            }
            UH_LDLT_CLK(3 + 4 * kb);
            __builtin_amdgcn_wave_barrier();   // every lane has read the block before lane 0 overwrites it
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
#pragma unroll
                    for (int c = 0; c < i; c++) M[(k0 + i) * ld + k0 + c] = a[i][c];
                    M[(k0 + i) * ld + k0 + i] = dk[i];
                }
            }
            for (int r = k0 + 6 + lane; r < nrow; r += 64) {   // panel: L_rk = A_rk * Lkk^-T * Dk^-1; y = L_rk * Dk goes to s_w
                double y[6];
#pragma unroll
                for (int j = 0; j < 6; j++) y[j] = M[r * ld + k0 + j];
#pragma unroll
                for (int j = 0; j < 6; j++) {
#pragma unroll
                    for (int t = 0; t < j; t++) y[j] = fma(-y[t], a[j][t], y[j]);
                }
#pragma unroll
                for (int j = 0; j < 6; j++) { M[r * ld + k0 + j] = y[j] * ik[j]; s_w[kb & 1][r][j] = y[j]; }
            }
            UH_LDLT_CLK(4 + 4 * kb);
        };
        UH_LDLT_CLK(0);
        if (wv == 0 && nb > 0) diag_and_panel(0);
        for (int kb = 0; kb < nb; kb++) {
            __syncthreads();   // panel kb (M columns of block kb, s_w[kb & 1]) is complete; trailing update kb-1 is done
            UH_LDLT_CLK(5 + 4 * kb);
            if (kb == nb - 1) break;
            const int k0 = 6 * kb;
            {   // block column kb+1 first, by everybody (one element per thread and round): wave 0 needs it to go on
                const int c0 = k0 + 6, m = nrow - c0;
                for (int e = tid; active && e < m * 6; e += kSolveThreads) {
                    const int r = c0 + e / 6, c = c0 + e % 6;
                    double acc = M[r * ld + c];
                    double lr[6], wc[6];
#pragma unroll
                    for (int t = 0; t < 6; t++) { lr[t] = M[r * ld + k0 + t]; wc[t] = s_w[kb & 1][c][t]; }
#pragma unroll
                    for (int t = 0; t < 6; t++) acc = fma(-lr[t], wc[t], acc);
                    if (c <= r) M[r * ld + c] = acc;
                }
            }
            __syncthreads();
            UH_LDLT_CLK(6 + 4 * kb);
            if (wv == 0) {
                diag_and_panel(kb + 1);
            } else if (active) {
                // tiles (s1 <= s2) with s1 >= kb+2: the tail of the s1-major pair list; 5 tile slots of 36 threads
                const int t3 = tid - 64;
                const int tq = t3 / 36, te = t3 - 36 * tq, ti = te / 6, tj = te - 6 * ti;
                constexpr int kSlots = (kSolveThreads - 64) / 36;
                // the threads left over by the tile slots update the right-hand-side row behind block column kb+1
                for (int c = 6 * (kb + 2) + (t3 - 36 * kSlots); t3 >= 36 * kSlots && c < n; c += kSolveThreads - 64 - 36 * kSlots) {
                    double acc = M[(size_t)n * ld + c];
#pragma unroll
                    for (int t = 0; t < 6; t++) acc = fma(-M[(size_t)n * ld + k0 + t], s_w[kb & 1][c][t], acc);
                    M[(size_t)n * ld + c] = acc;
                }
                const int s1min = kb + 2;
                const int tile0 = s1min * nfree - s1min * (s1min - 1) / 2;   // first pair with s1 >= kb+2
                const int ntile = npairs - tile0;
                for (int tl = tq; tl < ntile && tq < kSlots; tl += kTrailU * kSlots) {
                    int rr[kTrailU], cc[kTrailU]; bool on[kTrailU];
                    double lr[kTrailU][6], wc[kTrailU][6], acc[kTrailU];
#pragma unroll
                    for (int u = 0; u < kTrailU; u++) {
                        const int t_ = tl + u * kSlots;
                        const int tc = t_ < ntile ? t_ : tl;
                        const int s1 = s_pair[tile0 + tc][0], s2 = s_pair[tile0 + tc][1];
                        rr[u] = 6 * s2 + ti; cc[u] = 6 * s1 + tj;
                        on[u] = t_ < ntile && cc[u] <= rr[u];
#pragma unroll
                        for (int t = 0; t < 6; t++) { lr[u][t] = M[rr[u] * ld + k0 + t]; wc[u][t] = s_w[kb & 1][cc[u]][t]; }
                        acc[u] = M[rr[u] * ld + cc[u]];
                    }
#pragma unroll
                    for (int u = 0; u < kTrailU; u++) {
#pragma unroll
                        for (int t = 0; t < 6; t++) acc[u] = fma(-lr[u][t], wc[u][t], acc[u]);
                        if (on[u]) M[rr[u] * ld + cc[u]] = acc[u];
                    }
                }
            }
        }
        if (wv != 0) failed = false;   // only wave 0 sees the pivots
    }
    return failed;
}

// ------------------------------------------------------------------------------------------------ solve + pose update
// One workgroup.  Assembles S = Hpp + lambda I - sum of the schur partials (chunk order), factorises it as L D L^T (no
// pivoting; fails on a zero / non-finite pivot like Eigen's SimplicialLDLT) in LDS when it fits (n <= 126) else in HBM,
// substitutes, and writes T_trial = exp(dx) * T_cur for the free poses.  Row stride is n+1 doubles (odd) so that column
// walks are LDS-bank-conflict free.
struct SolveOut { bool done; int ok, cur; double lambda; };   // done: the pass had finished, nothing was computed

template <bool USE_LDS, bool PACKED = false>
__device__ __forceinline__ void solve_body(const BAPtrs& p, const BADims& d, int nsplit, int slot, double* s_x /* LDS, 6*kMaxFree */, SolveOut& out) {
    extern __shared__ __attribute__((aligned(16))) double s_mat[];
    __shared__ int s_ok;
    out.done = true;
    constexpr int NT = PACKED ? kPackedThreads : (USE_LDS ? kFusedThreads : kHbmThreads);   // threads of the workgroup that runs this body
    const int n = d.n, ld = n + 1;
    const int npairs = d.nfree * (d.nfree + 1) / 2;
    // the address space must be known at compile time: a generic pointer would turn every access into a flat_load
    auto M = [&]() { if constexpr (USE_LDS || PACKED) return s_mat; else return p.S; }();
    // PACKED (stand-alone solve, 127 < n + 1 <= 182): the lower triangle row by row in LDS, row r at r (r + 1) / 2 — a 30-camera system is
    // 131 KB that way and never leaves the CU; only entries (r, c <= r) are ever written (reads of the few (r, c > r) the trailing update
    // loads and discards land in the next row)
    auto IX = [&](int r, int c) -> size_t { if constexpr (PACKED) return (size_t)r * (r + 1) / 2 + c; else return (size_t)r * ld + c; };
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long clk_begin = wall_clock64();
    // assemble the lower triangle (+ diagonal) from the pair partials.  Four elements per thread and round, every partial
    // load of the four issued before the first add: the kernel is one workgroup, so exposed L2 latency is its whole cost
    // (PACKED serves at most kPackedFree cameras: its static LDS is sized for that, so that a 32-camera triangle — 148 KB — fits beside it)
    constexpr int kFreeCap = PACKED ? kPackedFree : kMaxFree;
    __shared__ short s_pair[kFreeCap * (kFreeCap + 1) / 2][2];
    if constexpr (!PACKED) {   // (the packed solve's factorisation does not walk camera pairs)
        for (int t = tid; t < npairs; t += NT) {
            int s1 = 0, rem = t;
            while (rem >= d.nfree - s1) { rem -= d.nfree - s1; ++s1; }
            s_pair[t][0] = (short)s1; s_pair[t][1] = (short)(s1 + rem);
        }
    }
    // (no barrier: the assembly decodes its pairs arithmetically; s_pair is first read after the barrier that ends it)
    // Two consecutive elements per 16-byte load (42 is even, the partial blocks are 16-byte aligned), three such units per
    // thread and round: with 8 free poses (36 pairs, 756 units) every partial of the reduced system is requested in ONE
    // round — the data was written by the previous launch on other XCDs, so a round costs a full HBM/MALL round trip.
    const int total = npairs * 21;
    // the LM state (lambda, current buffer, "pass finished") is requested together with the first round of partials
    double lambda = 0;
    int cur = 0;
    bool have_state = false;
    bool finished = false;
    // One pass over the units of the camera pairs, AU units per thread and round with MS partial loads each in flight.  DIAG: the
    // diagonal pairs, whose elements also take the camera-side sums (Hpp, bp: kCamChunks more loads per element); the other pairs'
    // units carry no such loads or registers, so many more of them fit a round: with 17 free cameras (153 pairs x 3 chunks) the
    // assembly was five rounds of 12 + 16 loads per unit (every unit issued kMaxSplit loads, the unused ones on chunk 0, and the
    // camera loads' registers); now two rounds of 3 loads for the 136 off-diagonal pairs and one for the 17 diagonal ones.
    auto pass = [&](auto au_c, auto ms_c, auto diag_c) {
        constexpr int AU = decltype(au_c)::value, MS = decltype(ms_c)::value;
        constexpr bool DIAG = decltype(diag_c)::value;
        const int units = DIAG ? d.nfree * 21 : total;
        for (int t0 = tid; t0 < units && !finished; t0 += NT * AU) {
            double2 xs[AU][MS];
            double hs[DIAG ? AU : 1][2][kCamChunks];
            int s1v[AU], s2v[AU], qv[AU];
#pragma unroll
            for (int u = 0; u < AU; u++) {
                const int t = t0 + u * NT;
                const int tc = t < units ? t : 0;
                int pair = tc / 21;
                const int q = 2 * (tc - pair * 21);
                int s1 = 0, s2 = 0;
                if (DIAG) { s1 = s2 = pair; pair = s1 * d.nfree - s1 * (s1 - 1) / 2; }
                else { int rem = pair; while (rem >= d.nfree - s1) { rem -= d.nfree - s1; ++s1; } s2 = s1 + rem; }
                s1v[u] = s1; s2v[u] = s2; qv[u] = (t < units && (DIAG || s1 != s2)) ? q : -1;
#pragma unroll
                for (int k = 0; k < MS; k++)
                    xs[u][k] = *reinterpret_cast<const double2*>(p.Spart + ((size_t)(k < nsplit ? k : 0) * npairs + pair) * 42 + q);
                if constexpr (DIAG) {   // camera-side entries that join these elements: Hpp upper-triangle index or bp component
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int qq = q + j;
                        int hq = 0;
                        if (qq >= 36) hq = 21 + (qq - 36);
                        else { const int a = qq / 6, c = qq - a * 6, lo = a < c ? a : c, hi = a < c ? c : a; hq = lo * 6 - lo * (lo - 1) / 2 + (hi - lo); }
#pragma unroll
                        for (int cch = 0; cch < kCamChunks; cch++) hs[u][j][cch] = p.HppPart[((size_t)s1 * kCamChunks + cch) * 27 + hq];
                    }
                }
            }
            const long long clk_issued = wall_clock64();
            if (!have_state) {
                const BAState st = p.st[slot];
                if (st.phase == 2) { finished = true; break; }   // uniform: nothing has been written yet
                lambda = st.lambda; cur = st.cur; have_state = true;
                if (tid == 0 && blockIdx.x == 0) { p.clk[10] = clk_begin; p.clk[26] = clk_issued; p.clk[27] = wall_clock64(); }
            }
#pragma unroll
            for (int u = 0; u < AU; u++) {
                const int s1 = s1v[u], s2 = s2v[u];
                if (qv[u] < 0) continue;
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int q = qv[u] + j;
                    double v = 0;
#pragma unroll
                    for (int k = 0; k < MS; k++) if (k < nsplit) v += j ? xs[u][k].y : xs[u][k].x;
                    double h = 0;
                    if constexpr (DIAG) {
#pragma unroll
                        for (int cch = 0; cch < kCamChunks; cch++) h += hs[u][j][cch];
                    }
                    if (q >= 36) {   // b_schur = b_p - sum_l Hpl Dinv b_l (diagonal pairs carry it); b_p = sum of the camera chunks
                        if (DIAG) {
                            p.bp[6 * s1 + (q - 36)] = h;             // the decide stage needs b_p for computeScale
                            s_x[6 * s1 + (q - 36)] = h - v;
                            if constexpr (USE_LDS) M[(size_t)n * ld + 6 * s1 + (q - 36)] = h - v;   // row n of the bordered matrix
                            if constexpr (PACKED) M[IX(n, 6 * s1 + (q - 36))] = h - v;              // (the packed triangle carries that row too)
                            if constexpr (!PACKED && !USE_LDS) M[(size_t)n * ld + 6 * s1 + (q - 36)] = h - v;   // (and so does the system in HBM: p.S has n + 1 rows)
                        }
                        continue;
                    }
                    const int a = q / 6, c = q - a * 6;
                    v = -v;
                    if (DIAG) v += h + (a == c ? lambda : 0.0);
                    const int r = 6 * s1 + a, cc = 6 * s2 + c;   // upper-block entry (r,cc); store it mirrored into the lower triangle
                    if (DIAG) { if (c <= a) M[IX(r, cc)] = v; }
                    else M[IX(cc, r)] = v;
                }
            }
        }
    };
    using std::integral_constant;
    using std::true_type; using std::false_type;
    if (nsplit == 0) {
        // The dense Schur form's reduce launch (ba_schur_reduce_kernel) left the finished system in p.S — partials summed, camera sums
        // and lambda folded in, right-hand side as row n, in THIS body's layout (packed, or row stride n + 1): a straight copy.
        if constexpr (USE_LDS || PACKED) {
            const size_t total = PACKED ? (size_t)(n + 1) * (n + 2) / 2 : (size_t)(n + 1) * ld;
            constexpr int CU = 6;   // 16-byte units per thread in flight
            for (size_t i0 = 2 * (size_t)tid; i0 < total; i0 += 2 * (size_t)NT * CU) {
                double2 w[CU];
#pragma unroll
                for (int u = 0; u < CU; u++) { const size_t i = i0 + 2 * (size_t)NT * u; w[u] = i + 1 < total ? *reinterpret_cast<const double2*>(p.S + i) : double2{i < total ? p.S[i] : 0.0, 0.0}; }
                if (!have_state) {
                    const BAState st = p.st[slot];
                    if (st.phase == 2) { finished = true; break; }   // uniform: nothing has been written yet
                    lambda = st.lambda; cur = st.cur; have_state = true;
                    if (tid == 0 && blockIdx.x == 0) { p.clk[10] = clk_begin; p.clk[26] = clk_begin; p.clk[27] = wall_clock64(); }
                }
#pragma unroll
                for (int u = 0; u < CU; u++) { const size_t i = i0 + 2 * (size_t)NT * u; if (i + 1 < total) *reinterpret_cast<double2*>(M + i) = w[u]; else if (i < total) M[i] = w[u].x; }
            }
        }
    } else if (nsplit <= 3) {
        pass(integral_constant<int, 2>{}, integral_constant<int, 3>{}, true_type{});     // (first: it requests the state; its lambda goes onto the diagonal)
        if (!finished) pass(integral_constant<int, 8>{}, integral_constant<int, 3>{}, false_type{});
    } else {
        pass(integral_constant<int, 2>{}, integral_constant<int, kMaxSplit>{}, true_type{});
        if (!finished) pass(integral_constant<int, 3>{}, integral_constant<int, kMaxSplit>{}, false_type{});
    }
    if (finished) return;
    UH_BA_CLK(28);
    if (!have_state) {   // no free pose: nothing was assembled
        const BAState st = p.st[slot];
        if (st.phase == 2) return;
        lambda = st.lambda; cur = st.cur;
    }
    if (tid == 0) s_ok = 1;
    __syncthreads();
    UH_BA_CLKL(11);
    // Right-looking LDL^T of the lower triangle, blocked by the 6x6 camera blocks (n = 6*nfree).  Per block column: an
    // in-register factorisation of the diagonal block (the dependent chain of six reciprocals), a panel solve (thread per
    // row) and a rank-6 trailing update (thread per element).  The phase is bound by the instruction count of one wave (fp64
    // ops issue at 4-8 cycles), hence explicit fma() everywhere, the panel kept twice (L in M, L*D in s_w: no multiply by d
    // in the update) and a loop-invariant thread -> (row, column) mapping.  After it M holds L (unit lower) and d on the
    // diagonal.
    bool failed = false;
    bool solved = false;   // the factorisation below substituted too: x is in s_x
    if constexpr (PACKED) {
        solved = true;
        // 22-32 free keyframes: panels by every thread, MFMA trailing update, block back substitution (ldlt_mfma.hpp); x lands in s_x
        double* const s_aux = s_mat + ((((size_t)(n + 1) * (n + 2) / 2) + 1) & ~(size_t)1);
        failed = ldlt_solve_mfma_lds<true>(M, n, ld, s_aux, s_x);
    } else if constexpr (USE_LDS) {
        __shared__ double s_w_store[2][129][6];   // ([2][121][6] for the look-ahead form; the two-rows-per-lane form's panel buffer is [2][6][128] + alignment)
        double (*s_w)[121][6] = reinterpret_cast<double (*)[121][6]>(&s_w_store[0][0][0]);
        failed = ldlt_bordered_lds(M, n, ld, d.nfree, npairs, s_pair, s_w);
    } else {
        // 33-64 free keyframes, the system in HBM (row stride n + 1, n + 1 <= 385 rows: a thread per row): the same factorisation on
        // global memory — a pair of block columns is six dependent memory steps; the 6-column form it replaces (diagonal block in,
        // factor, out; panel in, out; update; three barriers that wait for stores: 25 us per block column) took twenty
        __shared__ __attribute__((aligned(16))) double s_aux_hbm[kLdltAux];
        failed = ldlt_solve_mfma_lds<false>(M, n, ld, s_aux_hbm, s_x);
        solved = true;
    }
    if (failed && tid == 0) s_ok = 0;   // zero / non-finite pivot (Eigen SimplicialLDLT would report failure)
    __syncthreads();

    const int ok = s_ok;
    UH_BA_CLKL(12);
    if (ok && solved) {
        for (int i = tid; i < n; i += NT) p.xp[i] = s_x[i];   // (s_x is complete: the barrier above)
    } else if (ok) {
        // the fused solve (system in LDS, bordered: z = D^-1 L^-1 b is row n): L^T x = z
        if constexpr (USE_LDS) {
            if (n <= 64) {
                // one wave, x_i lives in lane i; row j of L (column j of L^T) is read conflict-free thanks to the odd row stride.  x_j is
                // broadcast with v_readlane (j is wave-uniform) and the n dependent steps are branch-free: a lane the step does not touch
                // multiplies by 0
                if (wv == 0) {
                    double x = lane < n ? M[(size_t)n * ld + lane] : 0.0;
                    const int lr = lane < n ? lane : n - 1;
                    double l[8], ln[8];
#pragma unroll
                    for (int t = 0; t < 8; t++) { const int jj = n - 1 - t >= 0 ? n - 1 - t : 0; l[t] = M[IX(jj, lr)]; }
                    for (int j0 = n - 1; j0 >= 0; j0 -= 8) {
#pragma unroll
                        for (int t = 0; t < 8; t++) { const int jj = j0 - 8 - t >= 0 ? j0 - 8 - t : 0; ln[t] = M[IX(jj, lr)]; }
#pragma unroll
                        for (int t = 0; t < 8; t++) { const int j = j0 - t; l[t] = lane < j ? l[t] : 0.0; }
#pragma unroll
                        for (int t = 0; t < 8; t++) x = fma(-l[t], readlane_f64(x, j0 - t > 0 ? j0 - t : 0), x);
#pragma unroll
                        for (int t = 0; t < 8; t++) l[t] = ln[t];
                    }
                    if (lane < n) { s_x[lane] = x; p.xp[lane] = x; }
                }
            } else if (n <= 128) {   // two unknowns per lane of wave 0, one broadcast per column
                backsolve2_lds(M, n, ld, s_x);
                __syncthreads();
                for (int i = tid; i < n; i += NT) p.xp[i] = s_x[i];
            } else {                 // (not reached by the launch chain: the fused solve serves n <= 126) column sweeps, two barriers per column
                for (int i = tid; i < n; i += NT) s_x[i] = M[(size_t)n * ld + i];
                __syncthreads();
                for (int j = n - 1; j >= 0; j--) {
                    const double xj = s_x[j];
                    __syncthreads();
                    for (int i = tid; i < j; i += NT) s_x[i] -= M[IX(j, i)] * xj;
                    __syncthreads();
                }
                for (int i = tid; i < n; i += NT) p.xp[i] = s_x[i];
            }
        }
    } else {
        for (int i = tid; i < n; i += NT) { p.xp[i] = 0.0; s_x[i] = 0.0; }
    }
    __syncthreads();
    UH_BA_CLKL(13);
    if (tid == 0) p.st[slot].solve_ok = ok;
    // pose update into the trial buffer (fixed poses are identical in both buffers and never touched)
    if (tid < d.nfree) pose_update_one(p, tid, cur, ok, s_x);
    UH_BA_CLKL(14);
    out.done = false; out.ok = ok; out.cur = cur; out.lambda = lambda;
}

// stand-alone form: reduced systems too large for LDS (n > 126) factorise in the HBM workspace p.S, which only one
// workgroup may use
template <bool USE_LDS, bool PACKED = false>
__global__ __launch_bounds__(PACKED ? kPackedThreads : kHbmThreads) void ba_solve_kernel(BAPtrs p, BADims d, int nsplit, int slot) {
    uh_latency_critical();
    __shared__ double s_x[6 * (PACKED ? kPackedFree : kMaxFree)];
    SolveOut o;
    solve_body<USE_LDS, PACKED>(p, d, nsplit, slot, s_x, o);
}

// Stand-alone decision: closes a round of enqueued steps (the decision of a step is otherwise applied by the NEXT step's
// schur kernel).  One wave, state slot `slot`, in place.
__global__ __launch_bounds__(64) void ba_decide_kernel(BAPtrs p, BADims d, int slot) {
    uh_latency_critical();
    const unsigned char stopv = p.stop ? *p.stop : (unsigned char)0;
    const BAState st0 = p.st[slot];
    const DecideSums sm = decide_sums(p, d, threadIdx.x, st0.lambda);
    if (st0.phase == 2 || !st0.pending || threadIdx.x != 0) return;
    UH_BA_CLKL(24);
    p.st[slot] = apply_decision(st0, sm, stopv != 0 || st0.stop_seen != 0);
    p.clk[25] = wall_clock64();
}

// ------------------------------------------------------------------------------------------------ wide problems
// More than kMaxFree free keyframes (global BA, System::globalOptimization system.cpp:7809-7939): the dense "every camera pair x
// every landmark" schur launch and the one-workgroup solve do not scale, so
//   * the camera pairs that actually share landmarks are listed on the host (pair -> its (landmark, edge1, edge2) triples in
//     landmark order, cut into work items of <= 256 triples); one workgroup per item writes a partial 6x6 (+6), one workgroup per
//     pair adds its items in order into the dense reduced system S (lower triangle, row n = right-hand side: bordered form);
//   * S is factorised as L D L^T in HBM by a right-looking blocked sweep (64-column panels: diagonal block in one workgroup's
//     LDS, panel rows one per thread, trailing update in 64x64 tiles of register-blocked fp64 FMAs — fp64 MFMA has the same peak
//     as vector FMA on gfx950, so it would buy nothing here), the border row comes out as D^-1 L^-1 b, and L^T x = z is solved
//     block by block from the last panel upwards;
//   * the LM decision that every schur workgroup of the small form takes for itself is one 64-lane launch (ba_advance_kernel).
// Linearisation, back-substitution, relabelling and results are the kernels of the small form (they never depended on nfree).
constexpr int kWNB = 64;   // panel width of the blocked factorisation

struct BAWide {
    int n_pairs, n_items, ld;
    const int* pair_s1; const int* pair_s2; const int* pair_item_ptr;   // [n_pairs], [n_pairs], [n_pairs + 1]
    const int* item_pair; const int* item_begin; const int* item_count;  // [n_items]
    const int* tri_pt; const int* tri_e1; const int* tri_e2;             // triples, pair-major, landmark order inside a pair
    double* Wpart;          // [n_items][42]
    double* S;              // (n + 1) x ld, ld = n + 1; row n = right-hand side
    double* Y;              // (n + 1) x kWNB: L*D of the current panel's rows
    int* fail;              // set by the factorisation on a zero / non-finite pivot
};

// The prologue of ba_schur_kernel as a launch of its own: previous trial's decision, lambda at the first iteration, publish.
__global__ __launch_bounds__(64) void ba_advance_kernel(BAPtrs p, BADims d, int slot) {
    const int lane = threadIdx.x;
    const BAState st0 = p.st[slot];
    const DecideSums sm = decide_sums(p, d, lane, st0.lambda);
    BAState st = (st0.phase != 2 && st0.pending) ? apply_decision(st0, sm, st0.stop_seen != 0) : st0;
    if (st.phase == 2) {
        if (lane == 0) { st.pending = 0; p.st[slot ^ 1] = st; }
        return;
    }
    if (st.iteration == 0 && st.qmax == 0) {   // computeLambdaInit: tau * max |H_jj| over poses and landmarks
        double m = 0;
        for (int i = lane; i < d.nPointBlocks; i += 64) m = fmax(m, p.part_maxdiag[i]);
        for (int i = lane; i < d.nfree * 6; i += 64) {
            const int s = i / 6, a = i - 6 * s;
            const int q = a * 6 - a * (a - 1) / 2;   // diagonal entries of the 21-entry upper triangle: 0, 6, 11, 15, 18, 20
            double v = 0;
            for (int c = 0; c < kCamChunks; c++) v += p.HppPart[((size_t)s * kCamChunks + c) * 27 + q];
            m = fmax(m, fabs(v));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
        st.lambda = 1e-5 * m; st.ni = 2;
    }
    if (lane == 0) { BAState pub = st; pub.pending = 1; pub.stop_seen = 0; p.st[slot ^ 1] = pub; }
}

// grid = n_items + nfree * kCamChunks; `slot` = the state ba_advance_kernel published
__global__ __launch_bounds__(kThreads) void ba_schurw_kernel(BAPtrs p, BADims d, BAWide w, int slot) {
    __shared__ double s_part[4 * 42];
    __shared__ double s_out[42];
    const BAState st = p.st[slot];
    if (st.phase == 2) return;
    if ((int)blockIdx.x >= w.n_items) {
        if (!st.first_trial) camera_block(p, d, blockIdx.x - w.n_items, p.poseR[st.cur], p.pts[st.cur]);
        return;
    }
    const int item = blockIdx.x, pair = w.item_pair[item];
    const bool diag = w.pair_s1[pair] == w.pair_s2[pair];
    const double lambda = st.lambda;
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; i++) acc[i] = 0;
    if ((int)threadIdx.x < w.item_count[item]) {
        const int t = w.item_begin[item] + threadIdx.x;
        const int pt = w.tri_pt[t], e1 = w.tri_e1[t], e2 = w.tri_e2[t];
        if ((p.e_active[e1] != 0) & (p.e_active[e2] != 0)) {
            double D[9], Di[9];
#pragma unroll
            for (int i = 0; i < 9; i++) D[i] = p.Hll[st.cur][9 * (size_t)pt + i];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            inv3(D, Di);
            double b1[18], b2[18];
            const double* B1 = p.Hpl[st.cur] + 18 * (size_t)e1;
            const double* B2 = p.Hpl[st.cur] + 18 * (size_t)e2;
#pragma unroll
            for (int i = 0; i < 18; i++) { b1[i] = B1[i]; b2[i] = B2[i]; }
            double l0 = 0, l1 = 0, l2 = 0;
            if (diag) { const double* bi = p.bl[st.cur] + 3 * (size_t)pt; l0 = bi[0]; l1 = bi[1]; l2 = bi[2]; }
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double y0 = b1[a * 3] * Di[0] + b1[a * 3 + 1] * Di[3] + b1[a * 3 + 2] * Di[6];
                const double y1 = b1[a * 3] * Di[1] + b1[a * 3 + 1] * Di[4] + b1[a * 3 + 2] * Di[7];
                const double y2 = b1[a * 3] * Di[2] + b1[a * 3 + 1] * Di[5] + b1[a * 3 + 2] * Di[8];
#pragma unroll
                for (int c = 0; c < 6; c++) acc[a * 6 + c] += y0 * b2[c * 3] + y1 * b2[c * 3 + 1] + y2 * b2[c * 3 + 2];
                acc[36 + a] += y0 * l0 + y1 * l1 + y2 * l2;
            }
        }
    }
    block_sum_vec<42>(acc, s_part, s_out);
    if (threadIdx.x < 42) w.Wpart[(size_t)item * 42 + threadIdx.x] = s_out[threadIdx.x];
}

// grid = n_pairs, 64 threads (42 used): S block of the pair = [Hpp + lambda I on diagonal pairs] - sum of the pair's items (in
// order), mirrored into the lower triangle; diagonal pairs also write b_p (for the decision) and the border row b_p - B D^-1 b_l
__global__ __launch_bounds__(64) void ba_assemblew_kernel(BAPtrs p, BADims d, BAWide w, int slot) {
    const BAState st = p.st[slot];
    if (st.phase == 2) return;
    const int pair = blockIdx.x, q = threadIdx.x;
    if (q >= 42) return;
    const int s1 = w.pair_s1[pair], s2 = w.pair_s2[pair];
    double v = 0;
    for (int it = w.pair_item_ptr[pair]; it < w.pair_item_ptr[pair + 1]; it++) v += w.Wpart[(size_t)it * 42 + q];
    double h = 0;
    if (s1 == s2) {
        int hq;
        if (q >= 36) hq = 21 + (q - 36);
        else { const int a = q / 6, c = q - a * 6, lo = a < c ? a : c, hi = a < c ? c : a; hq = lo * 6 - lo * (lo - 1) / 2 + (hi - lo); }
        for (int cch = 0; cch < kCamChunks; cch++) h += p.HppPart[((size_t)s1 * kCamChunks + cch) * 27 + hq];
    }
    const size_t ld = w.ld;
    if (q >= 36) {
        if (s1 == s2) { p.bp[6 * s1 + (q - 36)] = h; w.S[(size_t)d.n * ld + 6 * s1 + (q - 36)] = h - v; }
        return;
    }
    const int a = q / 6, c = q - a * 6;
    v = -v;
    if (s1 == s2) v += h + (a == c ? st.lambda : 0.0);
    const int r = 6 * s1 + a, cc = 6 * s2 + c;
    if (s1 == s2) { if (c <= a) w.S[(size_t)r * ld + cc] = v; }
    else w.S[(size_t)cc * ld + r] = v;
}

// diagonal block [k0, k0+nb) of the blocked LDL^T: one workgroup, block in LDS (row stride kWNB+1); afterwards S holds L (strict
// lower) and d (diagonal) of the block
__global__ __launch_bounds__(256) void ba_ldlw_diag_kernel(BAPtrs p, BAWide w, int slot, int k0, int nb) {
    if (p.st[slot].phase == 2) return;
    __shared__ double A[kWNB][kWNB + 1];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const size_t ld = w.ld;
    if (tid == 0) s_bad = 0;
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, c = e - i * nb; if (c <= i) A[i][c] = w.S[(size_t)(k0 + i) * ld + k0 + c]; }
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        const double dj = A[j][j];
        if (dj == 0.0 || !isfinite(dj)) { if (tid == 0) s_bad = 1; }
        const double inv = 1.0 / dj;
        // trailing update of the block with column j: A[i][c] -= (A[i][j] / d_j) * A[c][j], j < c <= i   (A[.][j] still holds L*d_j)
        const int m = nb - 1 - j;
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, c = j + 1 + e % m;
            if (c <= i) A[i][c] = fma(-(A[i][j] * inv), A[c][j], A[i][c]);
        }
        __syncthreads();
        for (int i = j + 1 + tid; i < nb; i += 256) A[i][j] *= inv;
        __syncthreads();
    }
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, c = e - i * nb; if (c <= i) w.S[(size_t)(k0 + i) * ld + k0 + c] = A[i][c]; }
    if (tid == 0 && s_bad) *w.fail = 1;
}

// panel rows r in [k0+nb, n] (the border row included), one per thread: Y_r = A_rk Lkk^-T (= L_rk D), L_rk = Y_r D^-1
__global__ __launch_bounds__(256) void ba_ldlw_panel_kernel(BAPtrs p, BADims d, BAWide w, int slot, int k0, int nb) {
    if (p.st[slot].phase == 2) return;
    __shared__ double Lk[kWNB][kWNB + 1];
    const int tid = threadIdx.x;
    const size_t ld = w.ld;
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, c = e - i * nb; if (c <= i) Lk[i][c] = w.S[(size_t)(k0 + i) * ld + k0 + c]; }
    __syncthreads();
    const int r = k0 + nb + blockIdx.x * 256 + tid;
    if (r > d.n) return;
    double y[kWNB];
#pragma unroll
    for (int j = 0; j < kWNB; j++) y[j] = j < nb ? w.S[(size_t)r * ld + k0 + j] : 0.0;
#pragma unroll
    for (int j = 0; j < kWNB; j++) {
        if (j < nb) {
            double a = y[j];
#pragma unroll
            for (int t = 0; t < j; t++) a = fma(-y[t], Lk[j][t], a);
            y[j] = a;
        }
    }
#pragma unroll
    for (int j = 0; j < kWNB; j++) {
        if (j < nb) { w.Y[(size_t)r * kWNB + j] = y[j]; w.S[(size_t)r * ld + k0 + j] = y[j] / Lk[j][j]; }
    }
}

// trailing update A_rc -= sum_t L_rt Y_ct over 64x64 tiles with tile column <= tile row; rows/columns start at k1 = k0 + nb, the
// border row n is the last row.  256 threads, 4x4 outputs each, both operand tiles staged in LDS.
__global__ __launch_bounds__(256) void ba_ldlw_update_kernel(BAPtrs p, BADims d, BAWide w, int slot, int k0, int nb, int ntile) {
    if (p.st[slot].phase == 2) return;
    // linear block index -> (tr, tc) with tc <= tr
    int tr = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((tr + 1) * (tr + 2) / 2 <= (int)blockIdx.x) ++tr;
    while (tr * (tr + 1) / 2 > (int)blockIdx.x) --tr;
    const int tc = blockIdx.x - tr * (tr + 1) / 2;
    const int k1 = k0 + nb, r0 = k1 + 64 * tr, c0 = k1 + 64 * tc;
    __shared__ double Ls[64][kWNB + 1], Ys[64][kWNB + 1];
    const int tid = threadIdx.x;
    const size_t ld = w.ld;
    for (int e = tid; e < 64 * nb; e += 256) {
        const int i = e / nb, t = e - i * nb;
        const int r = r0 + i, c = c0 + i;
        Ls[i][t] = r <= d.n ? w.S[(size_t)r * ld + k0 + t] : 0.0;
        Ys[i][t] = c < d.n ? w.Y[(size_t)c * kWNB + t] : 0.0;   // columns stop at n-1: the border row has no column
    }
    __syncthreads();
    const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0;
    for (int t = 0; t < nb; t++) {
        double l[4], y[4];
#pragma unroll
        for (int a = 0; a < 4; a++) { l[a] = Ls[ti + a][t]; y[a] = Ys[tj + a][t]; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = fma(l[a], y[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = r0 + ti + a, c = c0 + tj + b;
            if (r <= d.n && c < d.n && c <= r) w.S[(size_t)r * ld + c] -= acc[a][b];
        }
}

// L^T x = z, one launch per panel from the last one upwards: every workgroup solves the panel's unit upper triangle for itself
// (x_k from xp[k0..k0+nb)), workgroup 0 stores it, then each workgroup takes 256 earlier rows: xp_r -= sum_c L[k0+c][r] x_c
__global__ __launch_bounds__(256) void ba_ldlw_back_kernel(BAPtrs p, BAWide w, int slot, int k0, int nb) {
    if (p.st[slot].phase == 2) return;
    __shared__ double xs[kWNB];
    __shared__ double Lk[kWNB][kWNB + 1];
    const int tid = threadIdx.x;
    const size_t ld = w.ld;
    for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, c = e - i * nb; if (c < i) Lk[i][c] = w.S[(size_t)(k0 + i) * ld + k0 + c]; }
    // z lives in the border row of S (row n: D^-1 L^-1 b, updated by the panels behind this one), x goes to xp: nobody writes what
    // another workgroup of this launch still has to read
    double* z = w.S + (size_t)(w.ld - 1) * ld;
    if (tid < nb) xs[tid] = z[k0 + tid];
    __syncthreads();
    if (tid < 64) {   // one wave: x_i = z_i - sum_{j > i} L_ji x_j, from the last row of the panel upwards
        double x = tid < nb ? xs[tid] : 0.0;
        for (int j = nb - 1; j > 0; j--) {
            const double xj = __shfl(x, j);
            if (tid < j) x = fma(-Lk[j][tid], xj, x);
        }
        if (tid < nb) xs[tid] = x;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < nb) p.xp[k0 + tid] = xs[tid];
    const int r = blockIdx.x * 256 + tid;
    if (r < k0) {
        double a = z[r];
        for (int c = 0; c < nb; c++) a = fma(-w.S[(size_t)(k0 + c) * ld + r], xs[c], a);
        z[r] = a;
    }
}

// trial poses of all free keyframes from the solved increment; publishes solve_ok in the state the step runs with
__global__ __launch_bounds__(256) void ba_posew_kernel(BAPtrs p, BADims d, BAWide w, int slot) {
    const BAState st = p.st[slot];
    if (st.phase == 2) return;
    const int ok = *w.fail ? 0 : 1;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s == 0) p.st[slot].solve_ok = ok;
    if (s < d.nfree) {
        if (!ok) for (int a = 0; a < 6; a++) p.xp[6 * s + a] = 0.0;
        pose_update_one(p, s, st.cur, ok, p.xp);
    }
}

// ------------------------------------------------------------------------------------------------ backsub + trial errors
// FUSED (reduced system in LDS, n <= 126): every workgroup first solves the reduced system itself — the same assembly,
// factorisation and substitution in all of them, ~22 us that would otherwise be a one-workgroup launch of its own — keeps dx_p
// in LDS and goes on with its landmarks.  All workgroups write identical trial poses / xp / solve_ok.  Saves one kernel
// boundary and the dependent reloads behind it per LM trial.
template <bool FUSED>
__global__ __launch_bounds__(FUSED ? kFusedThreads : kThreads) void ba_backsub_kernel(BAPtrs p, BADims d, int nsplit, int slot) {
    uh_latency_critical();
    __shared__ double s_red[kThreads];
    __shared__ double s_xp[6 * kMaxFree];
    // the force-stop flag lives in pinned host memory (a PCIe round trip): one thread of the grid samples it, early, and
    // leaves it in the state for the decision
    unsigned char stopv = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.stop) stopv = *p.stop;
    int cur, ok;
    double lambda;
    if constexpr (FUSED) {
        SolveOut so;
        solve_body<true>(p, d, nsplit, slot, s_xp, so);
        if (so.done) return;
        __syncthreads();   // trial poses (written to HBM by this workgroup) and s_xp are complete
        if (threadIdx.x >= kThreads) return;   // (the waves that were here for the solve: the landmark phase below is laid out for kThreads)
        cur = so.cur; ok = so.ok; lambda = so.lambda;
    } else {
        const BAState st = p.st[slot];
        if (st.phase == 2) return;
        cur = st.cur; lambda = st.lambda; ok = st.solve_ok;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && stopv) p.st[slot].stop_seen = 1;
    const int trial = cur ^ 1;
    UH_BA_CLK(20);
    const int gl = threadIdx.x & (kLanesPerPoint - 1);
    const int pt = blockIdx.x * kPointsPerBlock + (threadIdx.x >> 3);
    double chi_part = 0, scale_part = 0;
    {
        const bool live = pt < d.P;
        const int ptc = live ? pt : 0;
        double X[3] = {p.pts[cur][3 * ptc], p.pts[cur][3 * ptc + 1], p.pts[cur][3 * ptc + 2]};
        const int b = live ? p.pt_ptr[pt] : 0, e_end = live ? p.pt_ptr[pt + 1] : 0;
        // c = bl - sum_e Hpl_e^T xp, one observation per lane, fixed butterfly over the 8 lanes
        double c[3] = {0, 0, 0};
        bool any = false;
        for (int i = b + gl; i < e_end; i += kLanesPerPoint) {
            const int e = p.pt_edges[i];
            if (!p.e_active[e]) continue;
            any = true;
            const int s = p.slot[p.e_kf[e]];
            if (s < 0) continue;
            const double* B1 = p.Hpl[cur] + 18 * (size_t)e;
            double x[6];
#pragma unroll
            for (int a = 0; a < 6; a++) { if constexpr (FUSED) x[a] = s_xp[6 * s + a]; else x[a] = p.xp[6 * s + a]; }
#pragma unroll
            for (int a = 0; a < 6; a++) { c[0] -= B1[a * 3] * x[a]; c[1] -= B1[a * 3 + 1] * x[a]; c[2] -= B1[a * 3 + 2] * x[a]; }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int o = kLanesPerPoint / 2; o > 0; o >>= 1) c[i] += __shfl_xor(c[i], o);
        }
        const unsigned long long anym = __ballot(any);
        const bool any_pt = ((anym >> ((threadIdx.x & 63) & ~(kLanesPerPoint - 1))) & 0xFFull) != 0;
        if (live && any_pt && ok) {   // every lane of the group computes the same step (identical inputs)
            const double* bi = p.bl[cur] + 3 * (size_t)pt;
            const double blv[3] = {bi[0], bi[1], bi[2]};
            c[0] += blv[0]; c[1] += blv[1]; c[2] += blv[2];
            double D[9], Di[9];
#pragma unroll
            for (int i = 0; i < 9; i++) D[i] = p.Hll[cur][9 * (size_t)pt + i];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            inv3(D, Di);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double xl = Di[a * 3] * c[0] + Di[a * 3 + 1] * c[1] + Di[a * 3 + 2] * c[2];
                if (gl == 0) scale_part += xl * (lambda * xl + blv[a]);
                X[a] += xl;
            }
        }
        if (live && gl == 0) { p.pts[trial][3 * pt] = X[0]; p.pts[trial][3 * pt + 1] = X[1]; p.pts[trial][3 * pt + 2] = X[2]; }
        // trial errors — and, speculatively, the whole linearisation at the trial estimate, into the trial side of the
        // double-buffered Hll/bl/Hpl: an accepted trial flips both and goes straight to the schur kernel, a rejected one keeps
        // the current estimate with its linearisation intact.  The lin kernel is needed only once per pass.
        double maxd_unused;
        linearize_point(p, d, trial, pt, gl, live, p.poseR[trial], X, chi_part, maxd_unused);
    }
    const double cs = block_sum(chi_part, s_red);
    const double ss = block_sum(scale_part, s_red);
    if (threadIdx.x == 0) { p.part_chi[blockIdx.x] = cs; p.part_scale[blockIdx.x] = ss; }
    UH_BA_CLK(21);
}

// ------------------------------------------------------------------------------------------------ between / after passes
// globaloptimizer_g2o.cpp:434-449: outliers to level 1, every robust kernel removed
// pass 2 is enqueued without a host round trip: the gate kernel decides on the device whether pass 1 really was complete
__global__ void ba_gate_kernel(BAPtrs p, int slot) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    BAState& st = p.st[slot];
    const bool stop = p.stop && *p.stop;
    st.gate = (st.phase == 2 && !st.stopped && !stop && !st.pending) ? 1 : 0;
    if (st.gate) st.iters_pass1 = st.iters_done;
}

__global__ void ba_relabel_kernel(BAPtrs p, BADims d, int slot, int gated) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d.E) return;
    if (gated && !p.st[slot].gate) return;
    const int cur = p.st[slot].cur;
    const int k = p.e_kf[e], pt = p.e_pt[e];
    const double* Rt = p.poseR[cur] + 12 * k;
    const double* X = p.pts[cur] + 3 * pt;
    const double z = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
    if (p.e_chi2[e] > d.chi2_th || !(z > 0.0)) p.e_active[e] = 0;
    p.e_robust[e] = 0;
}

__global__ void ba_begin_pass_kernel(BAPtrs p, int max_iters, float minChi2, int slot, int gated) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    BAState& st = p.st[slot];
    if (gated && !st.gate) return;   // pass 1 still running: the steps enqueued behind this launch simply continue it
    st.pending = 0;
    st.stop_seen = 0;
    st.phase = max_iters > 0 ? 0 : 2;
    st.iteration = 0;
    st.max_iters = max_iters;
    st.qmax = 0;
    st.solve_ok = 1;
    st.first_trial = 1;
    st.iters_done = 0;
    st.prevChi2 = FLT_MAX;   // swapped at loop entry: prev = cur = FLT_MAX before the first solve
    st.curChi2 = FLT_MAX;
    st.minChi2 = minChi2;
    if (p.stop && *p.stop) { st.phase = 2; st.stopped = 1; }
}

// getResults (:466-537): float poses (free frames), float points, bad associations
__global__ void ba_results_kernel(BAPtrs p, BADims d, const float* __restrict__ poses_in, float* __restrict__ poses_out,
                                  float* __restrict__ points_out, unsigned char* __restrict__ bad_out, int stage, int slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cur = p.st[slot].cur;
    if (stage == 0) {
        if (i < d.K) {
            float* M = poses_out + 16 * i;
            if (p.slot[i] < 0) { for (int j = 0; j < 16; j++) M[j] = poses_in[16 * i + j]; }
            else {
                const double* Rt = p.poseR[cur] + 12 * i;
                for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r * 4 + c] = (float)Rt[r * 3 + c]; M[r * 4 + 3] = (float)Rt[9 + r]; }
                M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
            }
        }
        if (i < 3 * d.P) points_out[i] = (float)p.pts[cur][i];
    } else if (i < d.E) {
        bool bad = p.e_chi2[i] > d.chi2_th;
        if (!bad) {
            const float* M = poses_out + 16 * p.e_kf[i];
            const float* X = points_out + 3 * p.e_pt[i];
            const float z = M[8] * X[0] + M[9] * X[1] + M[10] * X[2] + M[11];
            if (z < 0) bad = true;
        }
        bad_out[i] = bad;
    }
}

__global__ void ba_init_state_kernel(BAPtrs p, BADims d, const double* __restrict__ pose0, const double* __restrict__ pts0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.K) {
        double q[4];
        for (int j = 0; j < 7; j++) { const double v = pose0[7 * i + j]; p.pose[0][7 * i + j] = v; p.pose[1][7 * i + j] = v; if (j < 4) q[j] = v; }
        double R[9];
        quat_to_R(q, R);
        for (int b = 0; b < 2; b++) {
            for (int j = 0; j < 9; j++) p.poseR[b][12 * i + j] = R[j];
            for (int j = 0; j < 3; j++) p.poseR[b][12 * i + 9 + j] = pose0[7 * i + 4 + j];
        }
    }
    if (i < 3 * d.P) { p.pts[0][i] = pts0[i]; p.pts[1][i] = pts0[i]; }
    if (i < d.E) { p.e_active[i] = 1; p.e_robust[i] = 1; p.e_chi2[i] = 0; p.e_err[2 * i] = 0; p.e_err[2 * i + 1] = 0; }
    if (i == 0) { BAState z; memset(&z, 0, sizeof(z)); z.phase = 2; z.lambda = -1; z.ni = 2; p.st[0] = z; p.st[1] = z; }
}

// setParams on the device (the staged form of uh_ba_set_problem): every observation drops (problem sequence | its index + 1) into the
// (point x frame) table the persistent kernel's lanes look their observation up in.  Cells of older problems carry older sequence
// numbers and read as empty, so the table is never cleared between problems (only when it is reallocated or the 12 bits wrap).
// A (point, frame) pair that occurs twice (g2o would add two edges; this solver owns one lane per pair) and an index out of range
// are reported through a pinned word the host reads after the optimisation: err[0] = 1 out of range / 2 duplicate, err[1] = observation.
// ba_ingest_direct_kernel does that with the H2D copy folded in: the kernel reads the pinned staging block over the host link itself,
// mirrors it into HBM (header + points as 16-byte words, observations as 24-byte records) and scatters the table cells from the
// records it has in registers — one launch instead of a DMA + a launch.
// (e_begin: the launch serves observations [e_begin, E) — uh_ba_set_problem sends the records in two halves, the first while it still packs the second)
__global__ __launch_bounds__(256) void ba_ingest_direct_kernel(const unsigned char* __restrict__ host, unsigned char* __restrict__ dev, size_t head_bytes, size_t obs_off,
                                                               int E, int P, int K, unsigned* __restrict__ T, unsigned tseq, unsigned* err, int head_blocks, int obs16, int e_begin) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if ((int)blockIdx.x < head_blocks) {   // header + frame arrays + points: [0, head_bytes), a multiple of 16
        const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
        if (i < head_bytes) *reinterpret_cast<u32x4*>(dev + i) = *reinterpret_cast<const u32x4*>(host + i);
        return;
    }
    const int e = e_begin + ((int)blockIdx.x - head_blocks) * 256 + threadIdx.x;
    if (e >= E) return;
    int pt, kf;
    if (obs16) {   // 16-byte records: a third fewer bytes over the host link, which is what this kernel's time is made of
        const u32x4 r = reinterpret_cast<const u32x4*>(host + obs_off)[e];
        reinterpret_cast<u32x4*>(dev + obs_off)[e] = r;
        pt = (int)(r.x & 0xFFFFFFu); kf = (int)(r.x >> 24);
    } else {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(host + obs_off) + 3 * (size_t)e;
        const unsigned long long w0 = src[0], w1 = src[1], w2 = src[2];
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(dev + obs_off) + 3 * (size_t)e;
        dst[0] = w0; dst[1] = w1; dst[2] = w2;
        pt = (int)(unsigned)w0; kf = (int)(unsigned)(w0 >> 32);
    }
    unsigned code = 0;
    if ((unsigned)pt >= (unsigned)P || (unsigned)kf >= (unsigned)K) code = 1;
    else {
        const unsigned old = atomicExch(T + (size_t)pt * K + kf, tseq | (unsigned)(e + 1));
        if ((old & 0xFFF00000u) == tseq) code = 2;
    }
    if (code) {
        __hip_atomic_store(err + 1, (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#include "ba_persist.hpp"

}  // namespace

// Staging block of the staged setParams (pinned host memory and its mirror in HBM): byte offsets, fixed by the capacities
struct StageLayout {
    size_t pose0, poseR0, intr, slot, free_kf, fix_kf;   // derived on the host per problem (K-sized)
    size_t poses_in, fixed, intr_f;                      // the caller's frame arrays
    size_t points, obs;                                  // P x 3 float, then E x uh_ba_obs: the copy ends behind the last observation
};
// Result block (pinned host memory the kernel's tail writes): byte offsets
struct ResLayout { size_t poses, state, points, chi2, bad, bytes, bytes_no_chi2; };   // chi2 last: it is handed over only on request

struct uh_ba {
    uh_ctx* ctx = nullptr;
    bool have_problem = false;
    BADims dims{};
    BAPtrs ptrs{};
    uh_ba_params params{5, 0.0, 0.0, 1.0f};
    std::vector<float> poses_in;          // K x 16 (fixed frames are returned unchanged)
    std::vector<unsigned char> fixed;
    uh::DevBuf arena;                     // one allocation for everything on the device
    uh::DevBuf d_poses_in, d_poses_out, d_points_out, d_bad;
    uh::PinBuf up_pin;                    // pinned mirror of the arena's constant prefix (setParams of the launch chain / wide form)
    hipEvent_t ev_up = nullptr; bool up_in_flight = false;
    uh::PinBuf res_pin;                   // getResults of the launch chain / wide form: the D2H copies land here (asynchronous DMA), then a host copy
    double* d_pose0 = nullptr; double* d_pts0 = nullptr;
    unsigned char* h_stop = nullptr;      // pinned, device-visible force-stop flag
    int iters[2] = {0, 0};
    int nsplit = 1;
    bool dense = false; SchurDense sd{};   // 17-32 free keyframes of the launch chain: ba_schur_dense_kernel + ba_schur_reduce_kernel (Spart then holds ONE chunk)
    bool wide = false;                    // more than kMaxFree free keyframes: sparse pair lists + blocked dense LDL^T in HBM
    bool persist = false;                 // 1..8 free keyframes: the whole optimisation is ONE persistent launch (ba_persist.hpp)
    int persist_blocked = 0;              // > 0: the persistent form's workgroups did not all become resident lately (another spinning kernel shares the GPU): so many of the next problems take the launch chain
    BAPersist pq{};
    uh::DevBuf parena;                    // the persistent form's exchange buffers (tagged words only; [0, 64): the error word)
    unsigned parena_gen = ~0u;            // the allocation the exchange tags refer to (a new one is zeroed)
    // ---- staged setParams / in-place getResults (the persistent form's whole host side)
    unsigned char* h_stage = nullptr;     // pinned, device-visible: derived header | caller's frame arrays | points | observations
    size_t h_stage_bytes = 0;
    int cap_K = 0, cap_P = 0, cap_E = 0;
    StageLayout slay{};
    uh::DevBuf dstage;                    // its mirror in HBM (same layout), filled by ONE H2D copy per problem
    hipEvent_t ev_stage = nullptr;        // recorded behind that copy: the staging block may be rewritten once it has completed
    bool stage_in_flight = false;
    uh::DevBuf dT;                        // (point x frame) -> problem sequence << 20 | observation index + 1
    unsigned stage_gen = 0, problem_stage_gen = 0;   // uh_ba_map_staging bumps the first; set_problem_fast records it: the residency fallback
                                          // re-reads the staging block and must not do so once the caller may have refilled it
    bool persist_not_resident = false;    // the last persistent launch ended because a workgroup never became resident (as opposed to a launch that vanished)
    unsigned dT_gen = ~0u, tseq = 0;
    uh::DevBuf dscratch;                  // BAState[2], 64 phase clocks, completion counter
    unsigned done_base = 0;               // what the completion counter holds before the next launch
    unsigned char* h_res = nullptr;       // pinned, device-visible result block: the kernel's tail writes getResults' outputs here
    size_t h_res_bytes = 0;
    ResLayout rlay{};
    uh::DevBuf d_res;                     // the result block in HBM the kernel's tail fills first (same layout)
    bool want_chi2 = true;                // hand the per-observation chi2 (208 KB of the 270 KB of results at 26k observations) over to the host
    bool fast = false;                    // the current problem was set through the staged path (persistent form)
    int p_nf = 0;                         // lanes per landmark of the persistent instantiation in use
    int p_lds_set[4] = {0, 0, 0, 0};      // dynamic LDS already granted to the instantiations (hipFuncSetAttribute once, not per problem)
    int max_lds = 0;                      // hipDeviceAttributeMaxSharedMemoryPerBlock of the device
    int job_kind = 0;                     // worker: 0 optimize, 1 setParams + optimize (uh_ba_solve_async)
    bool knob_hbm_solve = false, knob_prebuilt_off = false;   // UH_BA_SOLVE=hbm / UH_BA_PREBUILT=0 as uh_ba_set_problem found them
    const uh_ba_problem* job_problem = nullptr; uh_ba_problem job_problem_copy{};
    int job_dims[3] = {0, 0, 0};
    uh_ba_params job_params{}; bool job_has_params = false;
    unsigned p_seq = 0;                   // launches of the persistent kernel by this optimizer: 20 bits of it tag the exchanged words
    int p_lds = 0;
    BAWide wd{};
    int step = 0;                         // LM steps enqueued since uh_ba_optimize began: step s reads state slot s & 1
    bool optimized = false;
    // uh_ba_optimize_async: a persistent worker thread (the reference's mapper thread, mapmanager.cpp:150) runs uh_ba_optimize
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> job{0};              // 0 idle, 1 requested, 2 running, 3 done (result in job_rc), -1 quit; written under `mu`, spun on without it
    int job_rc = 0;
    std::string job_err;
    const volatile uint8_t* job_stop = nullptr;
    ~uh_ba() {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); job = -1; }
            cv.notify_all();
            worker.join();
        }
        if (h_stop) (void)hipHostFree(h_stop);
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_res) (void)hipHostFree(h_res);
        if (ev_stage) (void)hipEventDestroy(ev_stage);
        if (ev_up) (void)hipEventDestroy(ev_up);
    }
};

namespace {

void quat_from_R_host(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    if (q[3] < 0) for (int a = 0; a < 4; a++) q[a] = -q[a];
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= n;
}

struct Arena {
    size_t off = 0;
    template <typename T> size_t take(size_t count) { off = (off + 255) & ~(size_t)255; size_t o = off; off += count * sizeof(T); return o; }
};

int enqueue_steps(uh_ba* b, int nsteps, bool pass_start) {
    hipStream_t st = b->ctx->stream;
    const BADims& d = b->dims;
    const int npairs = d.nfree * (d.nfree + 1) / 2;
    const bool hbm_forced = b->knob_hbm_solve;   // (tests: the HBM solve on systems of any size; read by uh_ba_set_problem on the caller's thread)
    const int use_lds = (d.n <= 126 && !hbm_forced) ? 1 : 0;   // (21 free cameras: 127 rows = the two-rows-per-lane factorisation's limit; 129 KB of LDS)
    const size_t lds = use_lds ? (size_t)(d.n + 1) * (d.n + 1) * sizeof(double) : 0;   // n rows of S + the right-hand-side row
    for (int s = 0; s < nsteps; s++) {
        const int slot = b->step & 1;   // state left by the previous step (or by begin_pass / the closing decide kernel)
        if (pass_start && s == 0) UH_LAUNCH(b->ctx,ba_lin_kernel, dim3(d.nPointBlocks + d.nfree * kCamChunks), dim3(kThreads), 0, b->ptrs, d, slot);
        if (b->wide) {
            const BAWide& W = b->wd;
            const int run = slot ^ 1;   // the state this step runs with (published by the advance kernel)
            UH_LAUNCH(b->ctx, ba_advance_kernel, dim3(1), dim3(64), 0, b->ptrs, d, slot);
            UH_HIP_CHECK(hipMemsetAsync(W.S, 0, sizeof(double) * (size_t)W.ld * W.ld, st));
            UH_HIP_CHECK(hipMemsetAsync(W.fail, 0, sizeof(int), st));
            UH_LAUNCH(b->ctx, ba_schurw_kernel, dim3(W.n_items + d.nfree * kCamChunks), dim3(kThreads), 0, b->ptrs, d, W, run);
            UH_LAUNCH(b->ctx, ba_assemblew_kernel, dim3(W.n_pairs), dim3(64), 0, b->ptrs, d, W, run);
            for (int k0 = 0; k0 < d.n; k0 += kWNB) {
                const int nb = std::min(kWNB, d.n - k0), rows = d.n + 1 - (k0 + nb);   // rows behind the panel, border row included
                UH_LAUNCH(b->ctx, ba_ldlw_diag_kernel, dim3(1), dim3(256), 0, b->ptrs, W, run, k0, nb);
                UH_LAUNCH(b->ctx, ba_ldlw_panel_kernel, dim3(uh_div_up(rows, 256)), dim3(256), 0, b->ptrs, d, W, run, k0, nb);
                const int ntile = uh_div_up(rows, 64);
                if (k0 + nb < d.n) UH_LAUNCH(b->ctx, ba_ldlw_update_kernel, dim3(ntile * (ntile + 1) / 2), dim3(256), 0, b->ptrs, d, W, run, k0, nb, ntile);
            }
            for (int k0 = ((d.n - 1) / kWNB) * kWNB; k0 >= 0; k0 -= kWNB) {
                const int nb = std::min(kWNB, d.n - k0);
                UH_LAUNCH(b->ctx, ba_ldlw_back_kernel, dim3(std::max(uh_div_up(k0, 256), 1)), dim3(256), 0, b->ptrs, W, run, k0, nb);
            }
            UH_LAUNCH(b->ctx, ba_posew_kernel, dim3(uh_div_up(d.nfree, 256)), dim3(256), 0, b->ptrs, d, W, run);
            UH_LAUNCH(b->ctx, ba_backsub_kernel<false>, dim3(d.nPointBlocks), dim3(kThreads), 0, b->ptrs, d, b->nsplit, run);
            b->step++;
            continue;
        }
        // which solve follows: fused into the back-substitution launch (system in LDS, n <= 126), stand-alone on a packed triangle in LDS
        // (22-32 free cameras: 148 KB at 32, beside the kernel's static arrays — both checked against the device's limit), else in HBM
        const size_t packed = ((size_t)(d.n + 1) * (d.n + 2) / 2 + 2 + kLdltAux) * sizeof(double);   // rows 0 .. n (row n = right-hand side) + the solve's scratch
        bool use_packed = false;
        if (!use_lds) {
            static const size_t packed_static = [] { hipFuncAttributes fa{}; return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&ba_solve_kernel<false, true>)) == hipSuccess ? fa.sharedSizeBytes : (size_t)1 << 30; }();
            if (b->max_lds <= 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, b->ctx->device) == hipSuccess) b->max_lds = v; }
            use_packed = d.nfree <= kPackedFree && packed + packed_static <= (size_t)std::max(b->max_lds, 0) && !hbm_forced;
        }
        // dense Schur form: the reduce launch leaves the FINISHED system in p.S in the following solve's layout (1: row stride n + 1,
        // 2: packed) and the solve copies it (nsplit 0); the HBM solve keeps its own assembly from Spart (0)
        const bool pre_off = b->knob_prebuilt_off;   // (A/B knob, read by uh_ba_set_problem on the caller's thread: getenv here would race a setenv elsewhere)
        // (the solve in HBM, 33-64 free cameras, factorises p.S in place: row stride n + 1 = mode 1, and nothing to copy)
        const int pre_mode = (b->dense && !pre_off) ? ((use_lds || !use_packed) ? 1 : 2) : 0;
        const int ns = pre_mode ? 0 : b->nsplit;
        if (b->dense) {
            const SchurDense& sd = b->sd;
            if (sd.SP > 0) {
                const dim3 gridw(d.nfree * kCamChunks + sd.G * sd.SP);
                const size_t ldsw = (size_t)(24 * sd.ys + 24) * sizeof(double);
                if (sd.nown == 12) UH_LAUNCH(b->ctx, ba_schur_dense_wide_kernel<12>, gridw, dim3(kThreads), ldsw, b->ptrs, d, sd, slot);
                else if (sd.nown == 16) UH_LAUNCH(b->ctx, ba_schur_dense_wide_kernel<16>, gridw, dim3(kThreads), ldsw, b->ptrs, d, sd, slot);
                else UH_LAUNCH(b->ctx, ba_schur_dense_wide_kernel<20>, gridw, dim3(kThreads), ldsw, b->ptrs, d, sd, slot);
            } else
                UH_LAUNCH(b->ctx, ba_schur_dense_kernel, dim3(d.nfree * kCamChunks + sd.G), dim3(kThreads), (size_t)(48 * sd.ys + 48) * sizeof(double), b->ptrs, d, sd, slot);
            UH_LAUNCH(b->ctx, ba_schur_reduce_kernel, dim3(sd.T + uh_div_up(d.n, kThreads)), dim3(kThreads * kReduceGroups), 0, b->ptrs, d, sd, slot ^ 1, pre_mode);
        } else
        UH_LAUNCH(b->ctx,ba_schur_kernel, dim3(std::max(npairs, 1) * b->nsplit + d.nfree * kCamChunks), dim3(kThreads), 0, b->ptrs, d, b->nsplit, slot);
        // the solve in HBM takes the finished system from a launch of its own (every pair's words in parallel) instead of assembling alone
        const bool pre_hbm = !b->dense && !use_lds && !use_packed && npairs > 0 && !pre_off;
        if (pre_hbm) UH_LAUNCH(b->ctx, ba_assemble_pairs_kernel, dim3(uh_div_up(npairs * 42, kThreads)), dim3(kThreads), 0, b->ptrs, d, b->nsplit, slot ^ 1);
        if (use_lds) {
            UH_LAUNCH(b->ctx,ba_backsub_kernel<true>, dim3(d.nPointBlocks), dim3(kFusedThreads), lds, b->ptrs, d, ns, slot ^ 1);
        } else {
            if (use_packed)
                UH_LAUNCH(b->ctx, (ba_solve_kernel<false, true>), dim3(1), dim3(kPackedThreads), packed, b->ptrs, d, ns, slot ^ 1);
            else
                UH_LAUNCH(b->ctx,ba_solve_kernel<false>, dim3(1), dim3(kHbmThreads), 0, b->ptrs, d, pre_hbm ? 0 : ns, slot ^ 1);
            UH_LAUNCH(b->ctx,ba_backsub_kernel<false>, dim3(d.nPointBlocks), dim3(kThreads), 0, b->ptrs, d, ns, slot ^ 1);
        }
        b->step++;
    }
    // the last trial's decision (every other one is taken by the following step's schur kernel)
    UH_LAUNCH(b->ctx,ba_decide_kernel, dim3(1), dim3(64), 0, b->ptrs, d, b->step & 1);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

// waits for the stream and returns the state of the valid slot; keeps forwarding the caller's stop flag meanwhile
int wait_state(uh_ba* b, BAState* hs, const volatile uint8_t* stop_asap) {
    hipStream_t st = b->ctx->stream;
    UH_HIP_CHECK(hipMemcpyAsync(hs, b->ptrs.st + (b->step & 1), sizeof(BAState), hipMemcpyDeviceToHost, st));
    if (stop_asap && b->h_stop) {   // forward the caller's flag while the stream drains; the mapper thread yields between polls
        hipEvent_t ev;
        UH_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        const hipError_t er = hipEventRecord(ev, st);
        if (er != hipSuccess) {
            (void)hipEventDestroy(ev);
            uh::set_error("hipEventRecord failed: %s", hipGetErrorString(er));
            return UH_ENODEVICE;
        }
        while (hipEventQuery(ev) == hipErrorNotReady) {
            if (*stop_asap) *b->h_stop = 1;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        (void)hipEventDestroy(ev);
    }
    UH_HIP_CHECK(hipStreamSynchronize(st));
    return UH_OK;
}

// rejected trials consumed steps of the budget: enqueue short rounds until the running pass is finished
int finish_pass(uh_ba* b, BAState* hs, const volatile uint8_t* stop_asap) {
    for (int round = 0; round < 64 && hs->phase != 2; round++) {
        int rc = enqueue_steps(b, 4, false);
        if (rc) return rc;
        if ((rc = wait_state(b, hs, stop_asap))) return rc;
    }
    if (hs->phase != 2) {   // cannot happen with <= 10 trials per iteration and <= 2*nIters iterations, but never report success on a running pass
        uh::set_error("uh_ba_optimize: the LM pass did not finish within its step budget (phase %d, iteration %d)", hs->phase, hs->iteration);
        return UH_EINVAL;
    }
    return UH_OK;
}

// Admission of persistent launches: their workgroups spin on each other, so every launch must become fully resident.  One CU holds
// one such workgroup (LDS), hence at most 7/8 of the device's CUs' worth of them are in flight per device at any time (process-local:
// two PROCESSES sharing one GPU can still starve each other into the 3 s time-out — see INTEGRATION.md).
struct PersistAdmission {
    std::mutex m; std::condition_variable cv; int used = 0;
    void acquire(int g, int budget) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&]() { return used == 0 || used + g <= budget; }); used += g; }
    void release(int g) { { std::lock_guard<std::mutex> l(m); used -= g; } cv.notify_all(); }
};
PersistAdmission g_persist_adm[16];   // per device (index & 15)

// GlobalOptimizerG2O::optimize as ONE launch (ba_persist.hpp): both passes, relabelling, every trial
int run_persistent(uh_ba* b, const volatile uint8_t* stop_asap, int n1, int n2, float mc) {
    hipStream_t st = b->ctx->stream;
    if (const char* e = getenv("UH_BA_FAIL_RESIDENCY")) {   // test hook: behave as if the workgroups had not all become resident (nothing is launched)
        if (e[0] == '1') {
            (void)hipStreamSynchronize(st);   // (the ingest in front of the launch has run)
            b->stage_in_flight = false;
            b->persist_not_resident = true;
            uh::set_error("uh_ba_optimize: the persistent kernel's workgroups did not all become resident (forced by UH_BA_FAIL_RESIDENCY)");
            return UH_ENODEVICE;
        }
    }
    BAPersist q = b->pq;
    q.n1 = n1; q.n2 = n2; q.minChi2 = mc;
    q.stop_at_begin = (b->h_stop && *b->h_stop) ? 1 : 0;
    PersistAdmission& adm = g_persist_adm[b->ctx->device & 15];
    const int budget = std::max(q.G, (b->ctx->num_cus > 0 ? b->ctx->num_cus : 256) * 7 / 8);
    struct Hold { PersistAdmission& a; int g; Hold(PersistAdmission& a_, int g_, int bud) : a(a_), g(g_) { a.acquire(g, bud); } ~Hold() { a.release(g); } } hold(adm, q.G, budget);
    static std::atomic<unsigned> s_launch{0};
    q.launch_id = ++s_launch;   // (never 0) the error and completion words of this launch carry it
    // exchanged data words carry 20 bits of this optimizer's own launch count (+ 12 bits of round number): the buffer was zeroed when it
    // was allocated, every launch rewrites what it reads, and when the 20 bits wrap it is zeroed again
    if (((++b->p_seq) & 0xFFFFFu) == 0) {
        ++b->p_seq;
        UH_HIP_CHECK(hipMemsetAsync(b->parena.p, 0, b->parena.cap, st));
    }
    q.tag_base = (b->p_seq & 0xFFFFFu) << 12;
    {
        static const long long ticks = [] { const char* e = getenv("UH_BA_RESIDENCY_TIMEOUT_MS"); const long long ms = e ? atoll(e) : 0; return ms > 0 ? ms * 100000ll : kPTimeoutTicksDefault; }();
        q.timeout_ticks = ticks;
    }
    q.done_base = b->done_base;
    BAState hs;
    void* d_pin = nullptr;
    UH_HIP_CHECK(hipHostGetDevicePointer(&d_pin, b->h_stop, 0));
    volatile unsigned long long* h_done = reinterpret_cast<volatile unsigned long long*>(b->h_stop + 192);
    q.host_state = reinterpret_cast<BAState*>(static_cast<unsigned char*>(d_pin) + 64);
    q.host_done = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(d_pin) + 192);
    switch (b->p_nf) {
        case 8: UH_LAUNCH(b->ctx, ba_persist_kernel<8>, dim3(q.G), dim3(kPThreads), (size_t)b->p_lds, b->ptrs, b->dims, q); break;
        case 16: UH_LAUNCH(b->ctx, ba_persist_kernel<16>, dim3(q.G), dim3(kPThreads), (size_t)b->p_lds, b->ptrs, b->dims, q); break;
        default: uh::set_error("uh_ba_optimize: no persistent instantiation for %d lanes per landmark", b->p_nf); return UH_EINVAL;
    }
    UH_HIP_CHECK(hipGetLastError());
    bool err = false;
    b->persist_not_resident = false;
    {
        // the kernel's last act is a system-scope release store of (launch id << 32 | 1) behind the results and the final state: polling
        // that word costs a few hundred nanoseconds of latency, a stream synchronisation + pageable D2H copies cost ~45 us per optimize()
        const unsigned long long ok_word = ((unsigned long long)q.launch_id << 32) | 1ull, err_word = ok_word + 1;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            const unsigned long long w = *h_done;
            if (w == ok_word) break;
            if (w == err_word) { err = true; b->persist_not_resident = true; break; }
            if (stop_asap && *stop_asap) *b->h_stop = 1;
            __builtin_ia32_pause();
            if ((spin & 1023) == 1023) {
                if (hipStreamQuery(st) == hipSuccess && *h_done != ok_word && *h_done != err_word) { err = true; break; }   // the launch is gone without reporting
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) { err = true; break; }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(&hs, b->h_stop + 64, sizeof(BAState));
    }
    b->stage_in_flight = false;   // (the kernel ran behind the staging copy on the same stream)
    if (err) {
        (void)hipStreamSynchronize(st);
        (void)hipMemsetAsync(b->dscratch.as<char>() + 768, 0, 4, st);   // the completion count of an aborted launch is meaningless
        b->done_base = 0;
        if (b->persist_not_resident) uh::set_error("uh_ba_optimize: the persistent kernel's workgroups did not all become resident (%d workgroups, %d bytes of LDS each)", q.G, b->p_lds);
        else {
            const hipError_t le = hipGetLastError();
            uh::set_error("uh_ba_optimize: the persistent launch ended without reporting (%s): a device fault or a 30 s stall, not a residency problem — not retried",
                          le == hipSuccess ? "no HIP error pending" : hipGetErrorString(le));
        }
        return UH_ENODEVICE;
    }
    b->done_base += 2u * (unsigned)q.G;
    const unsigned* h_err = reinterpret_cast<const unsigned*>(b->h_stop + 200);
    if (h_err[0]) {   // left by ba_ingest_kernel (it ran in front of this launch on the same stream)
        const unsigned e = h_err[1] < (unsigned)b->dims.E ? h_err[1] : 0;
        int ept, ekf;
        if (b->pq.obs16) { const unsigned w = reinterpret_cast<const unsigned*>(b->h_stage + b->slay.obs)[4 * (size_t)e]; ept = (int)(w & 0xFFFFFFu); ekf = (int)(w >> 24); }
        else { const uh_ba_obs* ob = reinterpret_cast<const uh_ba_obs*>(b->h_stage + b->slay.obs); ept = ob[e].point; ekf = ob[e].frame; }
        if (h_err[0] == 2) uh::set_error("uh_ba_set_problem: point %d observed twice by frame %d (observation %u)", ept, ekf, e);
        else uh::set_error("uh_ba_set_problem: observation %u references point %d / frame %d out of range", e, ept, ekf);
        return UH_EINVAL;
    }
    b->iters[0] = hs.gate ? hs.iters_pass1 : hs.iters_done;
    b->iters[1] = hs.gate ? hs.iters_done : 0;
    b->optimized = true;
    return UH_OK;
}

}  // namespace

// spin (with the CPU's pause hint) until `done()` or for at most `us` microseconds
template <typename F>
static inline void spin_until(F&& done, int us) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; !done(); ++i) {
        __builtin_ia32_pause();
        if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return;
    }
}

extern "C" {

int uh_ba_create(uh_ctx* ctx, uh_ba** out) {
    UH_REQUIRE(ctx && out, "uh_ba_create: NULL argument");
    uh_ba* b = new uh_ba();
    b->ctx = ctx;
    // one pinned, device-visible block: [0] force-stop byte, [64..) the final BAState of a persistent launch, [192] its completion word
    if (hipHostMalloc(reinterpret_cast<void**>(&b->h_stop), 256, hipHostMallocMapped) != hipSuccess) b->h_stop = nullptr;
    if (b->h_stop) std::memset(b->h_stop, 0, 256);
    if (const char* e = getenv("UH_BA_SEQ0")) b->p_seq = (unsigned)strtoul(e, nullptr, 0);   // tests: start next to the wrap of the exchange tags
    *out = b;
    return UH_OK;
}

void uh_ba_destroy(uh_ba* b) { delete b; }

}  // extern "C"

// setParams for the forms that keep host-built tables: the launch chain (more free keyframes than the persistent kernel is
// instantiated for, or a window whose fixed frames do not fit its LDS) and the wide form (global BA).  Arrays anywhere in host memory.
static int set_problem_tables(uh_ba* b, const uh_ba_problem* pr) {
    b->fast = false;
    const int K = pr->n_frames, P = pr->n_points, E = pr->n_obs;
    std::vector<int> slot(K, -1), free_kf;
    for (int k = 0; k < K; k++) if (!pr->fixed[k]) { slot[k] = (int)free_kf.size(); free_kf.push_back(k); }
    const int nfree = (int)free_kf.size();
    constexpr int kMaxFreeWide = 4096;
    UH_REQUIRE(nfree <= kMaxFreeWide, "uh_ba_set_problem: %d free keyframes (supported: <= %d)", nfree, kMaxFreeWide);
    // wide form: sparse camera-pair lists + blocked dense LDL^T in HBM (see "wide problems" above); UH_BA_WIDE=1 forces it for
    // small problems (tests compare the two forms on the same inputs)
    const bool wide = nfree > kMaxFree || (getenv("UH_BA_WIDE") && atoi(getenv("UH_BA_WIDE")) != 0 && nfree > 0);
    b->wide = wide;
    for (int e = 0; e < E; e++)
        UH_REQUIRE(pr->obs_point[e] >= 0 && pr->obs_point[e] < P && pr->obs_frame[e] >= 0 && pr->obs_frame[e] < K,
                   "uh_ba_set_problem: observation %d references point %d / frame %d out of range", e, pr->obs_point[e], pr->obs_frame[e]);
    // host-side graph structure
    std::vector<int> pt_ptr(P + 1, 0), pt_edges(E), cam_ptr(nfree + 1, 0), cam_edges, edge_of(wide ? 1 : (size_t)P * std::max(nfree, 1), -1);
    for (int e = 0; e < E; e++) pt_ptr[pr->obs_point[e] + 1]++;
    for (int p = 0; p < P; p++) pt_ptr[p + 1] += pt_ptr[p];
    { std::vector<int> fill(pt_ptr.begin(), pt_ptr.end() - 1); for (int e = 0; e < E; e++) pt_edges[fill[pr->obs_point[e]]++] = e; }
    for (int e = 0; e < E; e++) { const int s = slot[pr->obs_frame[e]]; if (s >= 0) cam_ptr[s + 1]++; }
    for (int s = 0; s < nfree; s++) cam_ptr[s + 1] += cam_ptr[s];
    cam_edges.resize(cam_ptr[nfree]);
    { std::vector<int> fill(cam_ptr.begin(), cam_ptr.end() - 1);
      for (int e = 0; e < E; e++) { const int s = slot[pr->obs_frame[e]]; if (s >= 0) { cam_edges[fill[s]++] = e;
          if (!wide) {
              UH_REQUIRE(edge_of[(size_t)pr->obs_point[e] * nfree + s] < 0, "uh_ba_set_problem: point %d observed twice by frame %d", pr->obs_point[e], pr->obs_frame[e]);
              edge_of[(size_t)pr->obs_point[e] * nfree + s] = e; } } } }
    // wide form: (s1 <= s2) camera pairs that share landmarks, their (landmark, edge1, edge2) triples in landmark order, <= 256 per item
    std::vector<int> w_pair_s1, w_pair_s2, w_pair_item_ptr, w_item_pair, w_item_begin, w_item_count, w_tri_pt, w_tri_e1, w_tri_e2;
    if (wide) {
        // counting sort by pair key: pass 1 counts the triples of every (s1 <= s2), pass 2 drops them into place — landmarks
        // are visited in ascending order both times, so a pair's triples end up in landmark order
        std::vector<std::vector<std::pair<int, int>>> obs_of(P);
        for (int pt = 0; pt < P; pt++) {
            auto& obs = obs_of[pt];
            for (int i = pt_ptr[pt]; i < pt_ptr[pt + 1]; i++) { const int e = pt_edges[i], s = slot[pr->obs_frame[e]]; if (s >= 0) obs.push_back({s, e}); }
            std::sort(obs.begin(), obs.end());
            for (size_t i = 1; i < obs.size(); i++)
                UH_REQUIRE(obs[i].first != obs[i - 1].first, "uh_ba_set_problem: point %d observed twice by frame %d", pt, free_kf[obs[i].first]);
        }
        std::vector<size_t> key_count((size_t)nfree * nfree + 1, 0);
        for (int pt = 0; pt < P; pt++) {
            const auto& obs = obs_of[pt];
            for (size_t i = 0; i < obs.size(); i++)
                for (size_t j = i; j < obs.size(); j++) key_count[(size_t)obs[i].first * nfree + obs[j].first + 1]++;
        }
        for (size_t k = 0; k < (size_t)nfree * nfree; k++) key_count[k + 1] += key_count[k];
        const size_t n_tri = key_count[(size_t)nfree * nfree];
        UH_REQUIRE(n_tri < (size_t)1 << 31, "uh_ba_set_problem: %zu co-observation triples exceed the 32-bit index range", n_tri);
        w_tri_pt.resize(n_tri); w_tri_e1.resize(n_tri); w_tri_e2.resize(n_tri);
        {
            std::vector<size_t> fill(key_count.begin(), key_count.end() - 1);
            for (int pt = 0; pt < P; pt++) {
                const auto& obs = obs_of[pt];
                for (size_t i = 0; i < obs.size(); i++)
                    for (size_t j = i; j < obs.size(); j++) {
                        const size_t at = fill[(size_t)obs[i].first * nfree + obs[j].first]++;
                        w_tri_pt[at] = pt; w_tri_e1[at] = obs[i].second; w_tri_e2[at] = obs[j].second;
                    }
            }
        }
        std::vector<char> have_diag(std::max(nfree, 1), 0);
        auto add_pair = [&](int s1, int s2, int begin, int count) {
            w_pair_s1.push_back(s1); w_pair_s2.push_back(s2); w_pair_item_ptr.push_back((int)w_item_pair.size());
            for (int o = 0; o < count; o += kThreads) { w_item_pair.push_back((int)w_pair_s1.size() - 1); w_item_begin.push_back(begin + o); w_item_count.push_back(std::min(kThreads, count - o)); }
            if (s1 == s2) have_diag[s1] = 1;
        };
        for (int s1 = 0; s1 < nfree; s1++)
            for (int s2 = s1; s2 < nfree; s2++) {
                const size_t k = (size_t)s1 * nfree + s2;
                if (key_count[k + 1] > key_count[k]) add_pair(s1, s2, (int)key_count[k], (int)(key_count[k + 1] - key_count[k]));
            }
        for (int sfree = 0; sfree < nfree; sfree++) if (!have_diag[sfree]) add_pair(sfree, sfree, 0, 0);   // a camera without landmarks still owns its diagonal block
        w_pair_item_ptr.push_back((int)w_item_pair.size());
    }
    std::vector<double> pose0(7 * (size_t)K), pts0(3 * (size_t)P), intr(4 * (size_t)K), uv(2 * (size_t)E), w(E);
    for (int k = 0; k < K; k++) {   // toSE3Quat (globaloptimizer_g2o.cpp:80-90): float 4x4 -> double R,t -> quaternion
        const float* M = pr->poses_f2g + 16 * k;
        const double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        quat_from_R_host(R, &pose0[7 * k]);
        pose0[7 * k + 4] = M[3]; pose0[7 * k + 5] = M[7]; pose0[7 * k + 6] = M[11];
        for (int j = 0; j < 4; j++) intr[4 * k + j] = pr->intr[4 * k + j];
    }
    for (size_t i = 0; i < pts0.size(); i++) pts0[i] = pr->points[i];
    for (int e = 0; e < E; e++) { uv[2 * e] = pr->obs_uv[2 * e]; uv[2 * e + 1] = pr->obs_uv[2 * e + 1]; w[e] = pr->obs_inv_sigma[e]; }
    b->poses_in.assign(pr->poses_f2g, pr->poses_f2g + 16 * (size_t)K);
    b->fixed.assign(pr->fixed, pr->fixed + K);

    BADims& d = b->dims;
    d.K = K; d.P = P; d.E = E; d.nfree = nfree; d.n = 6 * nfree;
    d.nPointBlocks = std::max(uh_div_up(P, kPointsPerBlock), 1);
    d.delta = b->params.huber_delta; d.dsqr = d.delta * d.delta; d.chi2_th = b->params.chi2_threshold;

    // carve one arena
    Arena A;
    const size_t o_pt_ptr = A.take<int>(P + 1), o_pt_edges = A.take<int>(E), o_cam_ptr = A.take<int>(nfree + 1), o_cam_edges = A.take<int>(cam_edges.size());
    const size_t o_e_pt = A.take<int>(E), o_e_kf = A.take<int>(E), o_uv = A.take<double>(2 * (size_t)E), o_w = A.take<double>(E);
    const size_t o_slot = A.take<int>(K), o_free = A.take<int>(std::max(nfree, 1)), o_intr = A.take<double>(4 * (size_t)K), o_edge_of = A.take<int>(edge_of.size());
    const size_t o_pose0 = A.take<double>(7 * (size_t)K), o_pts0 = A.take<double>(3 * (size_t)P);
    size_t o_pose[2], o_poseR[2], o_pts[2];
    for (int i = 0; i < 2; i++) { o_pose[i] = A.take<double>(7 * (size_t)K); o_poseR[i] = A.take<double>(12 * (size_t)K); o_pts[i] = A.take<double>(3 * (size_t)P); }
    const size_t o_act = A.take<unsigned char>(E), o_rob = A.take<unsigned char>(E), o_err = A.take<double>(2 * (size_t)E), o_chi2 = A.take<double>(E);
    size_t o_Hll[2], o_bl[2], o_Hpl[2];
    for (int i = 0; i < 2; i++) { o_Hll[i] = A.take<double>(9 * (size_t)P); o_bl[i] = A.take<double>(3 * (size_t)P); o_Hpl[i] = A.take<double>(18 * (size_t)E); }
    const size_t o_Hpp = A.take<double>(27 * (size_t)kCamChunks * std::max(nfree, 1)), o_bp = A.take<double>(std::max(d.n, 1));
    const int npairs_h = nfree * (nfree + 1) / 2;
    b->nsplit = std::max(1, std::min(kMaxSplit, uh_div_up(P, kThreads)));
    // every workgroup of the back-substitution launch assembles ALL npairs x nsplit partials itself: with many camera pairs fewer, fatter
    // landmark chunks win (measured, 3000 landmarks: 17 / 20 / 32 free cameras 2.20 / 2.90 / 9.9 ms with 12 chunks, 1.77 / 2.14 / 7.3 with 2)
    if (npairs_h > 32) b->nsplit = std::max(1, std::min(b->nsplit, uh_div_up(300, npairs_h)));   // (17 free cameras: 2 chunks, measured best — scripts/ba_chain_kernels.py with UH_BA_NSPLIT)
    if (const char* e = getenv("UH_BA_NSPLIT")) b->nsplit = std::max(1, std::min(kMaxSplit, atoi(e)));   // (measurement override: scripts/ba_chain_kernels.py)
    // knobs the launch chain consults per step, possibly on the optimiser's worker thread: read ONCE here, on the caller's thread
    { const char* e = getenv("UH_BA_SOLVE"); b->knob_hbm_solve = e && std::string(e) == "hbm"; }
    { const char* e = getenv("UH_BA_PREBUILT"); b->knob_prebuilt_off = e && atoi(e) == 0; }
    // dense Schur form: windows of 17-32 free keyframes (UH_BA_SCHUR_DENSE=0 keeps the pair form)
    {
        const char* e = getenv("UH_BA_SCHUR_DENSE");
        b->dense = !wide && nfree >= 17 && nfree <= kMaxFree && !(e && atoi(e) == 0);   // (17-32: 7..12 tile rows, one workgroup per landmark group; 33-64: the wide kernel)
        if (b->dense) {
            SchurDense& sd = b->sd;
            sd.ntt = uh_div_up(d.n, 16); sd.T = sd.ntt * (sd.ntt + 1) / 2;
            sd.ys = 16 * sd.ntt + ((sd.ntt & 1) ? 0 : 16);   // row stride = 16 (mod 32) doubles: the four k-rows of an MFMA operand read fall on disjoint banks
            const bool widek = nfree > 32;
            const int chunks = std::max(uh_div_up(P, widek ? 8 : 16), 1);
            const int gmax = getenv("UH_BA_DENSE_G") ? std::max(1, atoi(getenv("UH_BA_DENSE_G"))) : (widek ? 96 : 224);   // (tuning knob: scripts/ba_chain_kernels.py)
            sd.cpw = uh_div_up(chunks, gmax); sd.G = uh_div_up(chunks, sd.cpw);
            sd.SP = 0; sd.nown = 0;
            if (widek) {   // tile list cut over SP workgroups of at most 80 tiles, 4 waves x nown tiles each (kernel instantiations: 12, 16, 20)
                sd.SP = uh_div_up(sd.T, 80);
                const int need = uh_div_up(uh_div_up(sd.T, sd.SP), 4);
                sd.nown = need <= 12 ? 12 : (need <= 16 ? 16 : 20);
            }
            b->nsplit = 1;
        }
    }
    const size_t o_dpart = A.take<double>(b->dense ? (size_t)b->sd.G * b->sd.T * 256 : 1), o_dbpart = A.take<double>(b->dense ? (size_t)b->sd.G * 16 * b->sd.ntt : 1);
    const size_t o_S = A.take<double>((size_t)(d.n + 1) * (d.n + 1)), o_Sp = A.take<double>(wide ? 42 : (size_t)b->nsplit * std::max(npairs_h, 1) * 42), o_xp = A.take<double>(std::max(d.n, 1));
    const size_t wn_pairs = w_pair_s1.size(), wn_items = w_item_pair.size(), wn_tri = w_tri_pt.size();
    const size_t o_wps1 = A.take<int>(wn_pairs + 1), o_wps2 = A.take<int>(wn_pairs + 1), o_wpip = A.take<int>(wn_pairs + 2);
    const size_t o_wip = A.take<int>(wn_items + 1), o_wib = A.take<int>(wn_items + 1), o_wic = A.take<int>(wn_items + 1);
    const size_t o_wtp = A.take<int>(wn_tri + 1), o_wt1 = A.take<int>(wn_tri + 1), o_wt2 = A.take<int>(wn_tri + 1);
    const size_t o_wpart = A.take<double>((wn_items + 1) * 42), o_wY = A.take<double>(wide ? (size_t)(d.n + 1) * kWNB : 1), o_wfail = A.take<int>(4);
    const size_t o_plc = A.take<double>(d.nPointBlocks), o_pmd = A.take<double>(d.nPointBlocks), o_pc = A.take<double>(d.nPointBlocks), o_ps = A.take<double>(d.nPointBlocks);
    const size_t o_st = A.take<BAState>(2), o_clk = A.take<long long>(64);
    int rc = b->arena.reserve(A.off + 256);
    if (rc) return rc;
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    char* base = b->arena.as<char>();
    // The constant arrays of the problem form one contiguous prefix of the arena ([o_pt_ptr, end of pts0)): they are gathered in a pinned
    // mirror of that prefix and leave as ONE asynchronous copy (fourteen pageable copies were fourteen staged, synchronous transfers:
    // 0.40 ms of setParams at 48 000 observations).  The wide form's pair / triple lists (up to hundreds of MB) keep their direct copies.
    const size_t prefix_bytes = o_pts0 + 3 * (size_t)P * sizeof(double);
    if (b->up_in_flight) { UH_HIP_CHECK(hipEventSynchronize(b->ev_up)); b->up_in_flight = false; }
    if ((rc = b->up_pin.reserve(prefix_bytes + 256))) return rc;
    if (!b->ev_up) UH_HIP_CHECK(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
    char* hp = b->up_pin.as<char>();
    auto up = [&](size_t off, const void* src, size_t bytes) -> int {
        if (!bytes) return UH_OK;
        if (off + bytes <= prefix_bytes) std::memcpy(hp + off, src, bytes);
        else UH_HIP_CHECK(hipMemcpyAsync(base + off, src, bytes, hipMemcpyHostToDevice, st));
        return UH_OK;
    };
    if ((rc = up(o_pt_ptr, pt_ptr.data(), pt_ptr.size() * 4))) return rc;
    if ((rc = up(o_pt_edges, pt_edges.data(), pt_edges.size() * 4))) return rc;
    if ((rc = up(o_cam_ptr, cam_ptr.data(), cam_ptr.size() * 4))) return rc;
    if ((rc = up(o_cam_edges, cam_edges.data(), cam_edges.size() * 4))) return rc;
    if ((rc = up(o_e_pt, pr->obs_point, (size_t)E * 4))) return rc;
    if ((rc = up(o_e_kf, pr->obs_frame, (size_t)E * 4))) return rc;
    if ((rc = up(o_uv, uv.data(), uv.size() * 8))) return rc;
    if ((rc = up(o_w, w.data(), w.size() * 8))) return rc;
    if ((rc = up(o_slot, slot.data(), slot.size() * 4))) return rc;
    if ((rc = up(o_free, free_kf.data(), free_kf.size() * 4))) return rc;
    if ((rc = up(o_intr, intr.data(), intr.size() * 8))) return rc;
    if ((rc = up(o_edge_of, edge_of.data(), edge_of.size() * 4))) return rc;
    if (wide) {
        if ((rc = up(o_wps1, w_pair_s1.data(), wn_pairs * 4))) return rc;
        if ((rc = up(o_wps2, w_pair_s2.data(), wn_pairs * 4))) return rc;
        if ((rc = up(o_wpip, w_pair_item_ptr.data(), (wn_pairs + 1) * 4))) return rc;
        if ((rc = up(o_wip, w_item_pair.data(), wn_items * 4))) return rc;
        if ((rc = up(o_wib, w_item_begin.data(), wn_items * 4))) return rc;
        if ((rc = up(o_wic, w_item_count.data(), wn_items * 4))) return rc;
        if ((rc = up(o_wtp, w_tri_pt.data(), wn_tri * 4))) return rc;
        if ((rc = up(o_wt1, w_tri_e1.data(), wn_tri * 4))) return rc;
        if ((rc = up(o_wt2, w_tri_e2.data(), wn_tri * 4))) return rc;
    }
    if ((rc = up(o_pose0, pose0.data(), pose0.size() * 8))) return rc;
    if ((rc = up(o_pts0, pts0.data(), pts0.size() * 8))) return rc;
    if ((rc = b->d_poses_in.reserve(16 * (size_t)K * 4))) return rc;
    if ((rc = b->d_poses_out.reserve(16 * (size_t)K * 4))) return rc;
    if ((rc = b->d_points_out.reserve(std::max<size_t>(3 * (size_t)P * 4, 16)))) return rc;
    if ((rc = b->d_bad.reserve(std::max<size_t>(E, 16)))) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(base, hp, prefix_bytes, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipEventRecord(b->ev_up, st));
    b->up_in_flight = true;
    UH_HIP_CHECK(hipMemcpyAsync(b->d_poses_in.p, pr->poses_f2g, 16 * (size_t)K * 4, hipMemcpyHostToDevice, st));   // (640 bytes, pageable: staged at once)
    b->persist = false;
    if (wide) UH_HIP_CHECK(hipStreamSynchronize(st));   // the wide form's host lists die here
    BAPtrs& p = b->ptrs;
    p.pt_ptr = (int*)(base + o_pt_ptr); p.pt_edges = (int*)(base + o_pt_edges); p.cam_ptr = (int*)(base + o_cam_ptr); p.cam_edges = (int*)(base + o_cam_edges);
    p.e_pt = (int*)(base + o_e_pt); p.e_kf = (int*)(base + o_e_kf); p.e_uv = (double*)(base + o_uv); p.e_w = (double*)(base + o_w);
    p.slot = (int*)(base + o_slot); p.free_kf = (int*)(base + o_free); p.intr = (double*)(base + o_intr); p.edge_of = (int*)(base + o_edge_of);
    for (int i = 0; i < 2; i++) { p.pose[i] = (double*)(base + o_pose[i]); p.poseR[i] = (double*)(base + o_poseR[i]); p.pts[i] = (double*)(base + o_pts[i]); }
    p.e_active = (unsigned char*)(base + o_act); p.e_robust = (unsigned char*)(base + o_rob); p.e_err = (double*)(base + o_err); p.e_chi2 = (double*)(base + o_chi2);
    for (int i = 0; i < 2; i++) { p.Hll[i] = (double*)(base + o_Hll[i]); p.bl[i] = (double*)(base + o_bl[i]); p.Hpl[i] = (double*)(base + o_Hpl[i]); }
    p.HppPart = (double*)(base + o_Hpp); p.bp = (double*)(base + o_bp);
    p.S = (double*)(base + o_S); p.Spart = (double*)(base + o_Sp); p.xp = (double*)(base + o_xp);
    p.dpart = (double*)(base + o_dpart); p.dbpart = (double*)(base + o_dbpart);
    {
        BAWide& W = b->wd;
        W.n_pairs = (int)wn_pairs; W.n_items = (int)wn_items; W.ld = d.n + 1;
        W.pair_s1 = (const int*)(base + o_wps1); W.pair_s2 = (const int*)(base + o_wps2); W.pair_item_ptr = (const int*)(base + o_wpip);
        W.item_pair = (const int*)(base + o_wip); W.item_begin = (const int*)(base + o_wib); W.item_count = (const int*)(base + o_wic);
        W.tri_pt = (const int*)(base + o_wtp); W.tri_e1 = (const int*)(base + o_wt1); W.tri_e2 = (const int*)(base + o_wt2);
        W.Wpart = (double*)(base + o_wpart); W.S = p.S; W.Y = (double*)(base + o_wY); W.fail = (int*)(base + o_wfail);
    }
    p.part_lin_chi = (double*)(base + o_plc); p.part_maxdiag = (double*)(base + o_pmd); p.part_chi = (double*)(base + o_pc); p.part_scale = (double*)(base + o_ps);
    p.st = (BAState*)(base + o_st);
    p.clk = (long long*)(base + o_clk);
    p.stop = nullptr;
    if (b->h_stop) {
        void* dflag = nullptr;
        if (hipHostGetDevicePointer(&dflag, b->h_stop, 0) == hipSuccess) p.stop = (const volatile unsigned char*)dflag;
    }
    b->d_pose0 = (double*)(base + o_pose0);
    b->d_pts0 = (double*)(base + o_pts0);
    b->have_problem = true;
    return UH_OK;
}

// ------------------------------------------------------------------------------------------------ staged setParams (persistent form)
static StageLayout stage_layout(int Kc, int Pc) {
    StageLayout L;
    Arena A;
    L.pose0 = A.take<double>(7 * (size_t)Kc); L.poseR0 = A.take<double>(12 * (size_t)Kc); L.intr = A.take<double>(4 * (size_t)Kc);
    L.slot = A.take<int>(Kc); L.free_kf = A.take<int>(Kc); L.fix_kf = A.take<int>(Kc);
    L.poses_in = A.take<float>(16 * (size_t)Kc); L.fixed = A.take<unsigned char>(Kc); L.intr_f = A.take<float>(4 * (size_t)Kc);
    L.points = A.take<float>(3 * (size_t)Pc);
    L.obs = A.take<uh_ba_obs>(0);
    return L;
}
static ResLayout res_layout(int Kc, int Pc, int Ec) {
    ResLayout R;
    Arena A;
    R.poses = A.take<float>(16 * (size_t)Kc); R.state = A.take<double>(7 * (size_t)Kc); R.points = A.take<float>(3 * (size_t)Pc);
    R.bad = A.take<unsigned char>(Ec);
    R.bytes_no_chi2 = (A.off + 7) & ~(size_t)7;
    R.chi2 = A.take<double>(Ec);
    R.bytes = (A.off + 7) & ~(size_t)7;   // copied to the host as 8-byte words
    return R;
}

// The staging and result blocks for (K, P, E), grown with headroom and kept across problems.  Waits for the H2D copy of the previous
// problem first: the caller is about to overwrite the block.
static int ensure_staging(uh_ba* b, int K, int P, int E) {
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    if (b->stage_in_flight) { UH_HIP_CHECK(hipEventSynchronize(b->ev_stage)); b->stage_in_flight = false; }
    if (b->h_stage && K <= b->cap_K && P <= b->cap_P && E <= b->cap_E) return UH_OK;
    const int Kc = K <= b->cap_K ? b->cap_K : K + K / 4 + 4, Pc = P <= b->cap_P ? b->cap_P : P + P / 4 + 64, Ec = E <= b->cap_E ? b->cap_E : E + E / 4 + 1024;
    const StageLayout L = stage_layout(Kc, Pc);
    const ResLayout R = res_layout(Kc, Pc, Ec);
    const size_t bytes = L.obs + (size_t)Ec * sizeof(uh_ba_obs) + 256;
    unsigned char* ns = nullptr; unsigned char* nr = nullptr;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&ns), bytes, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&nr), R.bytes + 256, hipHostMallocMapped);
    if (e != hipSuccess) {
        if (ns) (void)hipHostFree(ns);
        uh::set_error("uh_ba: hipHostMalloc of the staging / result blocks (%zu + %zu bytes) failed: %s", bytes, R.bytes, hipGetErrorString(e));
        return UH_ENOMEM;
    }
    if (b->h_stage || b->h_res) UH_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));   // (a kernel may still be writing results into the old block)
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    if (b->h_res) (void)hipHostFree(b->h_res);
    std::memset(nr, 0, R.bytes + 256);
    b->h_stage = ns; b->h_stage_bytes = bytes; b->h_res = nr; b->h_res_bytes = R.bytes + 256;
    b->cap_K = Kc; b->cap_P = Pc; b->cap_E = Ec; b->slay = L; b->rlay = R;
    b->have_problem = false; b->optimized = false;   // (results of an earlier problem lived in the old block)
    if (!b->ev_stage) UH_HIP_CHECK(hipEventCreateWithFlags(&b->ev_stage, hipEventDisableTiming));
    return UH_OK;
}

struct PersistPlan { bool ok; int NF, Lw, G, krows, nelem, SL, max_fix, kfix, lds, nb4, nblk, KS, off_cam; };

template <int NF>
static int persist_lds_bytes(const PersistPlan& pl, int n) { return persist_lds<NF>(pl.krows, n, pl.max_fix, pl.kfix, pl.off_cam, pl.KS, pl.SL).total_bytes; }

// Which persistent instantiation runs a window of `nfree` free keyframes (0: none): one lane per (landmark, free-camera slot), so the
// lanes per landmark are the number of free cameras rounded up to 8 or 16.  Beyond 16 the reduced system (6 nfree + 1)^2 doubles and
// the landmark panel no longer share one workgroup's LDS, and 256 / 32 = 8 landmarks per workgroup would need more workgroups than the
// chip has compute units: those windows run the launch chain.
static int persist_lanes(int nfree) {
    if (const char* e = getenv("UH_BA_NF")) { const int v = atoi(e); if ((v == 8 || v == 16) && nfree <= v) return v; }   // (A/B: force the wider instantiation)
    return nfree <= 8 ? 8 : (nfree <= 16 ? 16 : 0);
}

template <int NF>
static hipError_t persist_grant_lds(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_persist_kernel<NF>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Does the problem run as ONE persistent launch?  Limits come from the device, not from constants: every workgroup must be resident
// at once (one per compute unit — their LDS blocks do not fit two to a CU), and its LDS block must fit what the device grants.
static PersistPlan plan_persistent(uh_ba* b, int K, int P, int E, int nfree) {
    PersistPlan pl{};
    const char* form = getenv("UH_BA_FORM");
    if (form && std::string(form) == "legacy") return pl;
    if (getenv("UH_BA_WIDE") && atoi(getenv("UH_BA_WIDE")) != 0) return pl;
    void* dflag = nullptr;
    if (!b->h_stop || hipHostGetDevicePointer(&dflag, b->h_stop, 0) != hipSuccess) return pl;   // (the kernel reports through pinned memory)
    if (nfree < 1 || P < 1 || E >= (1 << 20) - 1) return pl;
    if (b->persist_blocked > 0) { --b->persist_blocked; return pl; }
    const int NF = persist_lanes(nfree);
    if (!NF) return pl;
    const int kLwMax = kPThreads / NF;   // one lane per (landmark, free-camera slot)
    int Lw = std::max(std::min(8, kLwMax), std::min(kLwMax, uh_div_up(P, 64)));
    if (const char* e = getenv("UH_BA_LW")) Lw = std::max(1, std::min(kLwMax, atoi(e)));
    const int G = uh_div_up(P, Lw);
    const int cus = b->ctx->num_cus > 0 ? b->ctx->num_cus : 256;
    if (G > cus) return pl;
    const int kfix = K - nfree;
    if (kfix > 255) return pl;
    pl.NF = NF; pl.Lw = Lw; pl.G = G; pl.kfix = kfix;
    pl.max_fix = Lw * kfix;   // bound: every landmark of the tile seen by every fixed frame (nothing is counted on the host)
    pl.krows = (3 * Lw + 15) & ~15;
    const int n = 6 * nfree;
    pl.nb4 = (n + 3) / 4; pl.nblk = pl.nb4 * (pl.nb4 + 1) / 2;
    pl.off_cam = pl.nblk * 16;   // the product part of a partial: the upper 4 x 4 blocks (vector FMA; the v_mfma_f64 form of rounds 2-4 was slower on gfx950 and is gone: docs/DESIGN_history_r1_r3.md, profiles/r03_mfma_f64.json)
    pl.nelem = pl.off_cam + NF * 27 + 6 * NF + 4;
    pl.SL = (uh_div_up(pl.nelem, G) + 1) & ~1;
    // K-splits of the Schur product: work items = nblk * KS over 256 lanes, each krows / KS rows deep; the splits' partial blocks must
    // fit the LDS region the reduced system occupies later ((n + 1)^2 doubles)
    pl.KS = 1;
    {
        int best = 1 << 30;
        for (int ks = 1; ks <= 6; ks++) {
            if (ks > 1 && ks * pl.off_cam > (n + 1) * (n + 1) + 1452) break;
            const int rounds = uh_div_up(pl.nblk * ks, kPThreads), depth = (uh_div_up(pl.krows, ks) + 3) & ~3;
            if (rounds * depth < best) { best = rounds * depth; pl.KS = ks; }
        }
    }
    pl.lds = NF == 8 ? persist_lds_bytes<8>(pl, n) : persist_lds_bytes<16>(pl, n);
    if (b->max_lds <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, b->ctx->device) != hipSuccess || v <= 0) v = 64 * 1024;
        b->max_lds = v;
    }
    if (pl.lds > b->max_lds) return pl;
    const int cls = NF == 8 ? 0 : 1;
    if (pl.lds > b->p_lds_set[cls]) {   // once per size class, not per problem; a refusal selects the launch chain instead of failing setParams
        if ((NF == 8 ? persist_grant_lds<8>(pl.lds) : persist_grant_lds<16>(pl.lds)) != hipSuccess) { (void)hipGetLastError(); return pl; }
        b->p_lds_set[cls] = pl.lds;
    }
    pl.ok = true;
    return pl;
}

// setParams of the persistent form on a filled staging block: header on the host (K-sized), ONE H2D copy, the ingest kernel.
// No host-built table, no stream synchronisation; every device buffer is kept across problems.
// pack(e0, e1, as16, exact, oob): writes observation records [e0, e1) (e0 even) into the staging block in the 16- or 24-byte form; clears `exact`
// when a scalar does not survive the 16-byte form, sets `oob` on an index out of range.  NULL: the records are in the block already (24-byte form
// unless obs16).  With a packer the records leave in TWO ingest launches, the first running over the host link while the second half is packed.
// The 16-byte observation records eight at a time where the host has AVX-512 (round 6; the SSE loop below packs two per step): same records,
// same range test (sign bits of i and n - 1 - i OR-ed over the range), same exactness test of the information scalars.
// Returns the number of observations packed (a multiple of eight, from e0).
__attribute__((target("avx512f,avx512vl,avx512dq"))) static int pack_obs16_avx512(const int32_t* pt_, const int32_t* kf_, const float* uv_, const double* w_, int e0, int e1, int P, int K,
                                                                                 unsigned char* ob, bool& exact_out, unsigned& oob) {
    const __m256i pmax = _mm256_set1_epi32(P - 1), kmax = _mm256_set1_epi32(K - 1);
    __m256i bad = _mm256_setzero_si256();
    __mmask8 exact = 0xFF;
    const __m512i idx0 = _mm512_setr_epi32(0, 16, 17, 8, 1, 18, 19, 9, 2, 20, 21, 10, 3, 22, 23, 11);
    const __m512i idx1 = _mm512_setr_epi32(4, 24, 25, 12, 5, 26, 27, 13, 6, 28, 29, 14, 7, 30, 31, 15);
    int i = e0;
    unsigned char* dst = ob + 16 * (size_t)e0;
    for (; i + 8 <= e1; i += 8, dst += 128) {
        const __m256i pt = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(pt_ + i));
        const __m256i kf = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(kf_ + i));
        const __m512 uv = _mm512_loadu_ps(uv_ + 2 * i);
        const __m512d w = _mm512_loadu_pd(w_ + i);
        bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_or_si256(pt, _mm256_sub_epi32(pmax, pt)), _mm256_or_si256(kf, _mm256_sub_epi32(kmax, kf))));
        const __m256 wf = _mm512_cvtpd_ps(w);
        exact &= _mm512_cmp_pd_mask(_mm512_cvtps_pd(wf), w, _CMP_EQ_OQ);
        const __m256i pk = _mm256_or_si256(pt, _mm256_slli_epi32(kf, 24));
        const __m512 a = _mm512_insertf32x8(_mm512_castps256_ps512(_mm256_castsi256_ps(pk)), wf, 1);   // pk0..pk7 | w0..w7
        _mm512_storeu_ps(dst, _mm512_permutex2var_ps(a, idx0, uv));        // {pk, u, v, w} of observations 0..3
        _mm512_storeu_ps(dst + 64, _mm512_permutex2var_ps(a, idx1, uv));   // 4..7
    }
    if (exact != 0xFF) exact_out = false;
    oob |= _mm256_movemask_ps(_mm256_castsi256_ps(bad)) ? 1u : 0u;
    return i - e0;
}

using ObsPacker = std::function<void(int, int, bool, bool&, unsigned&)>;
static int set_problem_fast(uh_ba* b, int K, int P, int E, const PersistPlan& pl, bool obs16 = false, const ObsPacker* pack = nullptr, unsigned* oob_out = nullptr) {
    const StageLayout& L = b->slay;
    unsigned char* hs = b->h_stage;
    const float* poses = reinterpret_cast<const float*>(hs + L.poses_in);
    const unsigned char* fixed = hs + L.fixed;
    const float* intr_f = reinterpret_cast<const float*>(hs + L.intr_f);
    double* pose0 = reinterpret_cast<double*>(hs + L.pose0); double* R0 = reinterpret_cast<double*>(hs + L.poseR0);
    double* intr = reinterpret_cast<double*>(hs + L.intr);
    int* slot = reinterpret_cast<int*>(hs + L.slot); int* free_kf = reinterpret_cast<int*>(hs + L.free_kf); int* fix_kf = reinterpret_cast<int*>(hs + L.fix_kf);
    int nfree = 0, nfix = 0;
    for (int k = 0; k < K; k++) {   // toSE3Quat (globaloptimizer_g2o.cpp:80-90): float 4x4 -> double R,t -> quaternion
        if (!fixed[k]) { slot[k] = nfree; free_kf[nfree++] = k; } else { slot[k] = -1; fix_kf[nfix++] = k; }
        const float* M = poses + 16 * k;
        const double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        double* qq = pose0 + 7 * k;
        quat_from_R_host(R, qq);
        qq[4] = M[3]; qq[5] = M[7]; qq[6] = M[11];
        for (int j = 0; j < 4; j++) intr[4 * k + j] = intr_f[4 * k + j];
        double* Rk = R0 + 12 * k;   // R | t of the snapshot, from the normalised quaternion like the launch chain's init kernel
        const double tx = 2 * qq[0], ty = 2 * qq[1], tz = 2 * qq[2];
        const double twx = tx * qq[3], twy = ty * qq[3], twz = tz * qq[3];
        const double txx = tx * qq[0], txy = ty * qq[0], txz = tz * qq[0];
        const double tyy = ty * qq[1], tyz = tz * qq[1], tzz = tz * qq[2];
        Rk[0] = 1 - (tyy + tzz); Rk[1] = txy - twz; Rk[2] = txz + twy;
        Rk[3] = txy + twz; Rk[4] = 1 - (txx + tzz); Rk[5] = tyz - twx;
        Rk[6] = txz - twy; Rk[7] = tyz + twx; Rk[8] = 1 - (txx + tyy);
        Rk[9] = qq[4]; Rk[10] = qq[5]; Rk[11] = qq[6];
    }
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    int rc;
    // ---- device buffers (all of them survive the problem)
    if ((rc = b->dstage.reserve(b->h_stage_bytes))) return rc;
    // the (point x frame) table is indexed pt * K + kf with THIS problem's K: P * K cells, not the product of the staging capacities
    // (those only grow: one global BA staged through this object — 250 k points x 400 frames — would have made every later local
    // BA reserve and clear 380 MB).  Only problems that fit the persistent form come here (P <= Lw x workgroups, K <= 271).
    if ((rc = b->dT.reserve((size_t)P * K * sizeof(unsigned) + 256))) return rc;
    if (b->dT_gen != b->dT.gen || (b->tseq & 0xFFFu) == 0xFFFu) {   // fresh memory, or the 12 bits wrap: no cell may look current
        UH_HIP_CHECK(hipMemsetAsync(b->dT.p, 0, b->dT.cap, st));
        b->dT_gen = b->dT.gen; b->tseq = 0;
    }
    ++b->tseq;
    b->problem_stage_gen = b->stage_gen;
    unsigned tseq = (b->tseq & 0xFFFu) << 20;
    if (!b->dscratch.p) {
        if ((rc = b->dscratch.reserve(1024))) return rc;
        UH_HIP_CHECK(hipMemsetAsync(b->dscratch.p, 0, b->dscratch.cap, st));
        b->done_base = 0;
    }
    Arena PA;
    PA.off = 256;   // [0, 64): the error word — never part of a tagged region
    const size_t o_part = PA.take<unsigned long long>(2 * (size_t)pl.G * pl.G * pl.SL), o_red = PA.take<unsigned long long>(2 * (size_t)pl.G * pl.SL);
    const size_t o_pc = PA.take<unsigned long long>(2 * 4 * (size_t)pl.G);
    if ((rc = b->parena.reserve(PA.off + 256))) return rc;
    if (b->parena_gen != b->parena.gen) {   // the exchange words validate themselves by tag: only fresh memory is cleared (and on the tag wrap)
        UH_HIP_CHECK(hipMemsetAsync(b->parena.p, 0, b->parena.cap, st));
        b->parena_gen = b->parena.gen;
    }
    char* db = b->dstage.as<char>();
    unsigned* h_err = reinterpret_cast<unsigned*>(b->h_stop + 200);   // pinned: [200] ingest error code, [204] observation
    h_err[0] = 0; h_err[1] = 0;
    void* d_pin = nullptr;
    UH_HIP_CHECK(hipHostGetDevicePointer(&d_pin, b->h_stop, 0));
    unsigned* d_err = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(d_pin) + 200);
    {   // the kernel fetches the staging block over the host link itself: no DMA (the DMA + ingest-launch pair it replaced —
        // 0.588 against 0.574-0.581 ms per step — was kept behind UH_BA_INGEST=copy until round 5)
        void* d_hs = nullptr;
        UH_HIP_CHECK(hipHostGetDevicePointer(&d_hs, hs, 0));
        const int head_blocks = uh_div_up((int)(L.obs / 16), 256);
        auto ingest = [&](bool head, int e0, int e1, bool as16, unsigned seq) {
            UH_LAUNCH(b->ctx, ba_ingest_direct_kernel, dim3((head ? head_blocks : 0) + uh_div_up(std::max(e1 - e0, 1), 256)), dim3(256), 0, static_cast<const unsigned char*>(d_hs),
                      reinterpret_cast<unsigned char*>(db), L.obs, L.obs, e1, P, K, b->dT.as<unsigned>(), seq, d_err, head ? head_blocks : 0, as16 ? 1 : 0, e0);
        };
        if (!pack) ingest(true, 0, E, obs16, tseq);
        else {
            // two halves: the first half's records cross the host link (~25 GB/s: 10 us) while the host packs the second (round 5; one launch
            // behind the whole packing loop put its 21 us in front of the optimisation)
            static const int split_env = [] { const char* e = getenv("UH_BA_INGEST_SPLIT"); return e ? atoi(e) : -1; }();   // (A/B knob: 0 one launch, 1 two halves)
            // round 6: with the AVX-512 packer the whole loop is ~4 us — less than the second launch costs (two half transfers are latency-bound,
            // 13 us each; one is 20) — so the records go out in ONE launch there (step 0.4395 -> 0.4365 ms); the SSE packer keeps the halves
            static const bool has_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
            const bool split = split_env >= 0 ? split_env != 0 : !(has_avx512 && !getenv("UH_BA_NO_AVX512"));
            const int mid = (split && E >= 4096) ? (E / 2) & ~1 : E;
            bool exact = true;
            unsigned oob = 0;
            (*pack)(0, mid, obs16, exact, oob);
            if (obs16 && !exact) { obs16 = false; exact = true; (*pack)(0, mid, false, exact, oob); }
            if (!oob) ingest(true, 0, mid, obs16, tseq);
            if (mid < E && !oob) {
                (*pack)(mid, E, obs16, exact, oob);
                if (obs16 && !exact) {   // the second half does not fit the 16-byte form: everything again as 24-byte records, under a new table sequence
                    obs16 = false; exact = true;
                    UH_HIP_CHECK(hipStreamSynchronize(st));   // (the first half's launch still reads the 16-byte records this pass overwrites; rare path)
                    h_err[0] = 0; h_err[1] = 0;
                    (*pack)(0, E, false, exact, oob);
                    if ((b->tseq & 0xFFFu) == 0xFFFu) { UH_HIP_CHECK(hipMemsetAsync(b->dT.p, 0, b->dT.cap, st)); b->tseq = 0; }
                    ++b->tseq;
                    tseq = (b->tseq & 0xFFFu) << 20;
                    if (!oob) ingest(true, 0, E, false, tseq);
                } else if (!oob) ingest(false, mid, E, obs16, tseq);
            }
            if (oob_out) *oob_out = oob;
            if (oob) { (void)hipStreamSynchronize(st); h_err[0] = 0; h_err[1] = 0; return UH_EINVAL; }   // (a launch of the first half may be in flight: nothing of this problem survives)
        }
    }
    UH_HIP_CHECK(hipEventRecord(b->ev_stage, st));   // behind the last reader of the staging block
    b->stage_in_flight = true;
    UH_HIP_CHECK(hipGetLastError());
    // ---- kernel arguments
    BADims& d = b->dims;
    d.K = K; d.P = P; d.E = E; d.nfree = nfree; d.n = 6 * nfree;
    d.nPointBlocks = std::max(uh_div_up(P, kPointsPerBlock), 1);
    d.delta = b->params.huber_delta; d.dsqr = d.delta * d.delta; d.chi2_th = b->params.chi2_threshold;
    BAPtrs& p = b->ptrs;
    p = BAPtrs{};
    p.slot = reinterpret_cast<const int*>(db + L.slot); p.free_kf = reinterpret_cast<const int*>(db + L.free_kf); p.intr = reinterpret_cast<const double*>(db + L.intr);
    p.st = b->dscratch.as<BAState>();
    p.clk = reinterpret_cast<long long*>(b->dscratch.as<char>() + 256);
    p.stop = static_cast<const volatile unsigned char*>(d_pin);
    BAPersist& q = b->pq;
    q = BAPersist{};
    q.G = pl.G; q.Lw = pl.Lw; q.krows = pl.krows; q.SL = pl.SL; q.nelem = pl.nelem; q.max_fix = pl.max_fix; q.kfix = pl.kfix;
    { const char* e = getenv("UH_BA_SPEC"); q.speculate = e && e[0] == '0' ? 0 : 1; }   // (read per launch: the A/B scripts and a test switch it within one process)
    q.nb4 = pl.nb4; q.nblk = pl.nblk; q.KS = pl.KS;
    q.T = b->dT.as<unsigned>(); q.tseq = tseq;
    q.obs16 = obs16 ? 1 : 0;
    q.obs = reinterpret_cast<const uh_ba_obs*>(db + L.obs); q.points = reinterpret_cast<const float*>(db + L.points);
    q.poses_in = reinterpret_cast<const float*>(db + L.poses_in); q.fix_kf = reinterpret_cast<const int*>(db + L.fix_kf);
    q.pose0 = reinterpret_cast<const double*>(db + L.pose0); q.poseR0 = reinterpret_cast<const double*>(db + L.poseR0);
    char* pb = b->parena.as<char>();
    q.errw = reinterpret_cast<unsigned long long*>(pb);
    q.part = reinterpret_cast<unsigned long long*>(pb + o_part); q.red = reinterpret_cast<unsigned long long*>(pb + o_red); q.partC = reinterpret_cast<unsigned long long*>(pb + o_pc);
    void* d_res = nullptr;
    UH_HIP_CHECK(hipHostGetDevicePointer(&d_res, b->h_res, 0));
    b->rlay = res_layout(K, P, E);   // (exact sizes: the hand-over copies one contiguous range; the blocks are sized by the capacities)
    const ResLayout& R = b->rlay;
    if ((rc = b->d_res.reserve(b->h_res_bytes))) return rc;
    unsigned char* rb = b->d_res.as<unsigned char>();
    q.r_dev = b->d_res.as<unsigned long long>(); q.r_host = static_cast<unsigned long long*>(d_res); q.r_words = (b->want_chi2 ? R.bytes : R.bytes_no_chi2) / 8;
    q.want_chi2 = b->want_chi2 ? 1 : 0;
    q.r_poses = reinterpret_cast<float*>(rb + R.poses); q.r_state = reinterpret_cast<double*>(rb + R.state); q.r_points = reinterpret_cast<float*>(rb + R.points);
    q.r_chi2 = reinterpret_cast<double*>(rb + R.chi2); q.r_bad = rb + R.bad;
    q.done_ctr = reinterpret_cast<unsigned*>(b->dscratch.as<char>() + 768);
    b->p_lds = pl.lds; b->p_nf = pl.NF;
    b->persist = true; b->wide = false; b->fast = true;
    b->have_problem = true;
    return UH_OK;
}

// The problem that sits in the staging block (either record format), handed to the launch chain / wide form as the arrays it wants.
static int staged_problem_to_tables(uh_ba* b, int K, int P, int E, bool obs16) {
    const StageLayout& L = b->slay;
    std::vector<int32_t> op(E), of(E);
    std::vector<float> uv(2 * (size_t)E);
    std::vector<double> w(E);
    if (obs16) {
        const unsigned* r = reinterpret_cast<const unsigned*>(b->h_stage + L.obs);
        for (int e = 0; e < E; e++) {
            float f[3];
            std::memcpy(f, r + 4 * (size_t)e + 1, sizeof(f));
            op[e] = (int32_t)(r[4 * (size_t)e] & 0xFFFFFFu); of[e] = (int32_t)(r[4 * (size_t)e] >> 24);
            uv[2 * e] = f[0]; uv[2 * e + 1] = f[1]; w[e] = f[2];
        }
    } else {
        const uh_ba_obs* ob = reinterpret_cast<const uh_ba_obs*>(b->h_stage + L.obs);
        for (int e = 0; e < E; e++) { op[e] = ob[e].point; of[e] = ob[e].frame; uv[2 * e] = ob[e].u; uv[2 * e + 1] = ob[e].v; w[e] = ob[e].inv_sigma; }
    }
    uh_ba_problem pr{};
    pr.n_frames = K; pr.n_points = P; pr.n_obs = E;
    pr.poses_f2g = reinterpret_cast<const float*>(b->h_stage + L.poses_in); pr.fixed = b->h_stage + L.fixed; pr.intr = reinterpret_cast<const float*>(b->h_stage + L.intr_f);
    pr.points = reinterpret_cast<const float*>(b->h_stage + L.points);
    pr.obs_point = op.data(); pr.obs_frame = of.data(); pr.obs_uv = uv.data(); pr.obs_inv_sigma = w.data();
    return set_problem_tables(b, &pr);
}

static void set_problem_begin(uh_ba* b, const uh_ba_params* params) {
    b->have_problem = false;
    b->optimized = false;
    if (params) b->params = *params;
    if (b->params.huber_delta <= 0) b->params.huber_delta = std::sqrt(5.99);
    if (b->params.chi2_threshold <= 0) b->params.chi2_threshold = 5.99;
}

extern "C" {

// GlobalOptimizer::setParams: snapshot of everything the optimisation needs (the map may change afterwards)
int uh_ba_set_problem(uh_ba* b, const uh_ba_problem* pr, const uh_ba_params* params) {
    UH_REQUIRE(b && pr, "uh_ba_set_problem: NULL argument");
    set_problem_begin(b, params);
    const int K = pr->n_frames, P = pr->n_points, E = pr->n_obs;
    UH_REQUIRE(K >= 1 && P >= 0 && E >= 0, "uh_ba_set_problem: bad sizes K=%d P=%d E=%d", K, P, E);
    UH_REQUIRE(pr->poses_f2g && pr->fixed && pr->intr, "uh_ba_set_problem: NULL frame arrays");
    if (P > 0) UH_REQUIRE(pr->points, "uh_ba_set_problem: NULL points");
    if (E > 0) UH_REQUIRE(pr->obs_point && pr->obs_frame && pr->obs_uv && pr->obs_inv_sigma, "uh_ba_set_problem: NULL observation arrays");
    int nfree = 0;
    for (int k = 0; k < K; k++) nfree += pr->fixed[k] ? 0 : 1;
    const PersistPlan pl = plan_persistent(b, K, P, E, nfree);
    if (!pl.ok) return set_problem_tables(b, pr);
    int rc = ensure_staging(b, K, P, E);
    if (rc) return rc;
    const StageLayout& L = b->slay;
    std::memcpy(b->h_stage + L.poses_in, pr->poses_f2g, 16 * (size_t)K * sizeof(float));
    std::memcpy(b->h_stage + L.fixed, pr->fixed, K);
    std::memcpy(b->h_stage + L.intr_f, pr->intr, 4 * (size_t)K * sizeof(float));
    if (P) std::memcpy(b->h_stage + L.points, pr->points, 3 * (size_t)P * sizeof(float));
    uh_ba_obs* ob = reinterpret_cast<uh_ba_obs*>(b->h_stage + L.obs);
    // The ingest kernel's time is the bytes it fetches over the host link (~25 GB/s): when every information scalar is float-exact — the
    // reference's always are, (double)(float)(1. / scaleFactor) — and the indices fit 24 + 8 bits, the records go out as 16 bytes
    // {point | frame << 24, u, v, (float)inv_sigma} instead of 24.  Tried first; an inexact scalar falls through to the 24-byte form.
    const bool try16 = K <= 256 && P < (1 << 24) && !(getenv("UH_BA_OBS24") && atoi(getenv("UH_BA_OBS24")));
    // structure of arrays -> records [e0, e1) (e0 even), two observations per step with 128-bit moves (this loop is most of setParams' host
    // time: 26 000 observations, 31 us as scalar code, a third of that this way); indices checked on the way: an index is in range
    // iff neither i nor (n - 1 - i) is negative, the sign bits are OR-ed over the range
    const ObsPacker pack = [&](int e0, int e1, bool as16, bool& exact_out, unsigned& oob) {
        const __m128i pmax = _mm_set1_epi32(P - 1), kmax = _mm_set1_epi32(K - 1);
        __m128i bad = _mm_setzero_si128();
        int i = e0;
        if (as16) {
            static const bool has_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512dq");
            if (has_avx512 && !getenv("UH_BA_NO_AVX512"))   // (read per call: a test switches it within one process)
                i += pack_obs16_avx512(pr->obs_point, pr->obs_frame, pr->obs_uv, pr->obs_inv_sigma, e0, e1, P, K, reinterpret_cast<unsigned char*>(ob), exact_out, oob);
            __m128d exact = _mm_castsi128_pd(_mm_set1_epi32(-1));
            unsigned char* dst = reinterpret_cast<unsigned char*>(ob) + 16 * (size_t)i;
            for (; i + 2 <= e1; i += 2, dst += 32) {
                const __m128i pt = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(pr->obs_point + i));
                const __m128i kf = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(pr->obs_frame + i));
                const __m128 uv = _mm_loadu_ps(pr->obs_uv + 2 * i);
                const __m128d w = _mm_loadu_pd(pr->obs_inv_sigma + i);
                bad = _mm_or_si128(bad, _mm_or_si128(_mm_or_si128(pt, _mm_sub_epi32(pmax, pt)), _mm_or_si128(kf, _mm_sub_epi32(kmax, kf))));
                const __m128 wf = _mm_cvtpd_ps(w);                                                  // w0f w1f 0 0
                exact = _mm_and_pd(exact, _mm_cmpeq_pd(_mm_cvtps_pd(wf), w));
                const __m128i pk = _mm_or_si128(pt, _mm_slli_epi32(kf, 24));                          // p0 p1 0 0
                const __m128 a = _mm_castsi128_ps(_mm_unpacklo_epi32(pk, _mm_castps_si128(wf)));     // p0 w0 p1 w1
                const __m128i r0 = _mm_shuffle_epi32(_mm_castps_si128(_mm_shuffle_ps(a, uv, _MM_SHUFFLE(1, 0, 1, 0))), _MM_SHUFFLE(1, 3, 2, 0));   // p0 u0 v0 w0
                const __m128i r1 = _mm_shuffle_epi32(_mm_castps_si128(_mm_shuffle_ps(a, uv, _MM_SHUFFLE(3, 2, 3, 2))), _MM_SHUFFLE(1, 3, 2, 0));   // p1 u1 v1 w1
                _mm_storeu_si128(reinterpret_cast<__m128i*>(dst), r0);
                _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + 16), r1);
            }
            bool ok = _mm_movemask_pd(exact) == 3;
            oob |= (unsigned)_mm_movemask_ps(_mm_castsi128_ps(bad)) & 3u;
            for (; i < e1; i++, dst += 16) {
                const int pt = pr->obs_point[i], kf = pr->obs_frame[i];
                oob |= (unsigned)((unsigned)pt >= (unsigned)P) | (unsigned)((unsigned)kf >= (unsigned)K);
                const float wf = (float)pr->obs_inv_sigma[i];
                ok = ok && (double)wf == pr->obs_inv_sigma[i];
                const unsigned pk = ((unsigned)pt & 0xFFFFFFu) | ((unsigned)kf << 24);
                std::memcpy(dst, &pk, 4); std::memcpy(dst + 4, pr->obs_uv + 2 * i, 8); std::memcpy(dst + 12, &wf, 4);
            }
            if (!ok) exact_out = false;
            return;
        }
        unsigned char* dst = reinterpret_cast<unsigned char*>(ob) + 24 * (size_t)e0;
        for (; i + 2 <= e1; i += 2, dst += 48) {
            const __m128i pt = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(pr->obs_point + i));
            const __m128i kf = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(pr->obs_frame + i));
            const __m128i uv = _mm_loadu_si128(reinterpret_cast<const __m128i*>(pr->obs_uv + 2 * i));
            const __m128i w = _mm_loadu_si128(reinterpret_cast<const __m128i*>(pr->obs_inv_sigma + i));
            bad = _mm_or_si128(bad, _mm_or_si128(_mm_or_si128(pt, _mm_sub_epi32(pmax, pt)), _mm_or_si128(kf, _mm_sub_epi32(kmax, kf))));
            const __m128i pk = _mm_unpacklo_epi32(pt, kf);                                            // pt0 kf0 pt1 kf1
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst), _mm_unpacklo_epi64(pk, uv));           // pt0 kf0 u0 v0
            _mm_storel_epi64(reinterpret_cast<__m128i*>(dst + 16), w);                               // w0
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + 24), _mm_unpackhi_epi64(pk, uv));      // pt1 kf1 u1 v1
            _mm_storel_epi64(reinterpret_cast<__m128i*>(dst + 40), _mm_unpackhi_epi64(w, w));        // w1
        }
        oob |= (unsigned)_mm_movemask_ps(_mm_castsi128_ps(bad)) & 3u;   // (lanes 0, 1 hold the two observations; 2, 3 were zero-filled loads)
        for (; i < e1; i++) {
            const int pt = pr->obs_point[i], kf = pr->obs_frame[i];
            oob |= (unsigned)((unsigned)pt >= (unsigned)P) | (unsigned)((unsigned)kf >= (unsigned)K);
            ob[i].point = pt; ob[i].frame = kf; ob[i].u = pr->obs_uv[2 * i]; ob[i].v = pr->obs_uv[2 * i + 1]; ob[i].inv_sigma = pr->obs_inv_sigma[i];
        }
    };
    unsigned oob = 0;
    rc = set_problem_fast(b, K, P, E, pl, try16, &pack, &oob);
    if (oob)   // (an index out of range may not survive the 24 + 8 bit packing: found by the packer, named here)
        for (int e = 0; e < E; e++)
            UH_REQUIRE(pr->obs_point[e] >= 0 && pr->obs_point[e] < P && pr->obs_frame[e] >= 0 && pr->obs_frame[e] < K,
                       "uh_ba_set_problem: observation %d references point %d / frame %d out of range", e, pr->obs_point[e], pr->obs_frame[e]);
    return rc;
}

int uh_ba_map_staging(uh_ba* b, int n_frames, int n_points, int max_obs, uh_ba_staging* out) {
    UH_REQUIRE(b && out, "uh_ba_map_staging: NULL argument");
    UH_REQUIRE(n_frames >= 1 && n_points >= 0 && max_obs >= 0, "uh_ba_map_staging: bad sizes K=%d P=%d E=%d", n_frames, n_points, max_obs);
    UH_REQUIRE(b->job.load() == 0, "uh_ba_map_staging: an optimisation is in flight (call uh_ba_wait)");
    int rc = ensure_staging(b, n_frames, n_points, max_obs);
    if (rc) return rc;
    ++b->stage_gen;   // the caller may overwrite the block from here on
    const StageLayout& L = b->slay;
    out->poses_f2g = reinterpret_cast<float*>(b->h_stage + L.poses_in); out->fixed = b->h_stage + L.fixed; out->intr = reinterpret_cast<float*>(b->h_stage + L.intr_f);
    out->points = reinterpret_cast<float*>(b->h_stage + L.points); out->obs = reinterpret_cast<uh_ba_obs*>(b->h_stage + L.obs);
    out->cap_frames = b->cap_K; out->cap_points = b->cap_P; out->cap_obs = b->cap_E;
    return UH_OK;
}

int uh_ba_set_problem_staged(uh_ba* b, int K, int P, int E, const uh_ba_params* params) {
    UH_REQUIRE(b, "uh_ba_set_problem_staged: NULL argument");
    UH_REQUIRE(b->h_stage, "uh_ba_set_problem_staged: no staging block (call uh_ba_map_staging first)");
    UH_REQUIRE(K >= 1 && P >= 0 && E >= 0 && K <= b->cap_K && P <= b->cap_P && E <= b->cap_E,
               "uh_ba_set_problem_staged: sizes K=%d P=%d E=%d exceed the mapped capacities %d / %d / %d", K, P, E, b->cap_K, b->cap_P, b->cap_E);
    set_problem_begin(b, params);
    const StageLayout& L = b->slay;
    const uh_ba_obs* ob = reinterpret_cast<const uh_ba_obs*>(b->h_stage + L.obs);
    const unsigned char* fixed = b->h_stage + L.fixed;
    unsigned oob = 0;
    for (int e = 0; e < E; e++) oob |= (unsigned)((unsigned)ob[e].point >= (unsigned)P) | (unsigned)((unsigned)ob[e].frame >= (unsigned)K);
    if (oob)
        for (int e = 0; e < E; e++)
            UH_REQUIRE(ob[e].point >= 0 && ob[e].point < P && ob[e].frame >= 0 && ob[e].frame < K,
                       "uh_ba_set_problem: observation %d references point %d / frame %d out of range", e, ob[e].point, ob[e].frame);
    int nfree = 0;
    for (int k = 0; k < K; k++) nfree += fixed[k] ? 0 : 1;
    const PersistPlan pl = plan_persistent(b, K, P, E, nfree);
    if (pl.ok) return set_problem_fast(b, K, P, E, pl);
    // a window the persistent form does not take: hand the launch chain / wide form the arrays it wants
    return staged_problem_to_tables(b, K, P, E, false);
}

// the per-observation chi2 is an extra of this ABI (the reference's getResults does not return it): a host that never asks for it saves
// the kernel three quarters of its result hand-over.  Takes effect with the next set_problem.
int uh_ba_want_chi2(uh_ba* b, int on) {
    UH_REQUIRE(b, "uh_ba_want_chi2: NULL");
    b->want_chi2 = on != 0;
    return UH_OK;
}

int uh_ba_form(uh_ba* b, int* lanes_out) {
    UH_REQUIRE(b && b->have_problem, "uh_ba_form: no problem set");
    if (lanes_out) *lanes_out = b->persist ? b->p_nf : 0;
    return b->wide ? 2 : (b->persist ? 1 : 0);
}

// GlobalOptimizer::optimize(bool* stopASAP).  The caller may flip *stop_asap asynchronously; it is sampled into a pinned,
// device-visible flag before each pass and whenever the host waits, and the decide kernel reads that flag every trial.
int uh_ba_optimize(uh_ba* b, const volatile uint8_t* stop_asap) {
    UH_REQUIRE(b && b->have_problem, "uh_ba_optimize: no problem set (call uh_ba_set_problem first)");
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    const BADims& d = b->dims;
    if (b->h_stop) *b->h_stop = (stop_asap && *stop_asap) ? 1 : 0;
    const int nmax = std::max(std::max(d.K, 3 * d.P), std::max(d.E, 1));
    const int n1 = b->params.n_iters, n2 = 2 * b->params.n_iters;
    const float mc = b->params.min_chi2_between_iter;
    b->iters[0] = b->iters[1] = 0;
    b->step = 0;
    if (b->persist) {
        const int rcp = run_persistent(b, stop_asap, n1, n2, mc);
        if (rcp != UH_ENODEVICE || !b->h_stage || !b->persist_not_resident) return rcp;
        if (b->problem_stage_gen != b->stage_gen) {   // uh_ba_map_staging was called since set_problem: the block may hold the NEXT keyframe's problem
            const std::string why0 = uh_last_error();
            uh::set_error("%s; the staging block has been remapped since set_problem, so the launch-chain fallback cannot rebuild this problem: set it again", why0.c_str());
            return UH_ENODEVICE;
        }
        // The persistent kernel's workgroups did not all become resident within its timeout (another process's or library's spinning
        // kernel holds CUs: INTEGRATION.md section 8).  The problem is still in the staging block: this optimisation and the next 64
        // problems of this object take the launch chain, which needs no co-residency; then the persistent form is tried again.
        const std::string why = uh_last_error();
        const int rc2 = staged_problem_to_tables(b, d.K, d.P, d.E, b->pq.obs16 != 0);   // (clears b->persist; the parameters stay)
        b->persist_blocked = 64;
        if (rc2 != UH_OK) { uh::set_error("%s; falling back to the launch chain failed too", why.c_str()); return UH_ENODEVICE; }
        return uh_ba_optimize(b, stop_asap);
    }
    UH_LAUNCH(b->ctx,ba_init_state_kernel, dim3(uh_div_up(nmax, 256)), dim3(256), 0, b->ptrs, d, b->d_pose0, b->d_pts0);
    // Both passes are enqueued in one go, one step per outer iteration: enough when no trial is rejected — a rejected trial is rare,
    // and a spare step whose pass is already finished would still cost its two launches (~10 us per pass); finish_pass() adds steps
    // when a pass needs them.  Between them the
    // gate kernel checks on the device that pass 1 is complete and not stopped; if it is not, relabel / begin_pass do nothing
    // and the steps enqueued for pass 2 simply continue pass 1.  The host looks at the state once, at the end.
    int rc;
    UH_LAUNCH(b->ctx,ba_begin_pass_kernel, dim3(1), dim3(64), 0, b->ptrs, n1, mc, b->step & 1, 0);
    if ((rc = enqueue_steps(b, n1, true))) return rc;
    UH_LAUNCH(b->ctx,ba_gate_kernel, dim3(1), dim3(64), 0, b->ptrs, b->step & 1);
    if (d.E > 0) UH_LAUNCH(b->ctx,ba_relabel_kernel, dim3(uh_div_up(d.E, 256)), dim3(256), 0, b->ptrs, d, b->step & 1, 1);
    UH_LAUNCH(b->ctx,ba_begin_pass_kernel, dim3(1), dim3(64), 0, b->ptrs, n2, mc, b->step & 1, 1);
    if ((rc = enqueue_steps(b, n2, true))) return rc;
    BAState hs;
    if ((rc = wait_state(b, &hs, stop_asap))) return rc;
    if (hs.gate) {                       // the common case: pass 2 is under way or done
        b->iters[0] = hs.iters_pass1;
        if ((rc = finish_pass(b, &hs, stop_asap))) return rc;
        b->iters[1] = hs.iters_done;
    } else {                             // pass 1 needed more steps than its budget, or a stop was requested
        if ((rc = finish_pass(b, &hs, stop_asap))) return rc;
        b->iters[0] = hs.iters_done;
        bool cont = !hs.stopped;
        if (stop_asap && *stop_asap) cont = false;
        if (b->h_stop && *b->h_stop) cont = false;
        if (cont) {
            if (d.E > 0) UH_LAUNCH(b->ctx,ba_relabel_kernel, dim3(uh_div_up(d.E, 256)), dim3(256), 0, b->ptrs, d, b->step & 1, 0);
            UH_LAUNCH(b->ctx,ba_begin_pass_kernel, dim3(1), dim3(64), 0, b->ptrs, n2, mc, b->step & 1, 0);
            if ((rc = enqueue_steps(b, n2 + 1, true))) return rc;
            if ((rc = wait_state(b, &hs, stop_asap))) return rc;
            if ((rc = finish_pass(b, &hs, stop_asap))) return rc;
            b->iters[1] = hs.iters_done;
        }
    }
    b->optimized = true;
    return UH_OK;
}

// measurement hook: the 64 phase timestamps (10 ns ticks) written by the last executed LM step
int uh_ba_debug_clocks(uh_ba* b, int64_t* out64) {
    UH_REQUIRE(b && b->have_problem && out64, "uh_ba_debug_clocks: not ready");
    UH_HIP_CHECK(hipMemcpyAsync(out64, b->ptrs.clk, 64 * sizeof(long long), hipMemcpyDeviceToHost, b->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));
    return UH_OK;
}

// Asynchronous form: the optimisation runs on a worker thread of the object (GlobalOptimizer::optimize is what the reference's
// mapper thread spends its time in); uh_ba_wait returns its result.  The caller's thread is free to enqueue tracking work on
// another stream in between.  One optimisation in flight per object.
static void start_worker(uh_ba* b) {   // (called with b->mu held)
    if (b->worker.joinable()) return;
    b->worker = std::thread([b]() {
        std::unique_lock<std::mutex> l(b->mu);
        for (;;) {
            // a mapper that is handed a keyframe every half millisecond: spin briefly for the next request before sleeping on the
            // condition variable (a futex wake-up is ~10-20 us on the hand-over in each direction, 4 % of a 0.5 ms optimisation)
            l.unlock();
            spin_until([b]() { const int j = b->job.load(std::memory_order_acquire); return j == 1 || j == -1; }, 300);
            l.lock();
            b->cv.wait(l, [b]() { return b->job == 1 || b->job == -1; });
            if (b->job == -1) return;
            b->job = 2;
            const volatile uint8_t* stop = b->job_stop;
            const int kind = b->job_kind;
            l.unlock();
            int rc = UH_OK;
            if (kind == 1) {   // setParams first (mapmanager.cpp:11388-11405: both calls run back to back on the mapper thread)
                const uh_ba_params* ps = b->job_has_params ? &b->job_params : nullptr;
                rc = b->job_problem ? uh_ba_set_problem(b, b->job_problem, ps)
                                    : uh_ba_set_problem_staged(b, b->job_dims[0], b->job_dims[1], b->job_dims[2], ps);
            }
            if (rc == UH_OK) rc = uh_ba_optimize(b, stop);
            const std::string err = rc ? std::string(uh_last_error()) : std::string();   // thread-local: carry it over
            l.lock();
            b->job_rc = rc;
            b->job_err = err;
            b->job = 3;
            b->cv.notify_all();
        }
    });
}

int uh_ba_optimize_async(uh_ba* b, const volatile uint8_t* stop_asap) {
    UH_REQUIRE(b && b->have_problem, "uh_ba_optimize_async: no problem set (call uh_ba_set_problem first)");
    std::unique_lock<std::mutex> lk(b->mu);
    UH_REQUIRE(b->job == 0, "uh_ba_optimize_async: an optimisation is already in flight (call uh_ba_wait)");
    start_worker(b);
    b->job_stop = stop_asap;
    b->job_kind = 0;
    b->job = 1;
    lk.unlock();
    b->cv.notify_all();
    return UH_OK;
}

int uh_ba_solve_async(uh_ba* b, const uh_ba_problem* problem, int n_frames, int n_points, int n_obs, const uh_ba_params* params,
                      const volatile uint8_t* stop_asap) {
    UH_REQUIRE(b, "uh_ba_solve_async: NULL argument");
    std::unique_lock<std::mutex> lk(b->mu);
    UH_REQUIRE(b->job == 0, "uh_ba_solve_async: an optimisation is already in flight (call uh_ba_wait)");
    start_worker(b);
    if (problem) { b->job_problem_copy = *problem; b->job_problem = &b->job_problem_copy; }   // (the struct is copied, the arrays are the caller's)
    else { b->job_problem = nullptr; b->job_dims[0] = n_frames; b->job_dims[1] = n_points; b->job_dims[2] = n_obs; }
    b->job_has_params = params != nullptr;
    if (params) b->job_params = *params;
    b->job_stop = stop_asap;
    b->job_kind = 1;
    b->job = 1;
    lk.unlock();
    b->cv.notify_all();
    return UH_OK;
}

int uh_ba_wait(uh_ba* b) {
    UH_REQUIRE(b, "uh_ba_wait: NULL");
    UH_REQUIRE(b->job.load() != 0, "uh_ba_wait: nothing in flight");
    spin_until([b]() { return b->job.load(std::memory_order_acquire) == 3; }, 2000);   // (an optimisation is a fraction of a millisecond: usually no sleep at all)
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv.wait(lk, [b]() { return b->job == 3; });
    const int rc = b->job_rc;
    b->job = 0;
    if (rc) uh::set_error("%s", b->job_err.c_str());
    return rc;
}

uint8_t* uh_ba_stop_flag(uh_ba* b) { return b ? b->h_stop : nullptr; }

int uh_ba_get_results(uh_ba* b, float* poses_out, float* points_out, double* chi2_out, uint8_t* bad_out, int32_t* iters_out) {
    UH_REQUIRE(b && b->have_problem && b->optimized, "uh_ba_get_results: optimize() has not run");
    if (b->fast) {   // the optimisation kernel's tail has already written everything into the pinned result block
        const BADims& d = b->dims;
        const ResLayout& R = b->rlay;
        if (poses_out) std::memcpy(poses_out, b->h_res + R.poses, 16 * (size_t)d.K * sizeof(float));
        if (points_out && d.P) std::memcpy(points_out, b->h_res + R.points, 3 * (size_t)d.P * sizeof(float));
        UH_REQUIRE(!(chi2_out && d.E) || b->pq.want_chi2, "uh_ba_get_results: chi2 was switched off for this problem (uh_ba_want_chi2)");
        if (chi2_out && d.E) std::memcpy(chi2_out, b->h_res + R.chi2, (size_t)d.E * sizeof(double));
        if (bad_out && d.E) std::memcpy(bad_out, b->h_res + R.bad, (size_t)d.E);
        if (iters_out) { iters_out[0] = b->iters[0]; iters_out[1] = b->iters[1]; }
        return UH_OK;
    }
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const BADims& d = b->dims;
    const int n0 = std::max(d.K, 3 * d.P);
    UH_LAUNCH(b->ctx,ba_results_kernel, dim3(uh_div_up(std::max(n0, 1), 256)), dim3(256), 0, b->ptrs, d, b->d_poses_in.as<float>(),
                       b->d_poses_out.as<float>(), b->d_points_out.as<float>(), b->d_bad.as<unsigned char>(), 0, b->step & 1);
    if (d.E > 0)
        UH_LAUNCH(b->ctx,ba_results_kernel, dim3(uh_div_up(d.E, 256)), dim3(256), 0, b->ptrs, d, b->d_poses_in.as<float>(),
                           b->d_poses_out.as<float>(), b->d_points_out.as<float>(), b->d_bad.as<unsigned char>(), 1, b->step & 1);
    // through ONE pinned block: asynchronous DMA + one synchronisation (four pageable copies each staged and synchronised on their own: 0.16 ms)
    const size_t o_po = 0, o_pt = o_po + ((16 * (size_t)d.K * 4 + 255) & ~(size_t)255), o_bad = o_pt + ((3 * (size_t)d.P * 4 + 255) & ~(size_t)255);
    const size_t o_chi = o_bad + (((size_t)d.E + 255) & ~(size_t)255), total = o_chi + (chi2_out ? (size_t)d.E * 8 : 0) + 256;
    int rc = b->res_pin.reserve(total);
    if (rc) return rc;
    unsigned char* hp = b->res_pin.as<unsigned char>();
    if (poses_out) UH_HIP_CHECK(hipMemcpyAsync(hp + o_po, b->d_poses_out.p, 16 * (size_t)d.K * 4, hipMemcpyDeviceToHost, st));
    if (points_out && d.P) UH_HIP_CHECK(hipMemcpyAsync(hp + o_pt, b->d_points_out.p, 3 * (size_t)d.P * 4, hipMemcpyDeviceToHost, st));
    if (chi2_out && d.E) UH_HIP_CHECK(hipMemcpyAsync(hp + o_chi, b->ptrs.e_chi2, (size_t)d.E * 8, hipMemcpyDeviceToHost, st));
    if (bad_out && d.E) UH_HIP_CHECK(hipMemcpyAsync(hp + o_bad, b->d_bad.p, (size_t)d.E, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    if (poses_out) std::memcpy(poses_out, hp + o_po, 16 * (size_t)d.K * 4);
    if (points_out && d.P) std::memcpy(points_out, hp + o_pt, 3 * (size_t)d.P * 4);
    if (chi2_out && d.E) std::memcpy(chi2_out, hp + o_chi, (size_t)d.E * 8);
    if (bad_out && d.E) std::memcpy(bad_out, hp + o_bad, (size_t)d.E);
    if (iters_out) { iters_out[0] = b->iters[0]; iters_out[1] = b->iters[1]; }
    return UH_OK;
}

// final pose state (qx qy qz qw tx ty tz per frame, fp64) — used by the parity tests to state the tolerance on se3
int uh_ba_get_pose_state(uh_ba* b, double* pose7_out) {
    UH_REQUIRE(b && b->have_problem && b->optimized && pose7_out, "uh_ba_get_pose_state: not ready");
    if (b->fast) { std::memcpy(pose7_out, b->h_res + b->rlay.state, 7 * (size_t)b->dims.K * sizeof(double)); return UH_OK; }
    BAState hs;
    UH_HIP_CHECK(hipMemcpyAsync(&hs, b->ptrs.st + (b->step & 1), sizeof(BAState), hipMemcpyDeviceToHost, b->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));
    UH_HIP_CHECK(hipMemcpyAsync(pose7_out, b->ptrs.pose[hs.cur], 7 * (size_t)b->dims.K * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));
    return UH_OK;
}

// getResults in place
int uh_ba_results_view_get(uh_ba* b, uh_ba_results_view* out) {
    UH_REQUIRE(b && out, "uh_ba_results_view_get: NULL argument");
    UH_REQUIRE(b->have_problem && b->optimized, "uh_ba_results_view_get: optimize() has not run");
    UH_REQUIRE(b->fast, "uh_ba_results_view_get: the current problem runs in a form that keeps its results in HBM (use uh_ba_get_results)");
    const ResLayout& R = b->rlay;
    out->poses = reinterpret_cast<const float*>(b->h_res + R.poses); out->points = reinterpret_cast<const float*>(b->h_res + R.points);
    out->chi2 = b->pq.want_chi2 ? reinterpret_cast<const double*>(b->h_res + R.chi2) : nullptr; out->bad = b->h_res + R.bad;
    out->pose_state = reinterpret_cast<const double*>(b->h_res + R.state);
    out->iters[0] = b->iters[0]; out->iters[1] = b->iters[1];
    out->n_frames = b->dims.K; out->n_points = b->dims.P; out->n_obs = b->dims.E;
    return UH_OK;
}

}  // extern "C"
