// Persistent form of the local bundle adjustment (included by ba.hip inside its anonymous namespace).
//
// ONE launch runs the whole GlobalOptimizerG2O::optimize (globaloptimizer_g2o.cpp:418-464): both Levenberg passes, the outlier
// relabelling between them and every LM trial (g2o/core/optimization_algorithm_levenberg.cpp:58-150) — where the legacy form
// needs two launches per trial.  G <= 256 co-resident workgroups of 256 threads; workgroup g OWNS the <= 32 landmarks
// [g*Lw, (g+1)*Lw) for the whole optimisation:
//   * landmark coordinates (current + trial), the observations' constants, activity flags and last chi2 live in REGISTERS
//     (one lane per (landmark, free camera): 8 lanes per landmark), the free poses in LDS (every workgroup keeps its own,
//     bit-identical copy) — nothing of the estimate is re-read from HBM between trials;
//   * the linearisation is never materialised: a trial recomputes the 2x3 / 2x6 Jacobians (typesg2o.h:275-314, ~150 flop) from
//     the 24-byte observation, forms Hll, its Cholesky factor L (3x3) and the whitened blocks Y_e = Hpl_e L^-T straight into an
//     LDS panel Yt[3*landmark + k][6*camera + a];
//   * Schur complement (g2o/core/block_solver.hpp:341-392): sum_l Hpl D^-1 Hpl^T = Yt^T Yt, a dense 48 x 48 x (3*Lw) fp64 product
//     per workgroup — vector FMA on 4x4 register blocks (v_mfma_f64_16x16x4_f64 on the six upper 16x16 tiles was measured in rounds
//     2-4 and removed in round 5: 6.4 FMA / clock / SIMD against 11.4 for v_fma_f64 on gfx950);  b_schur = Yt^T (L^-1 b_l);
//   * back-substitution (block_solver.hpp:419-442): dx_l = L^-T (L^-1 b_l - Y_l^T dx_p) from the same panel;
//   * workgroups exchange only reduction partials, through write-through (sc1) stores of SELF-VALIDATING words: every double travels
//     as two 64-bit words (32 payload bits | 32-bit tag = launch sequence and exchange round), so a reader polls the data itself —
//     one memory round trip (~1.5 us on MI355X) per exchange instead of drain + flag + flag poll + data load:
//       A: every workgroup's partial (product with Hpp folded in, bp, b_schur - bp, chi2, max diag)  -> slice-wise reduction by all
//          workgroups (lambda goes onto the diagonal there)
//       B: the reduced vector -> EVERY workgroup scatters it into the bordered system and factorises it itself (no broadcast hop)
//       C: trial chi2 / scale partials -> every workgroup takes the accept / reject decision itself (apply_decision) — only for a
//          trial that may be its pass's last: any other trial is SPECULATIVE (see the trial loop): it linearises at the trial
//          estimate at once, runs phase 1 with lambda / 3 and lets its chi2 ride in the next trial's partial; the decision is taken
//          when B arrives
//     (each of them ONE batch of L2-bypassing loads per thread = one memory round trip)
//     so a trial costs two data hand-offs instead of two kernel boundaries + launch ramps, and all summation orders are
//     fixed: results are run-to-run deterministic and identical in every workgroup.
#pragma once

// The translation unit is compiled with -ffp-contract=off (the ORB stage reproduces the reference's un-contracted float arithmetic bit for
// bit).  The BA's parity bar is a tolerance (1e-6 on the state, identical iteration counts), not bits: from here to the end of the unit
// (this header is included last; only host code follows) a * b + c may fuse.  The edge evaluation and the 61 sums of a linearisation
// were 738 v_mul_f64 + 576 v_add_f64 beside 426 v_fma_f64 on a kernel whose every instruction sits on a lone wave's critical path.
// (apply_decision, the LDL^T and the back substitution are defined before this point and keep the unit's setting.)
#pragma clang fp contract(fast)

constexpr int kPThreads = 256;   // 4 waves, one per SIMD: each may use the whole 512-entry register file (256 VGPR + 256 AGPR)
constexpr int kPWaves = kPThreads / 64;
// How long a workgroup waits for one that never arrives (another tenant's kernels hold the CUs) before the launch gives up and
// uh_ba_optimize takes the launch chain: 100 ms by default (UH_BA_RESIDENCY_TIMEOUT_MS) — a hand-off normally takes microseconds and the
// whole launch half a millisecond; round 3 waited 3 s.
constexpr long long kPTimeoutTicksDefault = 10000000ll;   // 100 ms of the 100 MHz wall clock

struct BAPersist {
    int G, Lw, krows, SL, nelem, max_fix, kfix;
    int nb4, nblk, KS;   // the product's 4x4 block grid: nb4 = ceil(n / 4) block columns, nblk upper blocks, KS splits of the K range
    int n1, n2, stop_at_begin;
    int speculate;        // 1: speculative trials (see the trial loop); UH_BA_SPEC=0 keeps the three-hand-off form for A/B measurements
    unsigned launch_id;   // tags the error / completion words of this launch
    long long timeout_ticks;   // see kPTimeoutTicksDefault
    unsigned tag_base;    // (launch sequence of this optimizer & 0xFFFFF) << 12: the upper bits of every exchanged word's tag.  The exchange
                          // buffers are zeroed whenever they are (re)allocated and whenever the sequence wraps, so a stale word never matches.
    float minChi2;
    // the problem as uh_ba_set_problem left it in HBM: ONE H2D copy of the staging block + ba_ingest_kernel (ba.hip) — no table is built
    // on the host.  T[point * K + frame] = (problem sequence << 20) | (observation index + 1): a cell of an older problem never matches.
    const unsigned* T; unsigned tseq;
    const uh_ba_obs* obs;            // E x {point, frame, u, v, inv_sigma} (24 bytes, include/ucoslam_hip.h) — or, obs16 != 0, E x 16 bytes
    int obs16;                       // {point | frame << 24, u, v, (float)inv_sigma}: what uh_ba_set_problem writes when every scalar is float-exact
    const float* points;             // P x 3 float (MapPoint::getCoordinates); widened to double here exactly as setParams does
    const float* poses_in;           // K x 16 float: fixed frames are returned unchanged, and their rows test the depth of bad associations
    const int* fix_kf;   // [kfix] frame index of EVERY fixed frame, ascending
    const double* pose0; const double* poseR0;                           // K x 7, K x 12: the snapshot taken by setParams
    // results (globaloptimizer_g2o.cpp:466-537), written by the kernel's tail into a block in HBM (these pointers), then copied to pinned host memory
    float* r_poses; double* r_state; float* r_points; unsigned char* r_bad; double* r_chi2;
    int want_chi2;                   // 0: the per-observation chi2 stays on the device (uh_ba_want_chi2)
    unsigned long long* r_dev; unsigned long long* r_host; size_t r_words;   // the result block in HBM, its pinned host twin, 8-byte words in use
    unsigned* done_ctr; unsigned done_base;     // counts the workgroups through the two steps of the result hand-over (base: its value before this launch)
    unsigned long long* part;        // [slice][workgroup][SL] tagged doubles (two words each)
    unsigned long long* red;         // [G * SL]
    unsigned long long* partC;       // [G][4]: chi2, scale, (workgroup 0: stop flag), -
    BAState* host_state;             // pinned, device-visible host memory: the final LM state ...
    unsigned long long* host_done;   // ... and (launch id << 32 | 1 = finished, 2 = a workgroup never arrived), written last: the host polls this word
    unsigned long long* errw;    // error word (launch id << 32 | 1): some workgroup gave up waiting
};

constexpr int kFxCam = 20;   // doubles per fixed frame in LDS: R | t (12), fx fy cx cy (4), row 2 of the INPUT float pose widened to double (4: the depth test of getResults)
struct PersistLds {      // offsets in doubles into the dynamic LDS block
    int Yt, U, usz, out, wsum, x, bp, bs, bsp, wv, pose, poseR, red, sc, fxchi, fxobs, fxcam, fxact_bytes, fxk_bytes, fxid_bytes, fxptr_bytes, pair_bytes, blk_bytes, flag_bytes, total_bytes;
};
// off_cam = elements of the product part of a partial (nblk * 16, or the six MFMA tiles), KS = K-splits of the product, SL = slice length
template <int NF>
__host__ __device__ inline PersistLds persist_lds(int krows, int n, int max_fix, int kfix, int off_cam, int KS, int SL) {
    constexpr int NP = 6 * NF, YS = NP + 2;
    PersistLds o;
    int a = 0;
    o.Yt = a; a += krows * YS;
    o.U = a;
    o.usz = (n + 1) * (n + 1) + (n + 1 <= 64 ? 2 * 121 * 6 : 2 * 6 * 128 + 4);   // reduced system + the factorisation's panel buffer
    if (o.usz < KS * off_cam) o.usz = KS * off_cam;                               // the product's K-split partials
    { const int r = KS * off_cam + (SL > 256 ? SL : 256) + 8; if (o.usz < r) o.usz = r; }   // the product's K-split partials (added on the way out) + the slice reduction's scratch behind them
    if (o.usz < kPWaves * NF * 33) o.usz = kPWaves * NF * 33;
    if (o.usz < 2048) o.usz = 2048;
    a += o.usz;
    o.out = a; a += NF * 27 + NP + 4;
    o.wsum = a; a += kPWaves * NF * 33;   // the waves' camera-side sums (outside U: the product may overwrite U while they are being added)
    o.x = a; a += NP; o.bp = a; a += NP; o.bs = a; a += NP;
    o.bsp = a; a += kPWaves * NP;
    o.wv = a; a += krows;
    o.pose = a; a += 2 * NF * 7; o.poseR = a; a += 2 * NF * 12;
    o.red = a; a += 16; o.sc = a; a += 8;
    o.fxchi = a; a += max_fix;
    o.fxobs = a; a += 3 * max_fix;
    o.fxcam = a; a += kFxCam * kfix;
    o.fxact_bytes = a * 8;
    int b = o.fxact_bytes + ((max_fix + 15) & ~15);
    o.fxk_bytes = b; b += (max_fix + 15) & ~15;
    o.fxid_bytes = b; b += 4 * ((max_fix + 3) & ~3);
    o.fxptr_bytes = b; b += 4 * (kPThreads / NF + 4);
    o.pair_bytes = b; b += NF * (NF + 1) / 2 * 4;
    b = (b + 15) & ~15;
    o.blk_bytes = b; b += ((NP / 4) * (NP / 4 + 1) / 2 * 2 + 15) & ~15;
    o.flag_bytes = b; b += 16;
    o.total_bytes = b;
    return o;
}

// Tagged exchange words.  Element i of a buffer = words 2i (low 32 payload bits) and 2i+1 (high 32 bits), each with the round's tag in
// its upper half; both are single-copy-atomic 64-bit write-through stores / L2-bypassing loads, so a word whose tag matches IS the
// datum: no flag, no store drain, no ordering between words is needed.
typedef unsigned long long pword;
struct TWord { pword lo, hi; };
// The two words of an element are adjacent and 16-byte aligned: ONE 16-byte write-through store carries both (each 8-byte half is still
// written whole, and a reader still validates each half by its own tag) — half the store instructions and half the fabric writes of
// two 8-byte stores (the exchange was 80 MB of 8-byte fabric writes per optimisation).
__device__ __forceinline__ void tst(pword* base, size_t i, double v, unsigned tag) {
    const pword b = (pword)__double_as_longlong(v);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 w = {(unsigned)b, tag, (unsigned)(b >> 32), tag};
    // (a buffer store through the builtin, not inline asm: the compiler must know this is a memory instruction that reads its data
    // registers after issue — an asm store got its operands overwritten in flight; aux 16 = sc1, write-through)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)(i * 16), 0, 16);
}
__device__ __forceinline__ TWord tld_raw(const pword* base, size_t i) {
    TWord w;
    w.lo = __hip_atomic_load(base + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.hi = __hip_atomic_load(base + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w;
}
__device__ __forceinline__ bool tok(const TWord& w, unsigned tag) { return (unsigned)(w.lo >> 32) == tag && (unsigned)(w.hi >> 32) == tag; }
__device__ __forceinline__ double tval(const TWord& w) { return __longlong_as_double((long long)((w.hi << 32) | (w.lo & 0xffffffffull))); }

// 1/sqrt(x) to the last bit or two: v_rsq_f64 (~2^-23) + two Newton steps; x <= 0 / NaN gives NaN / inf, which the reduced
// system's factorisation then reports as a failed solve (g2o: a singular D^-1 poisons Hschur and LDLT reports failure)
__device__ __forceinline__ double rsqrt_nr(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x, r * r, 1.5);
    r = r * fma(-0.5 * x, r * r, 1.5);
    return r;
}

// Edge evaluation of the persistent form: EdgeSE3ProjectXYZ::computeError / linearizeOplus (typesg2o.h:260-314) with ONE reciprocal
// (v_rcp_f64 + two Newton steps, ~1 ulp) where the reference divides nine times, and the Huber weight through one reciprocal
// square root: the nine IEEE division sequences (~40 dependent instructions each) were 60 % of the linearisation phase.  The results
// differ from the division form in the last bit or two (state after a whole BA: ~1e-15, against a stated tolerance of 1e-6).
// JAC: 0 = error only, 1 = error + point Jacobian A (fixed cameras), 2 = error + A + pose Jacobian B.
template <int JAC>
__device__ __forceinline__ void edge_eval_p(double u, double v, double w, double fx, double fy, double cx, double cy, double delta, double dsqr,
                                            const double* Rt, const double* X, bool robust, EdgeLin& o) {
    const double x = Rt[0] * X[0] + Rt[1] * X[1] + Rt[2] * X[2] + Rt[9];
    const double y = Rt[3] * X[0] + Rt[4] * X[1] + Rt[5] * X[2] + Rt[10];
    const double z = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
    const double iz = fast_rcp(z);
    const double xz = x * iz, yz = y * iz;
    o.ex = u - (xz * fx + cx);
    o.ey = v - (yz * fy + cy);
    o.chi2 = w * (o.ex * o.ex + o.ey * o.ey);
    o.rho1 = 1.0;
    o.robchi = o.chi2;
    if (robust && o.chi2 > dsqr) {
        const double rs = rsqrt_nr(o.chi2);
        o.rho1 = delta * rs;
        o.robchi = 2 * (o.chi2 * rs) * delta - dsqr;
    }
    if (JAC >= 1) {
        o.ww = o.rho1 * w;
        o.r0 = -w * o.ex * o.rho1;
        o.r1 = -w * o.ey * o.rho1;
        const double fxz = fx * iz, fyz = fy * iz;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o.A[c] = -fxz * (Rt[c] - xz * Rt[6 + c]);
            o.A[3 + c] = -fyz * (Rt[3 + c] - yz * Rt[6 + c]);
        }
        if (JAC >= 2) {
            o.B[0] = xz * yz * fx; o.B[1] = -(1 + xz * xz) * fx; o.B[2] = yz * fx; o.B[3] = -fxz; o.B[4] = 0; o.B[5] = xz * fxz;
            o.B[6] = (1 + yz * yz) * fy; o.B[7] = -xz * yz * fy; o.B[8] = -xz * fy; o.B[9] = 0; o.B[10] = -fyz; o.B[11] = yz * fyz;
        }
    }
}

// T_trial = exp(dx) T_cur (SE3Quat::exp and the quaternion product of VertexSE3Expmap::oplusImpl) for the persistent kernel: the same
// formulas as se3_left_update (ba.hip), with every IEEE division and square root replaced by rsqrt_nr / fast_rcp and multiplications
// (~13 divisions + 3 square roots of ~35 dependent instructions each sit on ONE lone wave per workgroup here: the pose update was
// most of the 2.2 us between the back substitution and the trial errors).  Last-bit differences against the division form.
__device__ __forceinline__ void quat_norm_pos_p(double (&q)[4]) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double rn = rsqrt_nr(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= rn; q[1] *= rn; q[2] *= rn; q[3] *= rn;
}
__device__ __forceinline__ void quat_from_R_p(const double (&R)[9], double (&q)[4]) {   // Eigen::Quaternion(Matrix3), 1/sqrt form
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double x = tr + 1.0, r = rsqrt_nr(x), h = 0.5 * r;   // sqrt(x) = x r, 0.5 / sqrt(x) = 0.5 r
        q[3] = 0.5 * (x * r);
        q[0] = (R[7] - R[5]) * h; q[1] = (R[2] - R[6]) * h; q[2] = (R[3] - R[1]) * h;
    } else if (!(R[4] > R[0]) && !(R[8] > R[0])) {
        const double x = R[0] - R[4] - R[8] + 1.0, r = rsqrt_nr(x), h = 0.5 * r;
        q[0] = 0.5 * (x * r);
        q[3] = (R[7] - R[5]) * h; q[1] = (R[3] + R[1]) * h; q[2] = (R[6] + R[2]) * h;
    } else if (R[4] > R[0] && !(R[8] > R[4])) {
        const double x = R[4] - R[8] - R[0] + 1.0, r = rsqrt_nr(x), h = 0.5 * r;
        q[1] = 0.5 * (x * r);
        q[3] = (R[2] - R[6]) * h; q[2] = (R[7] + R[5]) * h; q[0] = (R[1] + R[3]) * h;
    } else {
        const double x = R[8] - R[0] - R[4] + 1.0, r = rsqrt_nr(x), h = 0.5 * r;
        q[2] = 0.5 * (x * r);
        q[3] = (R[3] - R[1]) * h; q[0] = (R[2] + R[6]) * h; q[1] = (R[5] + R[7]) * h;
    }
}
__device__ __forceinline__ void se3_left_update_p(double (&q)[4], double (&t)[3], const double* dx) {
    const double w0 = dx[0], w1 = dx[1], w2 = dx[2];
    const double th2 = w0 * w0 + w1 * w1 + w2 * w2;
    const double O[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double O2[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) O2[r * 3 + c] = O[r * 3] * O[c] + O[r * 3 + 1] * O[3 + c] + O[r * 3 + 2] * O[6 + c];
    double a, b, c2;
    if (!(th2 >= 1e-10)) { a = 1; b = 0.5; c2 = 1.0 / 6.0; }   // theta < 1e-5 (also th2 == 0, where 1/theta does not exist)
    else {
        const double rth = rsqrt_nr(th2), theta = th2 * rth, rth2 = rth * rth;
        double sn, cs;
        sincos(theta, &sn, &cs);   // (a Taylor branch for small angles beside this call made the phase slower: 1.88 -> 2.72 us)
        a = sn * rth; b = (1 - cs) * rth2; c2 = (theta - sn) * (rth2 * rth);
    }
    double Rm[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; Rm[i] = I + a * O[i] + b * O2[i]; V[i] = I + b * O[i] + c2 * O2[i]; }
    double qe[4];
    quat_from_R_p(Rm, qe);
    quat_norm_pos_p(qe);
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
    quat_norm_pos_p(qn);
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = qn[i];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = tn[i];
}


// (dpp_f64 / swap_rows / the kDpp* controls: reduce.hpp)
// sum over the groups of W consecutive lanes (W = 8, 16, 64), the same bits in every lane of a group
template <int W>
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppHalfMirror>(v);
    if (W >= 16) v += dpp_f64<kDppMirror>(v);
    if (W >= 32) { double a = v, b = v; swap_rows<16>(a, b); v = a + b; }
    if (W >= 64) { double a = v, b = v; swap_rows<32>(a, b); v = a + b; }
    return v;
}
template <int W>
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, dpp_f64<kDppXor1>(v));
    v = fmax(v, dpp_f64<kDppXor2>(v));
    v = fmax(v, dpp_f64<kDppHalfMirror>(v));
    if (W >= 16) v = fmax(v, dpp_f64<kDppMirror>(v));
    if (W >= 32) { double a = v, b = v; swap_rows<16>(a, b); v = fmax(a, b); }
    if (W >= 64) { double a = v, b = v; swap_rows<32>(a, b); v = fmax(a, b); }
    return v;
}

// chi2 sum + max (or two sums) over the workgroup with ONE pair of barriers: the two butterflies interleave
template <int NW, bool SECOND_IS_MAX>
__device__ __forceinline__ void block_reduce2(double& a, double& b, double* s_red) {
    a = group_sum<64>(a);
    b = SECOND_IS_MAX ? group_max<64>(b) : group_sum<64>(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = a; s_red[8 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    double ra = s_red[0], rb = s_red[8];
#pragma unroll
    for (int w = 1; w < NW; w++) { ra += s_red[w]; rb = SECOND_IS_MAX ? fmax(rb, s_red[8 + w]) : rb + s_red[8 + w]; }
    a = ra; b = rb;
}

// Butterfly TRANSPOSE over the lane bits 32 .. STOP (reduce.hpp's WaveTranspose stopped early): lanes that differ only in those bits
// end up owning disjoint index ranges [off, off + real) of the N sums over their group — N-1-ish cross-lane moves instead of 3N.
template <int N, int BIT, int STOP>
struct PartTranspose {
    static constexpr int H = (N + 1) / 2;
    __device__ static __forceinline__ void run(double (&v)[N], int lane, int& off, int& real) {
        const bool up = (lane & BIT) != 0;
        double nv[H];
#pragma unroll
        for (int i = 0; i < H; i++) {
            const double lo = v[i];
            const double hi = (i + H < N) ? v[(i + H < N) ? i + H : 0] : 0.0;
            if constexpr (BIT == 32 || BIT == 16) {   // rows exchanged between the two registers: (own lo + partner's lo) below, (partner's hi + own hi) above
                double a = lo, b = hi;
                swap_rows<BIT>(a, b);
                nv[i] = a + b;
            } else if constexpr (BIT == 8) {
                const double keep = up ? hi : lo, send = up ? lo : hi;
                nv[i] = keep + dpp_f64<kDppRor8>(send);
            } else {
                const double keep = up ? hi : lo, send = up ? lo : hi;
                nv[i] = keep + __shfl_xor(send, BIT);
            }
        }
        if (up) { off += H; real = real - H > 0 ? real - H : 0; }
        else real = real < H ? real : H;
#pragma unroll
        for (int i = 0; i < H; i++) v[i] = nv[i];
        if constexpr (BIT > STOP) {
            double (&w)[H] = reinterpret_cast<double (&)[H]>(v);
            PartTranspose<H, BIT / 2, STOP>::run(w, lane, off, real);
        }
    }
};

// chi2 sum, max and a second sum with the same pair of barriers (s_red holds 16 doubles)
template <int NW>
__device__ __forceinline__ void block_reduce3(double& a, double& b, double& c, double* s_red) {
    a = group_sum<64>(a); b = group_max<64>(b); c = group_sum<64>(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = a; s_red[8 + (threadIdx.x >> 6)] = b; s_red[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    double ra = s_red[0], rb = s_red[8], rc = s_red[4];
#pragma unroll
    for (int w = 1; w < NW; w++) { ra += s_red[w]; rb = fmax(rb, s_red[8 + w]); rc += s_red[4 + w]; }
    a = ra; b = rb; c = rc;
}


template <int NW>
__device__ __forceinline__ double block_max_n(double v, double* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = fmax(r, s_red[w]);
    return r;
}

template <int NF>
__global__ __launch_bounds__(kPThreads) void ba_persist_kernel(BAPtrs p, BADims d, BAPersist q) {
    UH_BA_CLK(0);
    // one wave per SIMD, every instruction on the critical path: when the tracking stream's waves share the SIMD, this wave issues first
    __builtin_amdgcn_s_setprio(3);
    uh_latency_critical();
    static_assert(NF == 8 || NF == 16, "lanes per landmark = padded number of free cameras");
    constexpr int NP = 6 * NF, YS = NP + 2;
    constexpr int LG = NF == 8 ? 3 : 4;
    constexpr int NHP = NF == 8 ? 5 : 9;        // camera-side sums a lane owns after the butterfly transpose over 64 / NF landmarks
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = blockIdx.x;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);   // the wave index as a scalar: wave-uniform branches become s_cbranch
    const int n = d.n, ld = n + 1, nfree = d.nfree, npairs = nfree * (nfree + 1) / 2;
    const int OFF_CAM = q.nblk * 16, OFF_BS = OFF_CAM + NF * 27, OFF_SC = OFF_BS + NP;
    const PersistLds o = persist_lds<NF>(q.krows, n, q.max_fix, q.kfix, OFF_CAM, q.KS, q.SL);
    double* const Yt = lds + o.Yt;
    double* const U = lds + o.U;
    double* const Mm = U;
    double (*const s_w)[121][6] = reinterpret_cast<double (*)[121][6]>(U + (size_t)ld * ld);
    double* const s_out = lds + o.out;
    double* const s_wsum = lds + o.wsum;
    double* const s_x = lds + o.x;
    double* const s_bp = lds + o.bp;
    (void)o.bs;   // (the b_schur receive area of earlier forms: the reduced right-hand side now arrives in row n of Mm)
    double* const s_bsp = lds + o.bsp;
    double* const s_pose = lds + o.pose;
    double* const s_poseR = lds + o.poseR;
    double* const s_red = lds + o.red;
    double* const s_sc = lds + o.sc;
    double* const s_fxchi = lds + o.fxchi;
    double* const s_fxobs = lds + o.fxobs;     // [max_fix][3]: u, v, information scalar
    double* const s_fxcam = lds + o.fxcam;     // [kfix][kFxCam]: R | t (12), fx fy cx cy, row 2 of the input float pose
    unsigned char* const s_fxact = reinterpret_cast<unsigned char*>(lds) + o.fxact_bytes;
    unsigned char* const s_fxk = reinterpret_cast<unsigned char*>(lds) + o.fxk_bytes;
    short (*const s_pair)[2] = reinterpret_cast<short (*)[2]>(reinterpret_cast<unsigned char*>(lds) + o.pair_bytes);
    unsigned char (*const s_blk)[2] = reinterpret_cast<unsigned char (*)[2]>(reinterpret_cast<unsigned char*>(lds) + o.blk_bytes);   // 4x4 block -> (bi, bj)
    int* const s_flag = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + o.flag_bytes);   // [0] solve ok, [1] error

    const int G = q.G, SL = q.SL;
    // (OFF_CAM / OFF_BS / OFF_SC: element offsets of a partial — the product (the nblk upper 4x4 blocks of the vector-FMA form, or six 16x16
    // MFMA tiles), camera sums, b_schur, scalars)
    const int l0 = g * q.Lw;
    const int nl = min(q.Lw, d.P - l0);
    const int ll = tid >> LG, s = tid & (NF - 1);
    const bool live = ll < nl;
    const int l = live ? l0 + ll : l0;
    int* const s_fxid = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + o.fxid_bytes);     // observation index of every staged fixed-camera observation
    int* const s_fxptr = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + o.fxptr_bytes);   // [Lw + 1]: a landmark's range of them
    const int KK = d.K;
    // observation index of (point, frame), or -1: only a cell written by THIS problem's ingest carries its sequence number
    auto cell = [&](int pt, int kf) -> int {
        const unsigned t = q.T[(size_t)pt * KK + kf];
        return (t & 0xFFF00000u) == q.tseq ? (int)(t & 0xFFFFFu) - 1 : -1;
    };

    // ---- lane-resident constants and state
    int eid = -1;
    if (live && s < nfree) eid = cell(l, p.free_kf[s]);
    const bool has = eid >= 0;
    double ou = 0, ov = 0, ow = 0, fxk = 1, fyk = 1, cxk = 0, cyk = 0;
    // one observation's constants from either record format
    auto load_obs = [&](int e, double& u_, double& v_, double& w_) {
        if (q.obs16) {
            const float4 r = reinterpret_cast<const float4*>(q.obs)[e];
            u_ = r.y; v_ = r.z; w_ = r.w;
        } else { const uh_ba_obs ob = q.obs[e]; u_ = ob.u; v_ = ob.v; w_ = ob.inv_sigma; }
    };
    if (has) load_obs(eid, ou, ov, ow);
    if (s < nfree) { const int k = p.free_kf[s]; fxk = p.intr[4 * k]; fyk = p.intr[4 * k + 1]; cxk = p.intr[4 * k + 2]; cyk = p.intr[4 * k + 3]; }
    bool act = has;
    double chi_e = 0;
    double X[3] = {0, 0, 1}, Xt[3] = {0, 0, 1};
    if (live) { X[0] = q.points[3 * (size_t)l]; X[1] = q.points[3 * (size_t)l + 1]; X[2] = q.points[3 * (size_t)l + 2]; }
    double ci00 = 0, ci11 = 0, ci22 = 0, cl10 = 0, cl20 = 0, cl21 = 0, wl0 = 0, wl1 = 0, wl2 = 0, bl0 = 0, bl1 = 0, bl2 = 0;
    bool any_pt = false;

    for (int i = tid; i < q.krows * YS; i += kPThreads) Yt[i] = 0.0;
    // LDS arrives with whatever the previous workgroup on this CU left in it: every word that is read before this kernel writes it must
    // be cleared here.  With fewer than NF free cameras the lanes of the unused slots read s_x[6 * nfree ..) in the back-substitution
    // (multiplied by panel zeros, but 0 * NaN = NaN) — found by scripts/ba_stress.py: wrong results only with other kernels' garbage in LDS.
    for (int i = tid; i < 3 * NP; i += kPThreads) s_x[i] = 0.0;          // s_x, s_bp, s_bs are contiguous
    for (int i = tid; i < kPWaves * NP; i += kPThreads) s_bsp[i] = 0.0;
    for (int i = tid; i < NF * 27 + NP + 4; i += kPThreads) s_out[i] = 0.0;
    for (int i = tid; i < 2 * NF * 7; i += kPThreads) s_pose[i] = 0.0;
    for (int i = tid; i < 2 * NF * 12; i += kPThreads) s_poseR[i] = 0.0;
    if (tid < 8) s_sc[tid] = 0.0;
    __syncthreads();
    // observations by fixed cameras: staged in LDS per landmark, in ascending frame order (a fixed summation order); one thread per
    // landmark walks the fixed frames' cells twice (count, then place) — kfix is 1-2 for a local BA window
    if (tid < nl) {
        int c = 0;
        for (int j = 0; j < q.kfix; j++) c += cell(l0 + tid, q.fix_kf[j]) >= 0;
        s_fxptr[tid + 1] = c;
    }
    if (tid == 0) s_fxptr[0] = 0;
    __syncthreads();
    if (tid == 0) for (int i = 0; i < nl; i++) s_fxptr[i + 1] += s_fxptr[i];
    __syncthreads();
    if (tid < nl) {
        int at = s_fxptr[tid];
        for (int j = 0; j < q.kfix; j++) {
            const int e = cell(l0 + tid, q.fix_kf[j]);
            if (e < 0) continue;
            double fu, fv, fw;
            load_obs(e, fu, fv, fw);
            s_fxact[at] = 1; s_fxchi[at] = 0.0; s_fxk[at] = (unsigned char)j; s_fxid[at] = e;
            s_fxobs[3 * at] = fu; s_fxobs[3 * at + 1] = fv; s_fxobs[3 * at + 2] = fw;
            ++at;
        }
    }
    for (int i = tid; i < q.kfix * kFxCam; i += kPThreads) {
        const int f = i / kFxCam, j = i - f * kFxCam, k = q.fix_kf[f];
        s_fxcam[i] = j < 12 ? q.poseR0[12 * k + j] : (j < 16 ? p.intr[4 * k + (j - 12)] : (double)q.poses_in[16 * (size_t)k + 8 + (j - 16)]);
    }
    const int fxb = live ? s_fxptr[ll] : 0, fxe = live ? s_fxptr[ll + 1] : 0;   // this landmark's fixed-camera observations
    if (tid < nfree) {
        const int k = p.free_kf[tid];
        for (int b = 0; b < 2; b++) {
            for (int j = 0; j < 7; j++) s_pose[(b * NF + tid) * 7 + j] = q.pose0[7 * k + j];
            for (int j = 0; j < 12; j++) s_poseR[(b * NF + tid) * 12 + j] = q.poseR0[12 * k + j];
        }
    }
    for (int t = tid; t < npairs; t += kPThreads) {
        int s1 = 0, rem = t;
        while (rem >= nfree - s1) { rem -= nfree - s1; ++s1; }
        s_pair[t][0] = (short)s1; s_pair[t][1] = (short)(s1 + rem);
    }
    for (int t = tid; t < q.nblk; t += kPThreads) {   // upper blocks of the nb4 x nb4 grid, row-major
        int bi = 0, rem = t;
        while (rem >= q.nb4 - bi) { rem -= q.nb4 - bi; ++bi; }
        s_blk[t][0] = (unsigned char)bi; s_blk[t][1] = (unsigned char)(bi + rem);
    }
    if (tid == 0) { s_flag[0] = 1; s_flag[1] = 0; s_flag[2] = 0; }
    __syncthreads();

    // ---- exchange rounds: every workgroup runs the same sequence of rounds, so the round counter doubles as the tag
    unsigned ep = 0, tagA = 0;
    auto next_tag = [&]() -> unsigned { ++ep; return q.tag_base | (ep & 0xFFFu); };
    const pword err_word = ((pword)q.launch_id << 32) | 1ull;
    // a reader that has waited 3 s (a workgroup never became resident), or sees that somebody else gave up: flag it, everybody leaves
    // at the next uniform check of s_flag[1]
    // t0 < 63: a count of misses — the first 63 polls repeat at once: a poll is one memory round trip and the error word costs a second,
    // SERIAL one, which used to double the polling period of every hand-off —; from then on the wall-clock time the slow path began
    // (wall-clock values are far above 64).
    auto give_up = [&](long long& t0) -> bool {
        if (t0 < 63) { ++t0; return false; }
        if (t0 == 63) t0 = wall_clock64();
        if (__hip_atomic_load(q.errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == err_word || wall_clock64() - t0 > q.timeout_ticks) {
            s_flag[1] = 1;
            __hip_atomic_store(q.errw, err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // whichever workgroup gives up tells the host (the same word from all of them): the one that never became resident may be workgroup 0
            if (q.host_done) __hip_atomic_store(q.host_done, ((unsigned long long)q.launch_id << 32) | 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        return false;
    };
    // up to eight tagged elements (first, first + stride, ...), all loads in flight together; spins until every tag matches
    // (need: bit u clear = element u is not wanted — nobody sent it this round — and is neither fetched nor waited for)
    auto tload8 = [&](const pword* base, size_t first, size_t stride, int cnt, unsigned tag, double (&v)[8], unsigned need = 0xFFu) {
        long long t0 = 0;
        for (;;) {
            TWord w[8];
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 8; u++) if (u < cnt && ((need >> u) & 1u)) w[u] = tld_raw(base, first + u * stride);
#pragma unroll
            for (int u = 0; u < 8; u++) { if (u < cnt && ((need >> u) & 1u)) { ok = ok && tok(w[u], tag); v[u] = tval(w[u]); } else v[u] = 0.0; }
            if (ok || give_up(t0)) return;
        }
    };
    // (i / SL and h-range / HG by multiplication: a 32-bit division is ~25 instructions, and the hand-offs did nine of them per thread and trial;
    // m = floor(2^32 / d) + 1 gives floor(i / d) = mulhi(i, m) exactly for i * d < 2^32 — here i < 2^16)
    const unsigned sl_inv = (unsigned)(0x100000000ull / (unsigned)SL) + 1u;
    auto div_sl = [&](int i) -> int { return (int)__umulhi((unsigned)i, sl_inv); };
    auto part_at = [&](int i) -> size_t { const int sl = div_sl(i); return ((size_t)sl * G + g) * SL + (i - sl * SL); };   // element i of this workgroup's partial

    // element of the product partial that holds (row, col), row <= col (the inverse of dst_of's layouts)
    auto prod_index = [&](int row, int col) -> int {
        const int bi = row >> 2, bj = col >> 2;
        return (bi * q.nb4 - bi * (bi - 1) / 2 + (bj - bi)) * 16 + (row & 3) * 4 + (col & 3);
    };
    // is element idx of a partial a diagonal entry of S?  (the slice reduction puts lambda there)
    auto is_diag_elem = [&](int idx) -> bool {
        if (idx >= OFF_CAM) return false;
        int row, col;
        const int bq = idx >> 4;
        row = 4 * s_blk[bq][0] + ((idx >> 2) & 3); col = 4 * s_blk[bq][1] + (idx & 3);
        return row == col && col < n;
    };
    // this thread's share of folding Hpp (21 upper entries per free camera) into the product: product element and camera-sum index
    int fold_i[2], fold_s[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int t = tid + u * kPThreads;
        fold_i[u] = -1; fold_s[u] = 0;
        if (t < nfree * 21) {
            const int sc = t / 21, qq = t - 21 * sc;
            int a = 0, rem = qq;
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }
            fold_i[u] = prod_index(6 * sc + a, 6 * sc + a + rem);
            fold_s[u] = sc * 27 + qq;
        }
    }
    const bool red_diag0 = tid < SL && is_diag_elem(g * SL + tid);
    const int open_beg = div_sl(OFF_CAM) * SL;   // first element of the slice that straddles the end of the product part
    const int red_HG = max(1, min(G, kPThreads / SL));   // source groups per element in the slice reduction
    const unsigned hg_inv = (unsigned)(0x100000000ull / (unsigned)red_HG) + 1u;
    bool clk_on = false;
    int n_kept = 0, n_dropped = 0;   // speculative trials kept / dropped (debug clocks 58 / 59, written with the results)
#define UH_BA_CLKT(i) do { if (clk_on) UH_BA_CLK(i); } while (0)
    BAState st;
    memset(&st, 0, sizeof(st));
    st.phase = 2; st.lambda = -1; st.ni = 2;
    bool robust = true;
    double chi_lin_pass = 0;

    // ================================================================================ linearisation at a given estimate
    // The lambda-independent part of a linearisation: this lane's observation (+ its share of the landmark's fixed-camera observations)
    // evaluated at pose buffer `buf` and point Xp, into acc / hp / H.  Called at the top of a trial — or, speculatively, at the TRIAL
    // estimate while the chi2 hand-off (C) is in flight: if the trial is accepted (the normal case) the next trial starts with its
    // linearisation already in registers, the 1.6 us of edge evaluation having run inside a wait that was idle before.
    double acc[10];
    double hp[33];   // camera-side sums of this observation: Hpp upper triangle (21), bp (6), b_schur = Y_e (L^-1 b_l) (6)
    double H[18];
    bool any = false;
    auto lin_eval = [&](int buf, const double (&Xp)[3], bool first) {
#pragma unroll
        for (int i = 0; i < 10; i++) acc[i] = 0;
#pragma unroll
        for (int i = 0; i < 33; i++) hp[i] = 0;
#pragma unroll
        for (int i = 0; i < 18; i++) H[i] = 0;
        any = false;
        const bool on = has && act;
        if (on) {
            EdgeLin L;
            edge_eval_p<2>(ou, ov, ow, fxk, fyk, cxk, cyk, d.delta, d.dsqr, s_poseR + (buf * NF + s) * 12, Xp, robust, L);
            any = true;
            if (first) chi_e = L.chi2;
            acc[9] = L.robchi;
            acc[0] = L.ww * (L.A[0] * L.A[0] + L.A[3] * L.A[3]); acc[1] = L.ww * (L.A[0] * L.A[1] + L.A[3] * L.A[4]);
            acc[2] = L.ww * (L.A[0] * L.A[2] + L.A[3] * L.A[5]); acc[3] = L.ww * (L.A[1] * L.A[1] + L.A[4] * L.A[4]);
            acc[4] = L.ww * (L.A[1] * L.A[2] + L.A[4] * L.A[5]); acc[5] = L.ww * (L.A[2] * L.A[2] + L.A[5] * L.A[5]);
            acc[6] = L.A[0] * L.r0 + L.A[3] * L.r1; acc[7] = L.A[1] * L.r0 + L.A[4] * L.r1; acc[8] = L.A[2] * L.r0 + L.A[5] * L.r1;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) H[a * 3 + c] = L.ww * (L.B[a] * L.A[c] + L.B[6 + a] * L.A[3 + c]);
            int qq = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = a; c < 6; c++) hp[qq++] = L.ww * (L.B[a] * L.B[c] + L.B[6 + a] * L.B[6 + c]);
#pragma unroll
            for (int a = 0; a < 6; a++) hp[21 + a] = L.B[a] * L.r0 + L.B[6 + a] * L.r1;
        }
        {   // observations by fixed cameras (constants staged in LDS): Hll, bl and chi2 only
            for (int i = fxb + s; i < fxe; i += NF) {
                if (!s_fxact[i]) continue;
                any = true;
                const double* C = s_fxcam + kFxCam * s_fxk[i];
                EdgeLin L;
                edge_eval_p<1>(s_fxobs[3 * i], s_fxobs[3 * i + 1], s_fxobs[3 * i + 2], C[12], C[13], C[14], C[15], d.delta, d.dsqr, C, Xp, robust, L);
                if (first) s_fxchi[i] = L.chi2;
                acc[9] += L.robchi;
                acc[0] += L.ww * (L.A[0] * L.A[0] + L.A[3] * L.A[3]); acc[1] += L.ww * (L.A[0] * L.A[1] + L.A[3] * L.A[4]);
                acc[2] += L.ww * (L.A[0] * L.A[2] + L.A[3] * L.A[5]); acc[3] += L.ww * (L.A[1] * L.A[1] + L.A[4] * L.A[4]);
                acc[4] += L.ww * (L.A[1] * L.A[2] + L.A[4] * L.A[5]); acc[5] += L.ww * (L.A[2] * L.A[2] + L.A[5] * L.A[5]);
                acc[6] += L.A[0] * L.r0 + L.A[3] * L.r1; acc[7] += L.A[1] * L.r0 + L.A[4] * L.r1; acc[8] += L.A[2] * L.r0 + L.A[5] * L.r1;
            }
        }
    };
    // ================================================================================ phase 1: linearise + Schur partial
    // first == true: the pass's opening evaluation at the current estimate (computeActiveErrors + computeLambdaInit): chi2 of
    // every observation is recorded, the reduced-system product is skipped.  acc / hp / H hold the linearisation (lin_eval, called just before).
    auto phase1 = [&](double lambda, bool first, double scale_lane, double stop_val) {
        tagA = next_tag();
        if (!first) UH_BA_CLKT(52);
#pragma unroll
        for (int i = 0; i < 10; i++) acc[i] = group_sum<NF>(acc[i]);
        const unsigned long long anym = __ballot(any);
        any_pt = ((anym >> (lane & ~(NF - 1))) & ((1ull << NF) - 1)) != 0;
        bl0 = acc[6]; bl1 = acc[7]; bl2 = acc[8];
        if (first) {
            // (opening evaluation: chi2, max |H_jj| and the camera sums are all that is needed — no factor, no panel)
        } else if (any_pt) {   // D = Hll + lambda I = L L^T
            const double d00 = acc[0] + lambda, d11 = acc[3] + lambda, d22 = acc[5] + lambda;
            ci00 = rsqrt_nr(d00);
            cl10 = acc[1] * ci00; cl20 = acc[2] * ci00;
            ci11 = rsqrt_nr(d11 - cl10 * cl10);
            cl21 = (acc[4] - cl20 * cl10) * ci11;
            ci22 = rsqrt_nr(d22 - cl20 * cl20 - cl21 * cl21);
            wl0 = bl0 * ci00;
            wl1 = (bl1 - cl10 * wl0) * ci11;
            wl2 = (bl2 - cl20 * wl0 - cl21 * wl1) * ci22;
        } else {
            ci00 = ci11 = ci22 = cl10 = cl20 = cl21 = wl0 = wl1 = wl2 = 0;
        }
        if (!first && 3 * ll + 2 < q.krows) {   // whitened blocks Y_e = Hpl_e L^-T, transposed into the panel (zeros where there is no observation)
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double y0 = H[a * 3] * ci00;
                const double y1 = (H[a * 3 + 1] - cl10 * y0) * ci11;
                const double y2 = (H[a * 3 + 2] - cl20 * y0 - cl21 * y1) * ci22;
                Yt[(3 * ll) * YS + 6 * s + a] = y0; Yt[(3 * ll + 1) * YS + 6 * s + a] = y1; Yt[(3 * ll + 2) * YS + 6 * s + a] = y2;
                hp[27 + a] = y0 * wl0 + y1 * wl1 + y2 * wl2;
            }
        }
        // camera-side sums over this wave's landmarks: butterfly transpose over the landmark bits of the lane index — afterwards the
        // 64 / NF lanes of a camera slot each own a few of its 33 sums
        {
            int off = 0, real = 33;
            PartTranspose<33, 32, NF>::run(hp, lane, off, real);
#pragma unroll
            for (int i = 0; i < NHP; i++) if (i < real) s_wsum[(wv * NF + s) * 33 + off + i] = hp[i];
        }
        if (!first) UH_BA_CLKT(53);
        const double chi_part = (live && s == 0) ? acc[9] : 0.0;
        const double maxd = (live && s == 0 && any_pt) ? fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5]))) : 0.0;
        double cs = chi_part, mx = maxd, ss = scale_lane;
        block_reduce3<kPWaves>(cs, mx, ss, s_red);   // (its barriers also publish Yt / s_wv / the camera sums)
        for (int t = tid; t < NF * 33; t += kPThreads) {   // 264 sums, wave order
            double r = s_wsum[t];
#pragma unroll
            for (int w = 1; w < kPWaves; w++) r += s_wsum[w * NF * 33 + t];
            const int sc = t / 33, i = t - 33 * sc;
            s_out[i < 27 ? sc * 27 + i : NF * 27 + 6 * sc + (i - 27)] = r;
        }
        // scalars of a partial: chi2 at the linearisation point, (speculative trial:) the previous trial's scale sum, max |Hll_jj|, stop flag
        if (tid == 0) { s_out[NF * 27 + NP] = cs; s_out[NF * 27 + NP + 1] = ss; s_out[NF * 27 + NP + 2] = mx; s_out[NF * 27 + NP + 3] = g == 0 ? stop_val : 0.0; }
        if (!first) UH_BA_CLKT(54);
        // S = Yt^T Yt: vector FMA, 4x4 register blocks — the nblk upper blocks of the nb4 x nb4 block grid x KS splits of the K
        // range (78 x 3 = 234 items for eight free cameras: one per lane; more cameras: several per lane), 16 accumulators each, four
        // ds_read_b128 per 16 FMAs; the splits are added in order through LDS.  Measured on MI355X (scripts/micro/mfma_f64_rate.hip):
        // v_fma_f64 sustains 9.1-11.4 FMA/clk/SIMD, v_mfma_f64_16x16x4_f64 issues every ~160 cycles = 6.4: the MFMA form of this product
        // (six upper 16x16 tiles split over the waves, accumulators resident in AGPRs; rounds 2-4 behind UH_BA_SCHUR=mfma) lost on
        // gfx950 — kernel 0.454 against 0.430 ms — and was removed in round 5.
        if (!first) {   // (U is free: its last users — the reduced system, the slice reduction's scratch — are behind barriers)
            const int nitems = q.nblk * q.KS;
            const int kc = (q.krows + q.KS - 1) / q.KS;
            for (int item = tid; item < nitems; item += kPThreads) {
                const int kt = item / q.nblk, bq = item - kt * q.nblk;
                const int bi = s_blk[bq][0], bj = s_blk[bq][1];
                const int kbeg = kt * kc, kend = min(q.krows, kbeg + kc);
                double acc4[16];
#pragma unroll
                for (int i = 0; i < 16; i++) acc4[i] = 0;
                const double* ra = Yt + 4 * bi, * rb = Yt + 4 * bj;
                // four rows per iteration: their sixteen ds_read_b128 are issued together, so one LDS latency is paid per 64 FMAs
                // (the compiler serialises load -> wait -> FMA inside an iteration and undoes hand-rotated prefetching)
                for (int k = kbeg; k < kend; k += 4) {
                    double2 a01[4], a23[4], b01[4], b23[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kk = k + u < kend ? k + u : kbeg;
                        a01[u] = *reinterpret_cast<const double2*>(ra + kk * YS); a23[u] = *reinterpret_cast<const double2*>(ra + kk * YS + 2);
                        b01[u] = *reinterpret_cast<const double2*>(rb + kk * YS); b23[u] = *reinterpret_cast<const double2*>(rb + kk * YS + 2);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double z = k + u < kend ? 1.0 : 0.0;   // rows past the end contribute nothing
                        const double av[4] = {a01[u].x * z, a01[u].y * z, a23[u].x * z, a23[u].y * z}, bv[4] = {b01[u].x, b01[u].y, b23[u].x, b23[u].y};
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int c = 0; c < 4; c++) acc4[r * 4 + c] = fma(av[r], bv[c], acc4[r * 4 + c]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; i++) U[kt * OFF_CAM + bq * 16 + i] = acc4[i];
            }
            // (the K-splits' partials stay apart: the store loop at the end adds them, in order, on the way out)
        }
        if (!first) UH_BA_CLKT(55);
        __syncthreads();   // the product (U) and the camera sums (s_out) are complete
        if (!first) UH_BA_CLKT(56);
        if (!first) {
            // S = sum over workgroups of (Hpp_g - Y_g^T Y_g) + lambda I and b = sum of (bp_g - b_schur_g): each workgroup folds its camera
            // sums into its product partial and its b_schur partial HERE, so that what arrives after hand-off B is the reduced system
            // itself (the receiver used to add Hpp and form bp - b_schur in a second pass behind another barrier, on the critical path)
#pragma unroll
            for (int u = 0; u < 2; u++) if (fold_i[u] >= 0) U[fold_i[u]] -= s_out[fold_s[u]];
            for (int j = tid; j < n; j += kPThreads) { const int sc = j / 6; s_out[NF * 27 + j] -= s_out[sc * 27 + 21 + (j - 6 * sc)]; }
        }
        __syncthreads();
        if (!first) UH_BA_CLKT(60);
        // the partial goes out with consecutive lanes on consecutive words (one element per lane and instruction was 64 cache lines per store)
        // (a trial does not send the 21 Hpp entries of a camera's 27 sums: they have been folded into the product above, nobody reads them —
        // only the opening evaluation needs them, for max |H_jj|; 168 of 1516 elements less on the wire)
        for (int i = tid; i < NF * 27 + NP + 4; i += kPThreads) {
            if (!first && i < NF * 27 && i - 27 * (i / 27) < 21) continue;
            tst(q.part, part_at(OFF_CAM + i), s_out[i], tagA);
        }
        if (!first) UH_BA_CLKT(61);
        const int KSs = q.KS;
        // (opening evaluation: there is no product; only the slice that straddles the end of the product part is sent, as zeros, so that
        // its reducer finds the round's tag on every element — the slices below it are not reduced at all in that round)
        // (Round 5 stamps: camera-sum stores 0.36 us, these five stores per thread 1.04 us — the write-through stores themselves, ~250 ns per
        // wave instruction; fetching all K-split partials of four elements before the first addition made it 1.36: not the LDS reads.)
        for (int i = (first ? open_beg : 0) + tid; i < OFF_CAM; i += kPThreads) {
            double r = 0.0;
            if (!first) { r = U[i]; for (int kt = 1; kt < KSs; kt++) r += U[kt * OFF_CAM + i]; }
            tst(q.part, part_at(i), r, tagA);
        }
    };

    // ================================================================================ exchange A -> B: slice-wise reduction
    unsigned tagB = 0;
    auto reduce_slices = [&](double lam, bool first) -> bool {   // lam: added to the diagonal of S here; first: the opening evaluation (camera sums and scalars only)
        tagB = next_tag();
        // HG groups of sources per element, chosen so that one pass of the workgroup covers the slice (SL * HG <= 256 threads) and a
        // thread's sources (~G / HG <= 8) go out as ONE batch of loads: every extra pass or batch is a memory round trip (~1.5 us)
        const int HG = red_HG;
        double* const R = U + q.KS * OFF_CAM;   // behind the product's partials, which other waves may still be sending
        const size_t src = (size_t)g * G * SL;   // slice g of every workgroup's partial
        for (int idx = tid; idx < SL * HG; idx += kPThreads) {
            const int hg = div_sl(idx), e = idx - hg * SL;
            const bool is_max = g * SL + e == OFF_SC + 2;
            const int ge = g * SL + e;
            const bool hpp = ge >= OFF_CAM && ge < OFF_BS && (ge - OFF_CAM) - 27 * ((ge - OFF_CAM) / 27) < 21;   // (an Hpp entry: sent in the opening evaluation only)
            const bool used = ge < q.nelem && !(first && (g + 1) * SL <= OFF_CAM) && !(hpp && !first);   // (the last slice is padded: nobody writes or needs those elements; opening: nor the product's slices)
            // every source of this element in ONE batch of loads where possible (G <= 128: at most eight per thread): a separate load
            // for the first source put a second memory round trip (~1.5 us) in front of every slice reduction
            double r = 0.0;
            bool have = false;
            for (int h = hg; used && h < G; h += 8 * HG) {   // eight elements in flight, added in ascending order
                double v[8];
                const int cnt = min(8, HG == 1 ? G - h : (int)__umulhi((unsigned)(G - h + HG - 1), hg_inv));   // (the multiplier of a division by one does not fit 32 bits)
                tload8(q.part, src + (size_t)h * SL + e, (size_t)HG * SL, cnt, tagA, v);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (u >= cnt) continue;
                    r = !have ? v[u] : (is_max ? fmax(r, v[u]) : r + v[u]);
                    have = true;
                }
            }
            R[idx] = r;
        }
        UH_BA_CLKT(50);
        __syncthreads();
        if (s_flag[1]) return false;
        for (int e = tid; e < SL; e += kPThreads) {
            const bool is_max = g * SL + e == OFF_SC + 2;
            double r = R[e];
            for (int hg = 1; hg < HG; hg++) { const double v = R[hg * SL + e]; r = is_max ? fmax(r, v) : r + v; }
            if (e == tid ? red_diag0 : is_diag_elem(g * SL + e)) r -= lam;   // (the product arrives negated: - (Y^T Y - Hpp - lambda))
            {
                const int ge = g * SL + e;
                if (!first && ge >= OFF_CAM && ge < OFF_BS && (ge - OFF_CAM) - 27 * ((ge - OFF_CAM) / 27) < 21) continue;   // (nobody fetches an Hpp entry in a trial)
            }
            tst(q.red, (size_t)g * SL + e, r, tagB);
        }
        UH_BA_CLKT(51);
        __syncthreads();   // U is free again
        return true;
    };

    // Where the elements this thread fetches from the reduced vector go (the same every trial): an offset into `lds` (doubles), with bit 24
    // set for the entries that are copied (bp -> s_bp, the scalars -> s_sc) and clear for the entries that arrive negated: the product
    // (Y^T Y - Hpp - lambda on the diagonal) into the lower triangle of S, b_schur - bp into row n; -1: nothing to store.
    auto dst_of = [&](int idx) -> int {
        int t = -1;
        if (idx < q.nelem) {
            if (idx < OFF_CAM) {
                int row, col;
                // (4x4 block of the upper block triangle, r, c)
                const int bq = idx >> 4, rr = (idx >> 2) & 3, cq = idx & 3;
                row = 4 * s_blk[bq][0] + rr; col = 4 * s_blk[bq][1] + cq;
                if (row <= col && col < n) t = o.U + col * ld + row;
            } else if (idx < OFF_BS) {   // camera sums: bp is kept (computeScale needs it); the Hpp entries have been folded into the product by the senders
                const int i = idx - OFF_CAM, sc = i / 27, k = i - 27 * sc;
                if (k >= 21) t = (o.bp + 6 * sc + (k - 21)) | (1 << 24);
            } else if (idx < OFF_SC) t = o.U + n * ld + (idx - OFF_BS);   // b_schur - bp, negated on arrival: the bordered system's row n
            else t = (o.sc + 4 + (idx - OFF_SC)) | (1 << 24);   // the four scalars -> s_sc[4 .. 7] (the speculative trial's decision reads them)
        }
        return t;
    };
    int asm_dst[8];   // the first batch of eight (all of it for up to eight free cameras) is worked out once; later batches on the fly
#pragma unroll
    for (int u = 0; u < 8; u++) asm_dst[u] = dst_of(tid + u * kPThreads);
    UH_BA_CLK(1);
    for (int pass = 0; pass < 2; pass++) {
        UH_BA_CLK(2 + 3 * pass);
        // ---- begin_pass (legacy ba_begin_pass_kernel / ba_gate_kernel / ba_relabel_kernel)
        if (pass == 1) {
            if (st.stopped) break;
            st.gate = 1;
            st.iters_pass1 = st.iters_done;
            // globaloptimizer_g2o.cpp:434-449: chi2 > 5.99 or non-positive depth -> level 1; every robust kernel dropped
            if (has) {
                const double* Rt = s_poseR + (st.cur * NF + s) * 12;
                const double z = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
                if (chi_e > d.chi2_th || !(z > 0.0)) act = false;
            }
            for (int i = fxb + s; i < fxe; i += NF) {
                const double* Rt = s_fxcam + kFxCam * s_fxk[i];
                const double z = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
                if (s_fxchi[i] > d.chi2_th || !(z > 0.0)) s_fxact[i] = 0;
            }
            robust = false;
            __syncthreads();
        }
        const int max_iters = pass == 0 ? q.n1 : q.n2;
        st.pending = 0; st.stop_seen = 0;
        st.phase = max_iters > 0 ? 0 : 2;
        st.iteration = 0; st.max_iters = max_iters; st.qmax = 0; st.solve_ok = 1; st.first_trial = 1; st.iters_done = 0;
        st.prevChi2 = FLT_MAX; st.curChi2 = FLT_MAX; st.minChi2 = q.minChi2;
        if (pass == 0 && q.stop_at_begin) { st.phase = 2; st.stopped = 1; }
        if (st.phase == 2) continue;

        // ONE ITERATION of the loop below = [linearise] -> phase 1 -> hand-offs A, B -> [open decision] -> solve -> update -> (speculate | errors, C,
        // decision).  The linearisation and phase 1 have exactly one call site (they are ~600 instructions and ~150 live registers:
        // three inlined copies spilled 60 registers per lane and cost 2 us per trial); the pass's opening evaluation (chi2 at the current
        // estimate, lambda = tau * max |H_jj|: computeLambdaInit) is the loop's first iteration, which leaves after hand-off A.
        //
        // SPECULATIVE TRIALS.  g2o accepts a step with gain ratio rho >= 0.94 by lambda <- lambda / 3 — every trial of a converging local BA.
        // So a trial that cannot be the pass's last does not hand its chi2 around and wait for the verdict (hand-off C): it linearises at
        // the TRIAL estimate at once (that evaluation IS the trial's error evaluation), runs phase 1 with lambda / 3 and sends the next
        // trial's partial with the chi2 / scale sums of this one riding in its scalar slots.  Every workgroup takes the decision when the
        // reduced vector arrives (B); accepted with exactly lambda / 3: the assembled system is the next trial's, two hand-offs per
        // trial instead of three.  Anything else (rejected, another lambda, stop): the partial is dropped and phase 1 runs again from
        // the decided state, identically in every workgroup.
        bool opening = true;
        bool pend = false;        // a speculative partial is out; the decision on the trial it follows is still open
        double lambda_spec = 0, scale_lane = 0, stop_val = 0;   // what the speculative phase 1 runs with / carries
        int loop_no = 0;   // (the trial-phase clocks 40 .. 57 are those of the second pass's fourth loop iteration: a speculative trial in steady state)
        while (opening || st.phase != 2) {
            ++loop_no;
            clk_on = pass == 1 && loop_no == 4;
            double lambda = pend ? lambda_spec : st.lambda;
            int cur = st.cur, trial = cur ^ 1;
            UH_BA_CLKT(40);
            // The force-stop byte lives in pinned HOST memory: a PCIe round trip (1.5 - 2 us).  It is requested HERE, at the top of the
            // trial, as a load whose destination is LDS (global_load_lds_dword: no register, no wait in the issuing wave) and is read
            // behind the back substitution; the first barrier on the way (phase 1's block reduction, ~3 us later) is the first
            // instruction that waits for it.  Rounds 2-4 fetched it through a register in wave 1 DURING the back substitution: the
            // barrier behind the substitution then waited for the PCIe round trip — 2.3 us of every trial of workgroup 0, hence of
            // everybody, for 0.9 us of substitution.
            if (g == 0 && tid == 64 && p.stop)
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)const_cast<const unsigned char*>(p.stop),
                                                 (__attribute__((address_space(3))) void*)(s_flag + 2), 4, 0, 0);
            {
                const double Xe[3] = {pend ? Xt[0] : X[0], pend ? Xt[1] : X[1], pend ? Xt[2] : X[2]};
                lin_eval(pend ? trial : cur, Xe, opening || pend);   // (chi2 recorded: computeActiveErrors at the pass's start / at the trial estimate)
                phase1(opening ? 1.0 : lambda, opening, scale_lane, stop_val);
            }
            UH_BA_CLKT(41);
            if (!reduce_slices(opening ? 0.0 : lambda, opening)) return;
            UH_BA_CLKT(42);
            if (opening) {
                opening = false;
                // every wave for itself: the diagonal entries of Hpp over the lanes (n <= 128: at most two per lane), max butterfly
                // (all of a lane's words go out as one batch, not as dependent round trips)
                double m = 0.0;
                {
                    const int d0 = lane, d1 = lane + 64;
                    const int sc0 = d0 / 6, a0 = d0 - 6 * sc0, sc1 = d1 / 6, a1 = d1 - 6 * sc1;
                    const size_t i_d0 = d0 < n ? (size_t)(OFF_CAM + sc0 * 27 + (a0 * 6 - a0 * (a0 - 1) / 2)) : (size_t)OFF_SC;   // diagonal of the 21-entry upper triangle
                    const size_t i_d1 = d1 < n ? (size_t)(OFF_CAM + sc1 * 27 + (a1 * 6 - a1 * (a1 - 1) / 2)) : (size_t)OFF_SC;
                    long long t0 = 0;
                    for (;;) {
                        const TWord wm = tld_raw(q.red, OFF_SC + 2), wd0 = tld_raw(q.red, i_d0), wd1 = tld_raw(q.red, i_d1), wc = tld_raw(q.red, OFF_SC);
                        m = lane == 0 ? tval(wm) : 0.0;
                        if (d0 < n) m = fmax(m, fabs(tval(wd0)));
                        if (d1 < n) m = fmax(m, fabs(tval(wd1)));
                        chi_lin_pass = tval(wc);
                        if ((tok(wm, tagB) && tok(wd0, tagB) && tok(wd1, tagB) && tok(wc, tagB)) || give_up(t0)) break;
                    }
                }
#pragma unroll
                for (int oo = 32; oo > 0; oo >>= 1) m = fmax(m, __shfl_xor(m, oo));
                st.lambda = 1e-5 * m; st.ni = 2;
                __syncthreads();
                if (s_flag[1]) return;
                UH_BA_CLK(3 + 3 * pass);
                continue;
            }
            // ---- assemble S = Hpp + lambda I - Yt^T Yt (lower triangle, bordered with b = bp - b_schur), factorise, substitute
            for (int b0 = 0; b0 < q.nelem; b0 += 8 * kPThreads) {   // batches of eight elements per thread (one batch for up to eight free cameras)
                double rv[8];
                const int left = q.nelem - b0 - tid;
                const int cnt = left <= 0 ? 0 : min(8, (left + kPThreads - 1) / kPThreads);
                unsigned need = 0;
#pragma unroll
                for (int u = 0; u < 8; u++) need |= ((b0 == 0 ? asm_dst[u] : dst_of(b0 + tid + u * kPThreads)) >= 0 ? 1u : 0u) << u;
                tload8(q.red, (size_t)(b0 + tid), kPThreads, cnt, tagB, rv, need);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int t = b0 == 0 ? asm_dst[u] : dst_of(b0 + tid + u * kPThreads);
                    if (t >= 0) lds[t & 0xFFFFFF] = (t >> 24) ? rv[u] : -rv[u];   // bit 24: camera sums / b_schur keep their sign, product entries enter S negated
                }
            }
            UH_BA_CLKT(57);   // (round 6: the reduced vector has arrived and is scattered into the factorisation's layout)
            if (tid == 0) s_flag[0] = 1;
            __syncthreads();   // the reduced system [S b] (lower triangle + row n of Mm) and bp are complete
            UH_BA_CLKT(62);
            if (s_flag[1]) return;
            if (pend) {   // the open decision (every thread, same inputs, same code)
                pend = false;
                scale_lane = 0; stop_val = 0;
                DecideSums sm;
                sm.lin = chi_lin_pass; sm.chi = s_sc[4]; sm.scale = s_sc[5]; sm.xs = s_sc[0];
                const bool stopv = s_sc[7] != 0.0;
                st.solve_ok = 1;
                st.pending = 1;
                st = apply_decision(st, sm, stopv);
                const bool accepted = st.cur != cur;
                if (accepted) { X[0] = Xt[0]; X[1] = Xt[1]; X[2] = Xt[2]; }
                const bool kept = accepted && st.phase == 0 && st.lambda == lambda_spec;
                n_kept += kept ? 1 : 0; n_dropped += kept ? 0 : 1;   // (registers: a read-modify-write of the clock block here was a memory round trip on workgroup 0's path, every trial)
                if (!kept) {   // not what was speculated on: again from the decided state
                    __syncthreads();   // (the assembled entries in U are dropped; phase 1 writes there)
                    continue;
                }
                cur = st.cur; trial = cur ^ 1;
            }
            UH_BA_CLKT(43);
            // (the eight-lane instantiation never has more than 48 rows and does not carry the wider forms' code and registers)
            // Factorisation and back substitution.  Up to 10 free keyframes (n + 1 <= 64 rows): row per lane, wave 0 owns the pivots and
            // substitutes straight behind its last block column (its own data: no barrier, no flag round trip in between); ONE barrier
            // then publishes x and the verdict.  11-16 free keyframes: two rows per lane (ldlt_bordered_lds, backsolve2_lds).
            int ok;
            if (NF == 8 || n + 1 <= 64) {
                const bool failed = ldlt_rowlane_v2<false>(Mm, n, ld, nfree, npairs, s_pair, &s_w[0][0][0]);
                UH_BA_CLKT(44);
                if (wvu == 0) {
                    if (!failed) backsolve_v2(Mm, n, ld, s_x, s_sc + 3);   // (s_sc[3]: cleared at the kernel's start and never written again)
                    else if (lane < n) s_x[lane] = 0.0;
                    if (lane == 0) s_flag[0] = failed ? 0 : 1;
                }
                __syncthreads();
                ok = s_flag[0];
            } else {
                const bool failed = ldlt_bordered_lds(Mm, n, ld, nfree, npairs, s_pair, s_w);
                if (failed && lane == 0) s_flag[0] = 0;   // (the wave that saw the pivots)
                __syncthreads();
                UH_BA_CLKT(44);
                ok = s_flag[0];
                if (ok) backsolve2_lds(Mm, n, ld, s_x);
                else for (int i = tid; i < n; i += kPThreads) s_x[i] = 0.0;
                __syncthreads();
            }
            const unsigned char stop_byte = (unsigned char)(s_flag[2] & 0xFF);   // (workgroup 0; it travels with this trial's chi2)
            UH_BA_CLKT(45);
            if (wv == 0) {   // computeScale's pose part: sum x (lambda x + b)
                double xs = 0;
                for (int i = lane; i < n; i += 64) { const double x = s_x[i]; xs += x * (lambda * x + s_bp[i]); }
                xs = wave_sum_fixed(xs);
                if (lane == 0) s_sc[0] = xs;
            }
            if (wv == 1 && lane < nfree) {   // T_trial = exp(dx) T_cur
                const double* Tc = s_pose + (cur * NF + lane) * 7;
                double qv[4] = {Tc[0], Tc[1], Tc[2], Tc[3]}, tv[3] = {Tc[4], Tc[5], Tc[6]};
                if (ok) se3_left_update_p(qv, tv, s_x + 6 * lane);
                double* To = s_pose + (trial * NF + lane) * 7;
                To[0] = qv[0]; To[1] = qv[1]; To[2] = qv[2]; To[3] = qv[3]; To[4] = tv[0]; To[5] = tv[1]; To[6] = tv[2];
                double Rn[9];
                quat_to_R(qv, Rn);
                double* Ro = s_poseR + (trial * NF + lane) * 12;
#pragma unroll
                for (int i = 0; i < 9; i++) Ro[i] = Rn[i];
                Ro[9] = tv[0]; Ro[10] = tv[1]; Ro[11] = tv[2];
            }
            // ---- back-substitution: dx_l = L^-T (L^-1 b_l - Y_l^T dx_p)
            double t0 = 0, t1 = 0, t2 = 0;
            if (3 * ll + 2 < q.krows && s < nfree) {
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double xa = s_x[6 * s + a];
                    t0 = fma(Yt[(3 * ll) * YS + 6 * s + a], xa, t0);
                    t1 = fma(Yt[(3 * ll + 1) * YS + 6 * s + a], xa, t1);
                    t2 = fma(Yt[(3 * ll + 2) * YS + 6 * s + a], xa, t2);
                }
            }
            t0 = group_sum<NF>(t0); t1 = group_sum<NF>(t1); t2 = group_sum<NF>(t2);
            double scale_part = 0;
            Xt[0] = X[0]; Xt[1] = X[1]; Xt[2] = X[2];
            if (live && any_pt && ok) {
                const double r0 = wl0 - t0, r1 = wl1 - t1, r2 = wl2 - t2;
                const double x2 = r2 * ci22;
                const double x1 = (r1 - cl21 * x2) * ci11;
                const double x0 = (r0 - cl10 * x1 - cl20 * x2) * ci00;
                if (s == 0) scale_part = x0 * (lambda * x0 + bl0) + x1 * (lambda * x1 + bl1) + x2 * (lambda * x2 + bl2);
                Xt[0] += x0; Xt[1] += x1; Xt[2] += x2;
            }
            __syncthreads();   // trial poses complete
            UH_BA_CLKT(46);
            if (q.speculate && ok && st.iteration + 1 < st.max_iters) {
                lambda_spec = lambda * (1. / 3.);
                scale_lane = scale_part; stop_val = stop_byte ? 1.0 : 0.0;
                pend = true;
                UH_BA_CLKT(49);
                continue;
            }
            // ---- trial errors (computeActiveErrors at the trial estimate)
            double chi_part = 0;
            if (has && act) {
                EdgeLin L;
                edge_eval_p<0>(ou, ov, ow, fxk, fyk, cxk, cyk, d.delta, d.dsqr, s_poseR + (trial * NF + s) * 12, Xt, robust, L);
                chi_e = L.chi2;
                chi_part = L.robchi;
            }
            for (int i = fxb + s; i < fxe; i += NF) {
                if (!s_fxact[i]) continue;
                const double* C = s_fxcam + kFxCam * s_fxk[i];
                EdgeLin L;
                edge_eval_p<0>(s_fxobs[3 * i], s_fxobs[3 * i + 1], s_fxobs[3 * i + 2], C[12], C[13], C[14], C[15], d.delta, d.dsqr, C, Xt, robust, L);
                s_fxchi[i] = L.chi2;
                chi_part += L.robchi;
            }
            double cs = chi_part, ss = scale_part;
            block_reduce2<kPWaves, false>(cs, ss, s_red);
            const unsigned tagC = next_tag();
            if (tid == 0) {
                tst(q.partC, 4 * (size_t)g, cs, tagC); tst(q.partC, 4 * (size_t)g + 1, ss, tagC);
                if (g == 0) tst(q.partC, 2, stop_byte ? 1.0 : 0.0, tagC);
            }
            UH_BA_CLKT(47);
            // ---- decision (every wave of every workgroup, same inputs, same code)
            {
                double c = 0, sc = 0;
                bool stopv = false;
                {   // chi2 and scale partials of workgroups lane, lane + 64, ... (G <= 256) and the stop word: ONE batch of loads — three
                    // separate ones were three dependent memory round trips in front of every decision
                    const int cnt = lane < G ? (G - lane + 63) / 64 : 0;
                    long long t0 = 0;
                    for (;;) {
                        TWord wc[4], ws[4];
                        bool okw = true;
#pragma unroll
                        for (int u = 0; u < 4; u++) if (u < cnt) { wc[u] = tld_raw(q.partC, 4 * (size_t)(lane + 64 * u)); ws[u] = tld_raw(q.partC, 4 * (size_t)(lane + 64 * u) + 1); }
                        const TWord wst = tld_raw(q.partC, 2);
                        c = 0; sc = 0;
#pragma unroll
                        for (int u = 0; u < 4; u++) if (u < cnt) { okw = okw && tok(wc[u], tagC) && tok(ws[u], tagC); c += tval(wc[u]); sc += tval(ws[u]); }
                        okw = okw && tok(wst, tagC);
                        stopv = tval(wst) != 0.0;
                        if (okw || give_up(t0)) break;
                    }
                }
                UH_BA_CLKT(48);
                DecideSums sm;
                sm.lin = chi_lin_pass; sm.chi = wave_sum_fixed(c); sm.scale = wave_sum_fixed(sc); sm.xs = s_sc[0];
                st.solve_ok = ok;
                st.pending = 1;
                st = apply_decision(st, sm, stopv);
                if (st.cur != cur) { X[0] = Xt[0]; X[1] = Xt[1]; X[2] = Xt[2]; }
            }
            __syncthreads();   // (a wave that gave up waiting has decided on garbage: everybody leaves together)
            if (s_flag[1]) return;
            UH_BA_CLKT(49);
        }
    }

    UH_BA_CLK(8);
    if (g == 0 && tid == 0) { p.clk[58] = n_kept; p.clk[59] = n_dropped; }
    // ================================================================================ results (GlobalOptimizerG2O::getResults, :466-537)
    // Float poses / points exactly as the reference converts them; a bad association = chi2 > 5.99 or negative depth of the FLOAT
    // point under the FLOAT pose (fixed frames: their input pose).  Two steps, because tens of thousands of scattered system-scope
    // stores serialise on the host link (measured: 1.3 ms for 26k observations): (1) every workgroup writes its part into a result
    // block in HBM with write-through stores; (2) once all have, workgroup g copies slice g of that block to the pinned host block
    // with consecutive lanes on consecutive words — 270 KB leave as ~500 wave-wide posted writes.  uh_ba_get_results is a host copy.
    auto sst = [](auto* dst, auto v) { __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const float Xf0 = (float)X[0], Xf1 = (float)X[1], Xf2 = (float)X[2];
    if (live && s == 0) { sst(q.r_points + 3 * (size_t)l, Xf0); sst(q.r_points + 3 * (size_t)l + 1, Xf1); sst(q.r_points + 3 * (size_t)l + 2, Xf2); }
    if (has) {
        bool bad = chi_e > d.chi2_th;
        if (!bad) {
            const double* Rt = s_poseR + (st.cur * NF + s) * 12;
            const float z = (float)Rt[6] * Xf0 + (float)Rt[7] * Xf1 + (float)Rt[8] * Xf2 + (float)Rt[11];
            if (z < 0) bad = true;
        }
        if (q.want_chi2) sst(q.r_chi2 + eid, chi_e);
        sst(q.r_bad + eid, (unsigned char)bad);
    }
    for (int i = fxb + s; i < fxe; i += NF) {
        const double chi = s_fxchi[i];
        bool bad = chi > d.chi2_th;
        if (!bad) {
            // (row 2 of the fixed frame's input float pose, staged in LDS at the kernel's start: read from HBM here it was two dependent
            // memory round trips per observation in front of every workgroup's result stores)
            const double* M = s_fxcam + kFxCam * s_fxk[i] + 16;
            const float z = (float)M[0] * Xf0 + (float)M[1] * Xf1 + (float)M[2] * Xf2 + (float)M[3];
            if (z < 0) bad = true;
        }
        if (q.want_chi2) sst(q.r_chi2 + s_fxid[i], chi);
        sst(q.r_bad + s_fxid[i], (unsigned char)bad);
    }
    if (g == 0) {   // poses: one element per thread (16 floats of the 4 x 4 matrix + 7 doubles of the se3 state per frame; a thread per frame was 23 stores behind dependent loads)
        for (int t = tid; t < d.K * 23; t += kPThreads) {
            const int k = t / 23, j = t - 23 * k;
            const int sl = p.slot[k];
            if (j < 16) {
                float v;
                if (sl < 0) v = q.poses_in[16 * (size_t)k + j];
                else {
                    const double* Rt = s_poseR + (st.cur * NF + sl) * 12;
                    const int r = j >> 2, c = j & 3;
                    v = r == 3 ? (c == 3 ? 1.f : 0.f) : (float)(c < 3 ? Rt[r * 3 + c] : Rt[9 + r]);
                }
                sst(q.r_poses + 16 * (size_t)k + j, v);
            } else {
                const int i7 = j - 16;
                sst(q.r_state + 7 * (size_t)k + i7, sl < 0 ? q.pose0[7 * k + i7] : s_pose[(st.cur * NF + sl) * 7 + i7]);
            }
        }
    }
    // (1) -> (2): every workgroup's stores have been acknowledged before it is counted; everybody waits for the full count
    UH_BA_CLK(11);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    UH_BA_CLK(12);
    if (tid == 0) {
        __hip_atomic_fetch_add(q.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long t0 = 0;
        while (__hip_atomic_load(q.done_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - q.done_base < (unsigned)G) if (give_up(t0)) break;
    }
    __syncthreads();
    if (s_flag[1]) return;
    UH_BA_CLK(9);
    {
        const size_t per = (q.r_words + G - 1) / G, w0 = (size_t)g * per, w1 = w0 + per < q.r_words ? w0 + per : q.r_words;
        for (size_t w = w0 + tid; w < w1; w += 4 * kPThreads) {   // four words in flight per lane
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (w + u * kPThreads < w1) v[u] = __hip_atomic_load(q.r_dev + w + u * kPThreads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 4; u++) if (w + u * kPThreads < w1) __hip_atomic_store(q.r_host + w + u * kPThreads, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // the workgroup that completes the second count publishes the final state and the completion word the host polls (posted writes
    // of one device arrive in order)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    UH_BA_CLK(10);
    if (tid == 0) {
        // release (this workgroup's stores into pinned memory) / acquire (everybody else's, for the one that posts the completion word) at
        // SYSTEM scope: the counter is the only thing that orders the other workgroups' result stores before the completion word
        const unsigned before = __hip_atomic_fetch_add(q.done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (before + 1u - q.done_base == 2u * (unsigned)G) {
            BAState fin = st; fin.cur = 0; fin.pending = 0; p.st[0] = fin; p.st[1] = fin;
            if (q.host_state) {   // straight into pinned host memory: uh_ba_optimize polls host_done instead of synchronising the stream
                *q.host_state = fin;
                __hip_atomic_store(q.host_done, ((unsigned long long)q.launch_id << 32) | 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
