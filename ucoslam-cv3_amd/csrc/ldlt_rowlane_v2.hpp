// Round-5 forms of the persistent local BA's redundant solve (included by ba.hip behind ldlt_rowlane_lds / backsolve_lds, whose helpers
// it uses).  Measured with scripts/micro/solve48.hip and scripts/micro/chain_latency.hip on MI355X (one wave per SIMD): a dependent fp64
// VALU instruction costs 8.6 clocks, v_rcp_f64 20.5, a v_readlane pair in front of its use 16, an LDS round trip 70-80, a barrier 10.
#pragma once

// L^T x = z for the bordered row-per-lane factorisation (n + 1 <= 64 rows; z = D^-1 L^-1 b is row n of M).  Wave 0 only, no barrier
// inside: x_i lives in lane i; column j = n-1 .. 1: x_j is final, it is broadcast with a v_readlane pair and every lane below subtracts
// L[j][i] x_j — ONE link of v_readlane -> v_fma per column (24.7 clocks measured), nothing else on the chain:
//   * the column index is a compile-time constant (the loop is unrolled over the NB block columns): immediate lane selects, no
//     scalar arithmetic between the links;
//   * the mask "lane i < j" is applied to the ADDRESS of the entry, not to its value: a lane on or above the diagonal reads a zero
//     word instead of what the factorisation left there (one v_cndmask_b32 with a constant lane mask per entry, against two on the
//     value), so x_i stays what it was when it was broadcast and leaves through its own lane at the end;
//   * the entries are fetched twelve columns ahead of their use, one select + one ds_read_b64 in the shadow of every link's FMA
//     (the persistent kernel has no registers for all 47 of a lane's entries: a first form that loaded them up front went to scratch
//     memory there — 4.5 us against 1.0 us in scripts/micro/solve48.hip).
// backsolve_lds (the round-2 form: run-time column index, masked values, eight-column batches) took 120 clocks per column.
template <int NB>
__device__ __forceinline__ void backsolve_rowlane_t(const double* M, int ld, double* s_x, const double* s_zero) {
    constexpr int n = 6 * NB, AHEAD = 12;
    int lane = threadIdx.x & 63;
    // (the lane index goes through an opaque move: the 47 entry addresses are then computed here, every call — left visible they are
    // loop-invariant in the caller's trial loop and get hoisted out of it, 47 registers held across the whole kernel)
    asm volatile("" : "+v"(lane));
    const int lr = lane < n ? lane : n - 1;
    double x = M[n * ld + lr];
    double l[n > 0 ? n : 1];
    // entry L[j][lane] for lane < j, the zero word otherwise
    auto fetch = [&](int j) -> double {
        const double* p = lane < j ? M + j * ld + lane : s_zero;
        return *p;
    };
#pragma unroll
    for (int j = n - 1; j >= 1 && j > n - 1 - AHEAD; j--) l[j] = fetch(j);
#pragma unroll
    for (int j = n - 1; j >= 1; j--) {
        const double xj = readlane_f64(x, j);
        x = fma(-l[j], xj, x);
        if (j - AHEAD >= 1) l[j - AHEAD] = fetch(j - AHEAD);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane < n) s_x[lane] = x;
}
// s_zero: one LDS word holding 0.0 (outside M)
__device__ __forceinline__ void backsolve_v2(const double* M, int n, int ld, double* s_x, const double* s_zero) {
    if (threadIdx.x >= 64) return;
    switch (n / 6) {
        case 1: backsolve_rowlane_t<1>(M, ld, s_x, s_zero); break;
        case 2: backsolve_rowlane_t<2>(M, ld, s_x, s_zero); break;
        case 3: backsolve_rowlane_t<3>(M, ld, s_x, s_zero); break;
        case 4: backsolve_rowlane_t<4>(M, ld, s_x, s_zero); break;
        case 5: backsolve_rowlane_t<5>(M, ld, s_x, s_zero); break;
        case 6: backsolve_rowlane_t<6>(M, ld, s_x, s_zero); break;
        case 7: backsolve_rowlane_t<7>(M, ld, s_x, s_zero); break;
        case 8: backsolve_rowlane_t<8>(M, ld, s_x, s_zero); break;
        case 9: backsolve_rowlane_t<9>(M, ld, s_x, s_zero); break;
        case 10: backsolve_rowlane_t<10>(M, ld, s_x, s_zero); break;
        default: break;
    }
}
// Bordered LDL^T, row per lane (n + 1 <= 64 rows, n = 6 nb) — the round-5 form of ldlt_rowlane_lds: the same algorithm and hand-offs
// (wave 0 owns the pivots, block column by block column; the other waves apply panel kb to everything behind block column kb + 1; one
// barrier per block column), with wave 0's step cut down to what the pivot chain needs:
//   * the two roles run their own loops with a scalar block counter (a wave-uniform branch on the wave index: s_cbranch, the block
//     offsets stay in SGPRs — the round-3 form carried k0 in a VGPR and scalarised it with six v_readfirstlane per step);
//   * D is not kept (nobody reads it: the right-hand side row ends as D^-1 L^-1 b): the twelve selects, twelve moves and six compares per
//     block column that put d_j on the diagonal are gone — every lane stores its l_j unconditionally;
//   * the panel application runs columns 0-2 first: the first pivot's column is ready after 18 FMAs, the other 18 fill the stalls of
//     the first pivots' reciprocal chains.
// M ends up holding L below the diagonal (garbage on and above it) and D^-1 L^-1 b in row n.
template <bool FINAL_BARRIER = true>   // false: no barrier behind the last block column — the caller's wave 0 substitutes at once (its own data)
__device__ __forceinline__ bool ldlt_rowlane_v2(double* M, int n, int ld, int nfree, int npairs, const short (*s_pair)[2], double* s_y_raw) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = n / 6, nrow = n + 1;
    double* const s_y = s_y_raw + ((reinterpret_cast<uintptr_t>(s_y_raw) >> 3) & 1);   // [2][6][64], 16-byte aligned
    bool failed = false;
    if (wvu == 0) {
        const int r = lane < nrow ? lane : nrow - 1;   // idle lanes shadow the last row
        double* const Mr = M + r * ld;
        double lprev[6] = {0, 0, 0, 0, 0, 0};
        // A zero or non-finite pivot is detected ONCE, behind the last one: d = 0 gives r0 = inf and e = NaN, d = inf gives r0 = 0 and e = NaN,
        // a NaN stays one — and the refined reciprocal carries it into the next pivot's column (lane k0 + j + 1 takes a[j] * l * ikj), so the
        // LAST reciprocal is non-finite exactly when some pivot was bad.  (Rounds 3-5a tested every pivot: a class test, a compare and two
        // scalar ORs on wave 0's issue-bound step, 48 times.)
        double ikl = 1.0;
        for (int kb = 0; kb < nb; kb++) {
            const int k0 = 6 * kb;
            double a[6];
#pragma unroll
            for (int j = 0; j < 6; j++) a[j] = Mr[k0 + j];
            if (kb > 0) {
                const double* yb = s_y + ((kb - 1) & 1) * 384 + k0;
                double2 yv[6][3];
#pragma unroll
                for (int t = 0; t < 6; t++)
#pragma unroll
                    for (int h = 0; h < 3; h++) yv[t][h] = *reinterpret_cast<const double2*>(yb + t * 64 + 2 * h);
#pragma unroll
                for (int t = 0; t < 6; t++) { a[0] = fma(-lprev[t], yv[t][0].x, a[0]); a[1] = fma(-lprev[t], yv[t][0].y, a[1]); a[2] = fma(-lprev[t], yv[t][1].x, a[2]); }
#pragma unroll
                for (int t = 0; t < 6; t++) { a[3] = fma(-lprev[t], yv[t][1].y, a[3]); a[4] = fma(-lprev[t], yv[t][2].x, a[4]); a[5] = fma(-lprev[t], yv[t][2].y, a[5]); }
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const double dj = readlane_f64(a[j], k0 + j);
                // 1/d: v_rcp_f64 (2^-23 relative) and ONE cubic step r0 (1 + e + e^2), e = 1 - d r0: error e^3 = 2^-69.  (Tried: the next pivot's
                // column as (a - p r0) - (p r0)(e + e^2), one dependent operation less on the chain — two instructions more, 7 % slower.)
                const double r0 = __builtin_amdgcn_rcp(dj);
                const double e = fma(-dj, r0, 1.0);
                const double ikj = fma(fma(e, e, e), r0, r0);
                ikl = ikj;
                lprev[j] = a[j] * ikj;
                if (j + 1 < 6) a[j + 1] = fma(-(a[j] * readlane_f64(a[j], k0 + j + 1)), ikj, a[j + 1]);
#pragma unroll
                for (int c = j + 2; c < 6; c++) a[c] = fma(-lprev[j], readlane_f64(a[j], k0 + c), a[c]);
            }
            double* yo = s_y + (kb & 1) * 384 + lane;
#pragma unroll
            for (int j = 0; j < 6; j++) { yo[j * 64] = a[j]; Mr[k0 + j] = lprev[j]; }
            if (FINAL_BARRIER || kb + 1 < nb) __syncthreads();   // panel kb is in M / s_y[kb & 1]; the trailing update of panel kb - 1 is complete
        }
        failed = !isfinite(ikl);
    } else {
        const int nthr = (int)blockDim.x - 64;
        for (int kb = 0; kb < (FINAL_BARRIER ? nb : nb - 1); kb++) {
            __syncthreads();
            const int k0 = 6 * kb, J0 = kb + 2;
            if (J0 >= nb) continue;
            const int tile0 = J0 * nfree - J0 * (J0 - 1) / 2;   // first pair with s1 >= J0 in the s1-major pair list
            const int ntile = npairs - tile0;
            const int nunits = 6 * ntile + (nb - J0);
            const double* yb = s_y + (kb & 1) * 384;
            for (int u = tid - 64; u < nunits; u += nthr) {
                int rr, c0;
                bool diag = false;
                if (u < 6 * ntile) {
                    const int tile = u / 6, s1 = s_pair[tile0 + tile][0], s2 = s_pair[tile0 + tile][1];
                    rr = 6 * s2 + (u - 6 * tile); c0 = 6 * s1; diag = s1 == s2;
                } else { rr = n; c0 = 6 * (J0 + (u - 6 * ntile)); }
                double lr[6], acc[6];
#pragma unroll
                for (int t = 0; t < 6; t++) lr[t] = M[rr * ld + k0 + t];
#pragma unroll
                for (int j = 0; j < 6; j++) acc[j] = M[rr * ld + c0 + j];
#pragma unroll
                for (int t = 0; t < 6; t++) {
                    const double2 y01 = *reinterpret_cast<const double2*>(yb + t * 64 + c0), y23 = *reinterpret_cast<const double2*>(yb + t * 64 + c0 + 2),
                                  y45 = *reinterpret_cast<const double2*>(yb + t * 64 + c0 + 4);
                    acc[0] = fma(-lr[t], y01.x, acc[0]); acc[1] = fma(-lr[t], y01.y, acc[1]); acc[2] = fma(-lr[t], y23.x, acc[2]);
                    acc[3] = fma(-lr[t], y23.y, acc[3]); acc[4] = fma(-lr[t], y45.x, acc[4]); acc[5] = fma(-lr[t], y45.y, acc[5]);
                }
                // (no mask on the diagonal tiles: what lands above the diagonal is never read)
                (void)diag;
#pragma unroll
                for (int j = 0; j < 6; j++) M[rr * ld + c0 + j] = acc[j];
            }
        }
    }
    return failed;
}
