// Solve of the reduced camera system for 22-32 free keyframes (132 <= n <= 192; any n + 1 <= blockDim.x works): one workgroup, the system in LDS
// (included by ba.hip inside its anonymous namespace, behind fast_rcp).
//
// What it replaces: g2o hands Hschur to Eigen::SimplicialLDLT (g2o/solvers/eigen/linear_solver_eigen.h:92-120; natural ordering of a
// dense block matrix).  Round 3 factorised these sizes by 6-column block steps whose rank-6 trailing update went one (row, 6-column
// tile) per thread through LDS — 48 operand reads per 36 FMAs — and substituted on one wave (packed form: 40 us of substitution beside
// 70 us of factorisation at 24 free keyframes; row-per-lane form on wave 0: 29 us at 17).
//
// This form factorises the BORDERED system [S; b^T] (the right-hand side is row n, so z = D^-1 L^-1 b falls out of the factorisation
// and no forward substitution is run):
//   * PANEL, every thread: thread r owns row r.  The 6 x 6 diagonal block of a block column is factorised redundantly by every
//     thread from 21 broadcast reads; a thread then solves its own row against it (15 FMAs).  Block columns go in PAIRS: the second
//     panel of a pair takes the first one's rank-6 contribution from registers (the row's own L entries) and a 6 x 6 block of
//     y = L d handed through LDS;
//   * TRAILING UPDATE, once per pair: S(r, c) -= sum_t L(r, t) d_t L(c, t) over the pair's 12 columns as three
//     v_mfma_f64_16x16x4_f64 per 16 x 16 tile of the remaining triangle — two operand reads per 1024 FMAs.  The B operand is formed
//     as L * d on the way in (no y panel in LDS: a 32-keyframe packed triangle is 148 KB of the 160);
//   * BACKWARD SUBSTITUTION L^T x = z by block rows: x of a block is solved redundantly by every thread, thread r < block takes its six
//     terms from six rows of L (consecutive threads read consecutive words), one barrier per block.
// PACKED: row r at r (r + 1) / 2 (lower triangle and diagonal only).  !PACKED: row stride ld (odd).  Both hold n + 1 rows.
#pragma once

typedef double ldlt_f64x4 __attribute__((ext_vector_type(4)));
#ifndef UH_LDLTM_CLK
#define UH_LDLTM_CLK(i)   // scripts/micro/ldlt_mfma_time.hip stamps thread 0's phases through this hook
#endif

constexpr int kLdltAux = 64;   // doubles of LDS scratch the solve needs beside the matrix

// 6 x 6 diagonal block at (k0, k0), read from M by every thread (broadcast) and factorised in registers: unit lower L (strict part),
// d, 1 / d.  Returns whether a pivot was zero / non-finite.
struct DiagBlock { double l[6][6], d[6], ik[6]; };
template <typename IXF>
__device__ __forceinline__ bool factor_diag6(const double* M, IXF IX, int k0, DiagBlock& o) {
    double a[6][6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int c = 0; c <= i; c++) a[i][c] = M[IX(k0 + i, k0 + c)];
    bool failed = false;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        o.d[j] = a[j][j];
        failed = failed || o.d[j] == 0.0 || !isfinite(o.d[j]);
        o.ik[j] = fast_rcp(o.d[j]);
#pragma unroll
        for (int i = j + 1; i < 6; i++) o.l[i][j] = a[i][j] * o.ik[j];
#pragma unroll
        for (int i = j + 1; i < 6; i++)
#pragma unroll
            for (int c = j + 1; c <= i; c++) a[i][c] = fma(-o.l[i][j], a[c][j], a[i][c]);   // (a[c][j] keeps L d_j)
    }
    return failed;
}

// Every thread of the workgroup calls it (it contains barriers; blockDim.x >= n + 1, a multiple of 64).  M: rows 0 .. n, row n = b.
// On return s_x[0 .. n) = x (valid after the caller's next barrier) and M holds L (strictly below the diagonal), D on the diagonal,
// z in row n.  Returns (in the waves that hold rows, thread 0 among them) whether a pivot was zero / non-finite — the caller publishes it.
template <bool PACKED>
__device__ __forceinline__ bool ldlt_solve_mfma_lds(double* M, int n, int ld, double* s_aux, double* s_x) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    auto IX = [&](int r, int c) -> int { if constexpr (PACKED) return (__mul24(r, r + 1) >> 1) + c; else return __mul24(r, ld) + c; };
    double* const s_d = s_aux;          // [12]: the pair's pivots
    double* const s_ys = s_aux + 12;    // [6][6]: y(t, q) = L d of the pair's first panel at row k0 + 6 + q
    double* const s_z = s_aux + 48;     // [2][6]: the substitution's hand-over
    const int nb = n / 6, nrow = n + 1;
    const int r = tid < nrow ? tid : nrow - 1;   // (threads beyond the last row compute along with it and store nothing)
    const bool own = tid < nrow;
    const int rbase = IX(r, 0);
    bool failed = false;

    // Trailing update by MACRO TILES of 32 rows x 16 columns (two MFMA tiles that share the B operand and the address arithmetic;
    // their MFMAs alternate, so none waits for the one before it).  The tile coordinates are scalar (the wave index comes through
    // v_readfirstlane), so an INTERIOR macro tile — strictly below the diagonal tiles, no ragged edge — takes a path without masks,
    // clamps or exec juggling.  Why the care: gfx950 runs the f64 MFMA at 70 clocks behind a 230-clock start-up and does not overlap
    // it with vector work (scripts/micro/mfma_f64_burst.hip), so with one wave per SIMD the update is bound by the instructions
    // around the MFMAs: 80 per 16 x 16 tile in the first form of this loop.
    const int i16 = lane & 15, kq = lane >> 4;
    // the word that absent elements are redirected to: in the solve's LDS scratch when the packed triangle lives there; the row-stride
    // form (which may live in HBM) has a free word of its own, (0, n) — column n of a row above the border row
    const int dumpix = PACKED ? (int)(s_aux + 60 - M) : n;
    auto macro_tile = [&](int k0, int rb, int cb, const double (&dq)[3]) {
        const bool interior = rb >= cb + 16 && cb + 16 <= n;   // (uniform; the tiles are anchored at the last row: rb + 32 <= nrow always)
        const int cc = cb + i16, rw0 = rb + kq;
        const int i0 = IX(rw0, cc);
        int iv[8];
#pragma unroll
        for (int v = 0; v < 8; v++) iv[v] = PACKED ? i0 + 4 * v * rw0 + 2 * v * (4 * v + 1) : i0 + 4 * v * ld;   // tri(rw0 + 4 v) = tri(rw0) + 4 v rw0 + 2 v (4 v + 1)
        double a0[3], a1[3], b[3], m[8];
        ldlt_f64x4 acc0, acc1;
        if (interior) {
            const int ab0 = IX(rb + i16, k0 + kq), ab1 = IX(rb + 16 + i16, k0 + kq), bb = IX(cc, k0 + kq);
#pragma unroll
            for (int s = 0; s < 3; s++) { a0[s] = M[ab0 + 4 * s]; a1[s] = M[ab1 + 4 * s]; b[s] = M[bb + 4 * s]; }
#pragma unroll
            for (int v = 0; v < 8; v++) m[v] = M[iv[v]];
#pragma unroll
            for (int s = 0; s < 3; s++) { a0[s] = -a0[s]; a1[s] = -a1[s]; b[s] *= dq[s]; }
#pragma unroll
            for (int v = 0; v < 4; v++) { acc0[v] = m[v]; acc1[v] = m[4 + v]; }
#pragma unroll
            for (int s = 0; s < 3; s++) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b[s], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; v++) { M[iv[v]] = acc0[v]; M[iv[4 + v]] = acc1[v]; }
        } else {
            // a diagonal or ragged macro tile: an element that does not exist (above the diagonal, past the last row or column) is
            // REDIRECTED to a dump word in the scratch block — its lane loads, accumulates and stores like every other, nobody reads
            // the word — so the tile needs eight index selects and no select on data, no exec mask, no branch.
            // exists(v)  <=>  cc <= rw < nrow and cc < n  <=>  (unsigned)(rw - cc) < (cc < n ? nrow - cc : 0),  rw - cc = d0 + 4 v
            const int ar0 = rb + i16 > 0 ? rb + i16 : 0, ar1 = rb + 16 + i16 > 0 ? rb + 16 + i16 : 0, br = cc < n ? cc : n - 1;   // (rows above the diagonal only feed redirected elements)
            const int ab0 = IX(ar0, k0 + kq), ab1 = IX(ar1, k0 + kq), bb = IX(br, k0 + kq);
#pragma unroll
            for (int s = 0; s < 3; s++) { a0[s] = M[ab0 + 4 * s]; a1[s] = M[ab1 + 4 * s]; b[s] = M[bb + 4 * s]; }
            const unsigned lim = cc < n ? (unsigned)(nrow - cc) : 0u;
            const int d0 = rw0 - cc;
#pragma unroll
            for (int v = 0; v < 8; v++) iv[v] = (unsigned)(d0 + 4 * v) < lim ? iv[v] : dumpix;
#pragma unroll
            for (int v = 0; v < 4; v++) { acc0[v] = M[iv[v]]; acc1[v] = M[iv[4 + v]]; }
#pragma unroll
            for (int s = 0; s < 3; s++) { a0[s] = -a0[s]; a1[s] = -a1[s]; b[s] *= dq[s]; }
#pragma unroll
            for (int s = 0; s < 3; s++) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b[s], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; v++) { M[iv[v]] = acc0[v]; M[iv[4 + v]] = acc1[v]; }
        }
    };
    auto trailing = [&](int k0) {
        const int c0 = k0 + 12;
        if (c0 >= nrow) return;
        const int TC = (n - c0 + 15) >> 4;            // tile columns that hold a column of S (the border row has none of its own)
        double dq[3];
#pragma unroll
        for (int s = 0; s < 3; s++) dq[s] = s_d[4 * s + kq];
        // macro tiles: tile column tj (columns cb ..) holds ceil((nrow - cb) / 32) of them, anchored at the LAST row, so that only the top
        // one of a column — the one that meets the diagonal, and may start above it — takes the path with redirected elements
        int tj = 0, m0 = 0;
        for (int m = wv;; m += nw) {
            while (tj < TC && m0 + ((nrow - c0 - 16 * tj + 31) >> 5) <= m) { m0 += (nrow - c0 - 16 * tj + 31) >> 5; ++tj; }
            if (tj >= TC) break;
            macro_tile(k0, nrow - 32 * (m - m0 + 1), c0 + 16 * tj, dq);
        }
    };

    // own row against a factorised diagonal block: y (= L d, in place) and the row's L entries
    auto row_solve = [&](const DiagBlock& D, double (&y)[6], double (&lrow)[6]) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
#pragma unroll
            for (int t = 0; t < j; t++) y[j] = fma(-y[t], D.l[j][t], y[j]);
            lrow[j] = y[j] * D.ik[j];
        }
    };
    // A thread whose row lies IN the block (row k0 + q) runs the same row_solve as the rows below: its entries left of the diagonal
    // come out as L, the diagonal one as d.  It stores them only behind a barrier that every thread's factor_diag6 of that block has
    // passed (the block is read from M), and by static register indices: indexing the DiagBlock by q put it into scratch memory —
    // two HBM round trips per panel.
    auto store_own_diag = [&](const double (&y)[6], const double (&lrow)[6], int k0) {
        const int q = r - k0;
        if (q >= 0 && q < 6) {
#pragma unroll
            for (int j = 0; j < 6; j++) { if (j < q) M[rbase + k0 + j] = lrow[j]; else if (j == q) M[rbase + k0 + j] = y[j]; }
        }
    };
    auto store_pivots = [&](const DiagBlock& D, int half) {
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) s_d[half * 6 + j] = D.d[j];
        }
    };

    // waves that hold no row (a 512-thread workgroup on a 193-row system has four of them: they are there for the trailing update)
    // skip the panels' arithmetic — it would be redundant work on the SIMDs the row-holding waves run on — and only keep the barriers
    const bool pw = wv * 64 < nrow;
    for (int kb0 = 0; kb0 < nb; kb0 += 2) {
        const int k0 = 6 * kb0;
        const bool two = kb0 + 1 < nb;
        UH_LDLTM_CLK(0);
        // ---- first panel of the pair
        DiagBlock D;
        double y1[6], l1[6], y2[6], l2[6];
        const bool below1 = r >= k0 + 6;
        if (pw) {
            failed = factor_diag6(M, IX, k0, D) || failed;
#pragma unroll
            for (int j = 0; j < 6; j++) y1[j] = r >= k0 ? M[rbase + k0 + j] : 0.0;   // (a row IN the block: the entries right of its diagonal are never used)
            row_solve(D, y1, l1);
            if (below1 && own) {
#pragma unroll
                for (int j = 0; j < 6; j++) M[rbase + k0 + j] = l1[j];
                const int q = r - (k0 + 6);
                if (q < 6 && two) {
#pragma unroll
                    for (int t = 0; t < 6; t++) s_ys[t * 6 + q] = y1[t];
                }
            }
            store_pivots(D, 0);
        }
        UH_LDLTM_CLK(1);
        __syncthreads();   // A: s_ys, the first panel's L
        if (pw) store_own_diag(y1, l1, k0);
        if (!two) break;   // (an odd last block column: nothing lies behind it)
        // ---- the first panel's rank-6 contribution to the second block column, own row
        if (pw) {
            double yv[6][6];
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int h = 0; h < 3; h++) { const double2 w = *reinterpret_cast<const double2*>(s_ys + t * 6 + 2 * h); yv[t][2 * h] = w.x; yv[t][2 * h + 1] = w.y; }
            // (PACKED: a row of the second diagonal block reads past its own end for c > q — the next rows' words, never used)
#pragma unroll
            for (int c = 0; c < 6; c++) y2[c] = below1 ? M[rbase + k0 + 6 + c] : 0.0;
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int c = 0; c < 6; c++) y2[c] = fma(-l1[t], yv[t][c], y2[c]);
            const int q = r - (k0 + 6);
            if (q >= 0 && q < 6) {   // the second diagonal block goes back to M for everybody to read
#pragma unroll
                for (int c = 0; c < 6; c++) if (c <= q) M[rbase + k0 + 6 + c] = y2[c];
            }
        }
        UH_LDLTM_CLK(2);
        __syncthreads();   // B: the second diagonal block
        if (pw) {
            failed = factor_diag6(M, IX, k0 + 6, D) || failed;
            const bool below2 = r >= k0 + 12;
            row_solve(D, y2, l2);
            if (below2 && own) {
#pragma unroll
                for (int j = 0; j < 6; j++) M[rbase + k0 + 6 + j] = l2[j];
            }
            store_pivots(D, 1);
        }
        UH_LDLTM_CLK(3);
        __syncthreads();   // C: both panels and their pivots
        if (pw) store_own_diag(y2, l2, k0 + 6);   // (nothing of the trailing update reads these rows' block)
        // ---- trailing update behind the pair
        trailing(k0);
        UH_LDLTM_CLK(4);
        __syncthreads();   // D
    }
    UH_LDLTM_CLK(5);
    // ---- L^T x = z (row n), block rows from the last to the first; a block's L entries are fetched one block ahead (they are final)
    double z = tid < n ? M[IX(n, tid)] : 0.0;
    if (tid >= n - 6 && tid < n) s_z[((nb - 1) & 1) * 6 + tid - (n - 6)] = z;
    double lb[6][6], lr[6];
    auto fetch = [&](int kb) {
        const int k0 = 6 * kb;
#pragma unroll
        for (int i = 1; i < 6; i++)
#pragma unroll
            for (int c = 0; c < i; c++) lb[i][c] = M[IX(k0 + i, k0 + c)];
#pragma unroll
        for (int t = 0; t < 6; t++) lr[t] = tid < k0 ? M[IX(k0 + t, tid)] : 0.0;
    };
    __syncthreads();
    fetch(nb - 1);
    const bool sw = wv * 64 < n;   // (a wave without an unknown: barriers only)
    for (int kb = nb - 1; kb >= 0; kb--) {
        if (!sw) { __syncthreads(); continue; }
        const int k0 = 6 * kb;
        double x[6];
#pragma unroll
        for (int i = 0; i < 6; i++) x[i] = s_z[(kb & 1) * 6 + i];
#pragma unroll
        for (int j = 4; j >= 0; j--)
#pragma unroll
            for (int i = j + 1; i < 6; i++) x[j] = fma(-lb[i][j], x[i], x[j]);
#pragma unroll
        for (int t = 0; t < 6; t++) z = fma(-lr[t], x[t], z);
        if (tid >= k0 && tid < k0 + 6) {
#pragma unroll
            for (int i = 0; i < 6; i++) if (i == tid - k0) s_x[tid] = x[i];
        }
        if (tid >= k0 - 6 && tid < k0) s_z[((kb - 1) & 1) * 6 + tid - (k0 - 6)] = z;
        if (kb > 0) fetch(kb - 1);
        __syncthreads();
    }
    UH_LDLTM_CLK(6);
    return failed;
}
