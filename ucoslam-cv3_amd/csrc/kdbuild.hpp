// picoflann's KdTreeIndex<2>::build on the device: ONE workgroup builds the tree of a frame's undistorted keypoints in LDS,
// node for node and leaf for leaf what src/basictypes/picoflann.h:150-163,238-345 builds on the CPU (Frame::create_kdtree,
// src/utils/frameextractor.cpp:4258 / map_types/frame.h:124) and what the host restatement KdBuilder (projmatch.hip) builds.
//
// The recursion becomes a level-synchronous sweep (every node of a depth at once):
//   moments      mean / variance of both coordinates over <= ~100 evenly spaced samples, double sums IN SAMPLE ORDER (a rounding per
//                addition: the order is part of the result) — four lanes per node, one serial chain each
//   Hoare passes [< cut | == cut | > cut] exactly as picoflann's two scans permute the points (picoflann.h:403-424).  One pass = "with
//                m points belonging left, the misplaced points in front of b + m (ascending) are exchanged pairwise with the misplaced
//                points behind (descending)"; the ranks come from the wave ballots of the predicate over rows of 64 consecutive points and
//                one prefix sum over the rows (segments are contiguous: a node's count is a difference of two prefix values), the pairs
//                meet through two index lists.  The second pass moves nothing unless a point EQUALS its node's cut: it is skipped then
//   std::sort    where picoflann falls back to it (a side of the mean split would hold < 10 points — always for 11..19 points): libstdc++'s
//                introsort = a partitioning phase (median of three, unguarded Hoare partition, heapsort at the depth limit; serial, one lane
//                per node, nothing to do up to 16 points) followed by an insertion sort, which is a STABLE sort of what the first phase
//                left — computed here as a rank (smaller keys + equal keys in front) by every point in parallel
// The first levels are swept by the whole workgroup; once a level holds one node per wave every wave takes a subtree and runs the same
// sweep alone (no workgroup barrier below that point).  Boxes: picoflann's per-node box is only observable through divhigh (= the tight
// lower bound of the right subtree) and the root box; lower bounds and the counts of inner nodes climb from the leaves (the second child
// to arrive at a parent goes on), the root's upper bounds are the maximum over the leaves / cuts that no ancestor's `lbox.hi[dim] = cut`
// overrides.  Node numbers are picoflann's (children of the r-th split in depth-first order are 2r + 1, 2r + 2).
#pragma once
#include <cstdint>

#include "introselect.hpp"   // UH_HD, uh_sel::floor_log2, wave_mem_sync

namespace uh_kd {

constexpr int kLeafMax = 10;        // picoflann _maxLeafSize
constexpr int kDevMaxPoints = 4096; // one workgroup's LDS holds the points and the nodes up to here (a wave's lanes own <= 64 points each)

struct Elem { float x, y; uint32_t id; };

// ---------------------------------------------------------------------------------------------- libstdc++ std::sort, first phase
// GCC 11 <bits/stl_algo.h>: std::sort = __introsort_loop(first, last, 2 * lg(n)) + __final_insertion_sort.  The accessor A moves whole
// points: key(i), get(i), set(i, e), swap(i, j); comp(a, b) = key(a) < key(b).
template <class A> UH_HD void median_to_first(A& a, int result, int x, int y, int z) {
    if (a.key(x) < a.key(y)) {
        if (a.key(y) < a.key(z)) a.swap(result, y);
        else if (a.key(x) < a.key(z)) a.swap(result, z);
        else a.swap(result, x);
    } else if (a.key(x) < a.key(z)) a.swap(result, x);
    else if (a.key(y) < a.key(z)) a.swap(result, z);
    else a.swap(result, y);
}

template <class A> UH_HD int unguarded_partition(A& a, int first, int last, int pivot) {
    const float pv = a.key(pivot);   // (the pivot slot lies outside [first, last))
    for (;;) {
        while (a.key(first) < pv) ++first;
        --last;
        while (pv < a.key(last)) --last;
        if (!(first < last)) return first;
        a.swap(first, last);
        ++first;
    }
}

// std::__adjust_heap + std::__push_heap
template <class A> UH_HD void adjust_heap(A& a, int first, int hole, int len, const Elem& value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (a.key(first + child) < a.key(first + child - 1)) child--;
        a.set(first + hole, a.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.set(first + hole, a.get(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && a.key(first + parent) < a.ekey(value)) {
        a.set(first + hole, a.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.set(first + hole, value);
}

// std::__partial_sort(first, last, last): __heap_select (= __make_heap, nothing behind middle) + __sort_heap
template <class A> UH_HD void heap_sort(A& a, int first, int last) {
    const int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const Elem value = a.get(first + parent);
            adjust_heap(a, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {   // __pop_heap(first, last, last)
        --last;
        const Elem value = a.get(last);
        a.set(last, a.get(first));
        adjust_heap(a, first, 0, last - first, value);
    }
}

// __introsort_loop: the recursion on the right part becomes a stack (disjoint ranges: the order in which they are processed does not
// change the data movement inside either).  What is left for the insertion sort that follows is "a stable sort of this".
template <class A> UH_HD void sort_phase(A& a, int first, int last) {
    if (last - first <= 16) return;
    struct Fr { int first, last, depth; };
    Fr st[40];   // depth limits strictly decrease from the bottom of the stack to its top: <= 2 lg(n) + 1 entries
    int sp = 0;
    st[sp++] = Fr{first, last, 2 * uh_sel::floor_log2(last - first)};
    while (sp) {
        const Fr f = st[--sp];
        int fi = f.first, la = f.last, dl = f.depth;
        while (la - fi > 16) {
            if (dl == 0) { heap_sort(a, fi, la); break; }
            --dl;
            const int mid = fi + (la - fi) / 2;
            median_to_first(a, fi, fi + 1, mid, la - 1);
            const int cut = unguarded_partition(a, fi + 1, la, fi);
            st[sp++] = Fr{cut, la, dl};
            la = cut;
        }
    }
}

// LDS need of kd_build_workgroup for up to n_cap points and nwaves waves (host and device agree on the layout through this)
UH_HD int node_cap(int n_cap, int nwaves) { return 2 * n_cap / 5 + 4 * nwaves + 8; }
UH_HD int row_cap(int n_cap, int nwaves) { return n_cap / 64 + 2 * nwaves + 4; }
UH_HD size_t lds_bytes(int n_cap, int nwaves) {
    const size_t n = (size_t)n_cap + 2, m = (size_t)node_cap(n_cap, nwaves);
    size_t b = m * 8 + (size_t)row_cap(n_cap, nwaves) * 8;   // ncut, bal
    b += n * 4 * 3;                   // px, py, tmp (the scratch of the fallback permutation)
    b += m * 4 * 6;                   // nlo[2], ndivhigh, nlim, nbe, ncutf
    b += n * 2 * 3;                   // ord, eseg, scr
    b += m * 2 * 4;                   // nchild, npar, ncnt, nmid
    b += m;                           // nflag
    return (b + 64 + 15) & ~(size_t)15;
}

}  // namespace uh_kd

#if defined(__HIPCC__)
namespace uh_kd {

struct Lds {
    double* ncut; unsigned long long* bal;
    float* px; float* py; unsigned* tmp;
    float* nlo; float* ndivhigh; unsigned* nlim; unsigned* nbe; float* ncutf;
    unsigned short* ord; unsigned short* eseg; unsigned short* scr;
    unsigned short* nchild; unsigned short* npar; unsigned short* ncnt; unsigned short* nmid;
    unsigned char* nflag;   // bit 0 split dimension, bit 1 std::sort fallback taken, bits 2-3 "an ancestor's cut overrides my upper bound in x / y"
    int m_cap;
};

__device__ __forceinline__ Lds carve(unsigned char* base, int n_cap, int nwaves) {
    const size_t n = (size_t)n_cap + 2, m = (size_t)node_cap(n_cap, nwaves);
    Lds V;
    unsigned char* p = base;
    V.ncut = reinterpret_cast<double*>(p); p += m * 8;
    V.bal = reinterpret_cast<unsigned long long*>(p); p += (size_t)row_cap(n_cap, nwaves) * 8;
    V.px = reinterpret_cast<float*>(p); p += n * 4;
    V.py = reinterpret_cast<float*>(p); p += n * 4;
    V.tmp = reinterpret_cast<unsigned*>(p); p += n * 4;
    V.nlo = reinterpret_cast<float*>(p); p += m * 8;
    V.ndivhigh = reinterpret_cast<float*>(p); p += m * 4;
    V.nlim = reinterpret_cast<unsigned*>(p); p += m * 4;
    V.nbe = reinterpret_cast<unsigned*>(p); p += m * 4;
    V.ncutf = reinterpret_cast<float*>(p); p += m * 4;
    V.ord = reinterpret_cast<unsigned short*>(p); p += n * 2;
    V.eseg = reinterpret_cast<unsigned short*>(p); p += n * 2;
    V.scr = reinterpret_cast<unsigned short*>(p); p += n * 2;
    V.nchild = reinterpret_cast<unsigned short*>(p); p += m * 2;
    V.npar = reinterpret_cast<unsigned short*>(p); p += m * 2;
    V.ncnt = reinterpret_cast<unsigned short*>(p); p += m * 2;
    V.nmid = reinterpret_cast<unsigned short*>(p); p += m * 2;
    V.nflag = p;
    V.m_cap = (int)m;
    return V;
}

struct LdsAcc {   // the points of one node as std::sort sees them, keyed by coordinate `dim` (pointers by value: a reference to the Lds
                  // block would pin all of its seventeen pointers in scratch memory, reloaded behind every barrier)
    float* px; float* py; unsigned short* ord; int dim;
    __device__ __forceinline__ float key(int i) const { return dim ? py[i] : px[i]; }
    __device__ __forceinline__ float ekey(const Elem& e) const { return dim ? e.y : e.x; }
    __device__ __forceinline__ Elem get(int i) const { return Elem{px[i], py[i], ord[i]}; }
    __device__ __forceinline__ void set(int i, const Elem& e) const { px[i] = e.x; py[i] = e.y; ord[i] = (unsigned short)e.id; }
    __device__ __forceinline__ void swap(int i, int j) const { const Elem a = get(i), b = get(j); set(i, b); set(j, a); }
};

constexpr unsigned short kNoNode = 0xFFFF, kRootPar = 0xFFFE;

__device__ __forceinline__ double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl((int)(b & 0xffffffffll), src), hi = __shfl((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <bool WG> __device__ __forceinline__ void team_sync() {
    if (WG) __syncthreads(); else uh_sel::wave_mem_sync();
}

// One sum over the cnt samples v[b], v[b + step], .. in sample order: sq ? s += x * x (float square) : s += x, double accumulator
// (picoflann.h:362-391).  Full groups of eight run without predicates, the loads one group ahead of the dependent addition chain.
__device__ __forceinline__ double sample_sum(const float* __restrict__ v, int b, int step, int cnt, bool sq) {
    double s = 0;
    const float* p = v + b;
    const int full = cnt >> 3;
    float cur[8], nxt[8];
    if (full) {
#pragma unroll
        for (int u = 0; u < 8; u++) cur[u] = p[u * step];
        for (int g = 1; g < full; g++) {
            p += 8 * step;
#pragma unroll
            for (int u = 0; u < 8; u++) nxt[u] = p[u * step];
#pragma unroll
            for (int u = 0; u < 8; u++) { const float x = cur[u]; s += (double)(sq ? x * x : x); }
#pragma unroll
            for (int u = 0; u < 8; u++) cur[u] = nxt[u];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const float x = cur[u]; s += (double)(sq ? x * x : x); }
        p += 8 * step;
    }
    for (int u = 0; u < (cnt & 7); u++) { const float x = p[u * step]; s += (double)(sq ? x * x : x); }
    return s;
}

// inclusive prefix sum over the 64 lanes in the VALU (DPP row shifts + row broadcasts; __shfl_up is ds_bpermute: an LDS round trip a step)
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return x;
}

// x + (the value `ctrl` rows away within the lane's row of 16, 0 from outside the row): one step of a row-wide prefix sum of doubles in the VALU
template <int CTRL>
__device__ __forceinline__ double row_shr_add_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lane 15 of every row of 16 lanes ends up with the row's sum
__device__ __forceinline__ double row16_total_f64(double v) {
    v = row_shr_add_f64<0x111>(v); v = row_shr_add_f64<0x112>(v); v = row_shr_add_f64<0x114>(v); v = row_shr_add_f64<0x118>(v);
    return v;
}

// exponent range of a float's magnitude for the exact-sum test below (zeros do not count, a non-finite value poisons it)
__device__ __forceinline__ void exp_range(float v, int& emin, int& emax) {
    const unsigned bits = __float_as_uint(v);
    const int ex = (int)((bits >> 23) & 0xffu);
    if (ex == 0xff) { emin = -1000; return; }
    if ((bits & 0x7fffffffu) == 0) return;
    emin = ex < emin ? ex : emin; emax = ex > emax ? ex : emax;
}

// Sweeps levels of the tree below the nodes [lvl_b, lvl_e) (depth `depth`) whose points are [eb, ee), with the team's threads tid of nthr
// (the whole workgroup or one wave); runs at most max_levels levels and leaves the next level's node range and depth behind.  cursor: the
// team's node allocator; status[0]: "a node of this level took the std::sort fallback", status[1]: children that will split again,
// status[2]: "a point equals its node's cut" (all zero on entry).  bal: the team's row ballots (>= ceil((ee - eb) / 64) words).
// Point i = eb + 64 r + lane belongs to row r; the team's wave tw owns rows tw, tw + nw, ..
// FAST (the caller guarantees <= kKC rows per wave — always so for the first levels of <= 4096 points on 512 threads and for subtrees of
// <= 512 points): what a lane knows about its points (node, range) stays in registers from level to level and the rows of a phase are
// loaded side by side; otherwise every phase re-reads it row by row.
constexpr int kKC = 8;
template <bool WG, bool FAST>
__device__ __forceinline__ void sweep_levels(const Lds V, const int tid, const int nthr, const int eb, const int ee, int& lvl_b, int& lvl_e, unsigned* cursor,
                             unsigned* status, unsigned long long* bal, int& depth, const int max_levels, unsigned* s_maxdepth, long long* clk, const bool exact) {
#define UH_KD_STAMP(j) do { if (clk && tid == 0 && lv < 6) clk[lv * 8 + (j)] = wall_clock64(); } while (0)
    const int lane = threadIdx.x & 63;
    const int tw = WG ? (int)(threadIdx.x >> 6) : 0, nw = WG ? (int)(blockDim.x >> 6) : 1;
    const int nrows = (ee - eb + 63) >> 6;          // <= 64
    const int ept = (nrows + nw - 1) / nw;          // rows per wave
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    bool more = lvl_e > lvl_b;
    // FAST: per row u of this wave — the lane's point, its node and the node's range
    int fi[kKC], fb[kKC], fe[kKC];
    unsigned fg[kKC];
    bool fvalid[kKC];
    const int nrw = tw < nrows ? (nrows - tw + nw - 1) / nw : 0;   // rows this wave owns (FAST: <= kKC)
    if constexpr (FAST) {
#pragma unroll
        for (int u = 0; u < kKC; u++) {
            const int r = tw + u * nw;
            fi[u] = eb + r * 64 + lane;
            fvalid[u] = r < nrows && fi[u] < ee;
            fg[u] = fvalid[u] ? V.eseg[fi[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kKC; u++) {
            const unsigned be = V.nbe[fg[u]];
            fb[u] = fvalid[u] ? (int)(be & 0xffffu) : eb;
            fe[u] = fvalid[u] ? (int)(be >> 16) : eb;
        }
    }
    for (int lv = 0; lv < max_levels && more; ++lv) {
        const int nL = lvl_e - lvl_b;
        const unsigned c0 = *cursor;
        UH_KD_STAMP(0);
        // ---- mean / variance over the samples, split dimension, cut (picoflann.h:362-401).
        // exact (all non-zero coordinates finite, normal, within 11 binades — any extractor output is): no partial sum of the <= 199 samples
        // (multiples of 2^(e_min-23) below 2^(e_max+9); float squares: multiples of 2^(2 e_min-23) below 2^(2 e_max+10)) is ever rounded, so the
        // sums are the exact real sums in ANY order: a row of 16 lanes shares a node's samples and adds up through DPP (2.6 -> 1.2 us a level;
        // a whole wave per node with every load in flight at once measured no faster: the phase is the chain index -> loads -> sums -> division).
        // Otherwise lanes 4j .. 4j + 3 own node j's four sums and walk the samples in picoflann's order.
        if (exact) {
            const int sub = lane & 15, grp = tid >> 4, ngrp = nthr >> 4;
            for (int n0 = 0; n0 < nL; n0 += ngrp) {
                const int nd = n0 + grp, g = lvl_b + nd;
                int b = 0, e = 0;
                bool act = nd < nL;
                if (act) { const unsigned be = V.nbe[g]; b = (int)(be & 0xffffu); e = (int)(be >> 16); act = e - b > kLeafMax; }
                const int c = e - b, step = c >= 200 ? c / 100 : 1;
                const int cnt = act ? (c + step - 1) / step : 0;
                double a1 = 0, a2 = 0, c1 = 0, c2 = 0;
                for (int k = sub; k < cnt; k += 16) {
                    const float x = V.px[b + k * step], y = V.py[b + k * step];
                    a1 += (double)x; a2 += (double)(x * x); c1 += (double)y; c2 += (double)(y * y);
                }
                a1 = row16_total_f64(a1); a2 = row16_total_f64(a2); c1 = row16_total_f64(c1); c2 = row16_total_f64(c2);
                if (act && sub == 15) {
                    const double inv = 1. / double(cnt);
                    const double m0 = a1 * inv, m1 = c1 * inv;
                    const double v0 = a2 * inv - m0 * m0, v1 = c2 * inv - m1 * m1;
                    const int dim = v1 > v0 ? 1 : 0;
                    const double cut = dim ? m1 : m0;
                    V.ncut[g] = cut;
                    V.ncutf[g] = (float)cut;
                    V.nflag[g] = (unsigned char)((V.nflag[g] & ~3u) | (unsigned)dim);
                }
            }
        } else
        for (int q0 = 0; q0 < nL * 4; q0 += nthr) {
            const int q = q0 + tid, nd = q >> 2, ch = q & 3;
            const int g = lvl_b + nd;
            int b = 0, e = 0;
            bool act = nd < nL;
            if (act) { const unsigned be = V.nbe[g]; b = (int)(be & 0xffffu); e = (int)(be >> 16); act = e - b > kLeafMax; }
            double s = 0;
            int cnt = 0;
            if (act) {
                const int c = e - b, step = c >= 200 ? c / 100 : 1;
                cnt = (c + step - 1) / step;
                s = sample_sum((ch & 2) ? V.py : V.px, b, step, cnt, (ch & 1) != 0);
            }
            const int l0 = lane & ~3;
            const double s1x = shfl_f64(s, l0), s2x = shfl_f64(s, l0 + 1), s1y = shfl_f64(s, l0 + 2), s2y = shfl_f64(s, l0 + 3);
            if (act && ch == 0) {
                const double inv = 1. / double(cnt);
                const double m0 = s1x * inv, m1 = s1y * inv;
                const double v0 = s2x * inv - m0 * m0, v1 = s2y * inv - m1 * m1;
                const int dim = v1 > v0 ? 1 : 0;
                const double cut = dim ? m1 : m0;
                V.ncut[g] = cut;
                V.ncutf[g] = (float)cut;
                V.nflag[g] = (unsigned char)((V.nflag[g] & ~3u) | (unsigned)dim);
            }
        }
        // ---- the two Hoare passes: "v < cut" over the node, then "v <= cut" (over the whole node = over its part behind the first limit:
        // everything in front of it satisfies the predicate and stays where it is; without a point equal to a cut it moves nothing)
        for (int pass = 0; pass < 2; ++pass) {
            team_sync<WG>();
            UH_KD_STAMP(1 + pass);
            if (pass == 0 && tid == 0) { status[0] = 0; status[1] = 0; }   // (the previous level's readers are a barrier behind)
            if constexpr (FAST) {
                unsigned* const nrec = reinterpret_cast<unsigned*>(V.ndivhigh);   // (free until the climb) per node: predicates set in front of it | its middle << 16
                // ---- A: predicates, one ballot per row
                unsigned long long bm[kKC];
                {
                    float fx[kKC], fy[kKC], fcf[kKC];
                    unsigned ffl[kKC];
#pragma unroll
                    for (int u = 0; u < kKC; u++) {
                        if (u >= nrw) break;
                        fx[u] = fvalid[u] ? V.px[fi[u]] : 0.f; fy[u] = fvalid[u] ? V.py[fi[u]] : 0.f;
                        fcf[u] = V.ncutf[fg[u]]; ffl[u] = V.nflag[fg[u]];
                    }
                    bool eqany = false;
#pragma unroll
                    for (int u = 0; u < kKC; u++) {
                        if (u >= nrw) break;
                        const bool active = fvalid[u] && fe[u] - fb[u] > kLeafMax;
                        const float v = (ffl[u] & 1) ? fy[u] : fx[u];
                        bm[u] = __ballot(active && (pass ? v <= fcf[u] : v < fcf[u]));
                        if (pass == 0) eqany = eqany || __ballot(active && v == fcf[u]) != 0ull;
                        if (lane == 0) bal[tw + u * nw] = bm[u];
                    }
                    if (pass == 0 && eqany && lane == 0) atomicOr(&status[2], 1u);
                }
                team_sync<WG>();
                // ---- B1: row prefix; one lane per node: predicates in front of the node and inside it
                const unsigned long long rowbits = lane < nrows ? bal[lane] : 0ull;
                int rb = __popcll(rowbits);
                rb = wave_incl_scan(rb) - rb;
                const int total = __builtin_amdgcn_readlane(rb, 63) + __popc((unsigned)__builtin_amdgcn_readlane((int)(rowbits >> 32), 63)) +
                                  __popc((unsigned)__builtin_amdgcn_readlane((int)(rowbits & 0xffffffffull), 63));
                const bool need2f = status[2] != 0;
                for (int q0 = 0; q0 < nL; q0 += nthr) {   // (uniform trip count: the shuffles below need every lane)
                    const int nd = q0 + tid;
                    const int g = lvl_b + (nd < nL ? nd : 0);
                    const unsigned be = V.nbe[g];
                    const int b = (int)(be & 0xffffu), e = (int)(be >> 16);
                    int Sb, Se;
                    { const int o_ = b - eb, r_ = o_ >> 6; const int base_ = __shfl(rb, r_ & 63); const unsigned long long bv_ = bal[r_ < nrows ? r_ : 0];
                      Sb = r_ < nrows ? base_ + __popcll(bv_ & ((1ull << (o_ & 63)) - 1ull)) : total; }
                    { const int o_ = e - eb, r_ = o_ >> 6; const int base_ = __shfl(rb, r_ & 63); const unsigned long long bv_ = bal[r_ < nrows ? r_ : 0];
                      Se = r_ < nrows ? base_ + __popcll(bv_ & ((1ull << (o_ & 63)) - 1ull)) : total; }
                    if (nd < nL && e - b > kLeafMax) {
                        const int m = Se - Sb;
                        nrec[g] = (unsigned)Sb | ((unsigned)(b + m) << 16);
                        V.nlim[g] = pass ? ((V.nlim[g] & 0xffffu) | ((unsigned)m << 16)) : ((unsigned)m | ((unsigned)m << 16));
                    }
                }
                team_sync<WG>();
                // ---- B2: a misplaced point of the back part lists itself (k-th from the end), one of the front part remembers its k
                int kf[kKC];
#pragma unroll
                for (int u = 0; u < kKC; u++) {
                    kf[u] = -1;
                    if (u >= nrw) break;
                    const unsigned rec = nrec[fg[u]];
                    const int Sb = (int)(rec & 0xffffu), mid = (int)(rec >> 16);
                    const int i = fi[u], b = fb[u], e = fe[u];
                    const int Si = __builtin_amdgcn_readlane(rb, (tw + u * nw) & 63) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm[u], 0u));
                    const bool f = (bm[u] >> lane) & 1ull;
                    if (fvalid[u] && e - b > kLeafMax) {
                        if (i < mid && !f) kf[u] = (i - b) - (Si - Sb);
                        else if (i >= mid && f) V.scr[e - 1 - (Sb + (mid - b) - Si - 1)] = (unsigned short)i;
                    }
                }
                team_sync<WG>();
                // ---- C: every misplaced point of the front part changes places with its partner
                {
                    int ip[kKC];
#pragma unroll
                    for (int u = 0; u < kKC; u++) { if (u >= nrw) break; ip[u] = kf[u] >= 0 ? (int)V.scr[fe[u] - 1 - kf[u]] : fi[u]; }
                    float ax[kKC], ay[kKC], bx[kKC], by[kKC];
                    unsigned short ao[kKC], bo[kKC];
#pragma unroll
                    for (int u = 0; u < kKC; u++) {
                        if (u >= nrw) break;
                        const int i = fvalid[u] ? fi[u] : eb, j = fvalid[u] ? ip[u] : eb;
                        ax[u] = V.px[i]; ay[u] = V.py[i]; ao[u] = V.ord[i];
                        bx[u] = V.px[j]; by[u] = V.py[j]; bo[u] = V.ord[j];
                    }
#pragma unroll
                    for (int u = 0; u < kKC; u++) {
                        if (u >= nrw) break;
                        if (kf[u] >= 0) {
                            V.px[fi[u]] = bx[u]; V.py[fi[u]] = by[u]; V.ord[fi[u]] = bo[u];
                            V.px[ip[u]] = ax[u]; V.py[ip[u]] = ay[u]; V.ord[ip[u]] = ao[u];
                        }
                    }
                }
                if (!need2f) break;
                continue;
            }
            bool eqany = false;
            for (int k = 0; k < ept; k++) {
                const int r = tw + k * nw;
                if (r >= nrows) break;
                const int i = eb + r * 64 + lane;
                const bool valid = i < ee;
                const unsigned g = valid ? V.eseg[i] : 0u;
                const float x = valid ? V.px[i] : 0.f, y = valid ? V.py[i] : 0.f;
                const unsigned be = V.nbe[g];
                const float cf = V.ncutf[g];
                const unsigned fl = V.nflag[g];
                const bool active = valid && (int)(be >> 16) - (int)(be & 0xffffu) > kLeafMax;
                const float v = (fl & 1) ? y : x;
                const bool f = active && (pass ? v <= cf : v < cf);
                const unsigned long long bm = __ballot(f);
                if (lane == 0) bal[r] = bm;
                if (pass == 0) eqany = eqany || __ballot(active && v == cf) != 0ull;
            }
            if (pass == 0 && eqany && lane == 0) atomicOr(&status[2], 1u);
            team_sync<WG>();
            // row prefix: lane r of every wave holds the number of set predicates in front of row r
            const unsigned long long rowbits = lane < nrows ? bal[lane] : 0ull;
            int rb = __popcll(rowbits);
            rb = wave_incl_scan(rb) - rb;
            const int total = __builtin_amdgcn_readlane(rb, 63) + __popc((unsigned)__builtin_amdgcn_readlane((int)(rowbits >> 32), 63)) +
                              __popc((unsigned)__builtin_amdgcn_readlane((int)(rowbits & 0xffffffffull), 63));
            const bool need2 = status[2] != 0;
            // S(x): predicates set in [eb, x)
#define UH_KD_S(x, out) do { const int o_ = (x) - eb, r_ = o_ >> 6; const int base_ = __shfl(rb, r_ & 63); \
                             const unsigned long long bv_ = bal[r_ < nrows ? r_ : 0]; \
                             (out) = r_ < nrows ? base_ + __popcll(bv_ & ((1ull << (o_ & 63)) - 1ull)) : total; } while (0)
            for (int k = 0; k < ept; k++) {
                const int r = tw + k * nw;
                if (r >= nrows) break;
                const int i = eb + r * 64 + lane;
                const bool valid = i < ee;
                const unsigned g = valid ? V.eseg[i] : 0u;
                const unsigned be = V.nbe[g];
                const int b = valid ? (int)(be & 0xffffu) : eb, e = valid ? (int)(be >> 16) : eb;
                const bool active = valid && e - b > kLeafMax;
                const unsigned long long own = bal[r];
                const bool f = (own >> lane) & 1ull;
                const int Si = __builtin_amdgcn_readlane(rb, r) + __popcll(own & ltmask);
                int Sb, Se;
                UH_KD_S(b, Sb);
                UH_KD_S(e, Se);
                if (active) {
                    const int m = Se - Sb, mid = b + m;
                    if (i == b) V.nlim[g] = pass ? ((V.nlim[g] & 0xffffu) | ((unsigned)m << 16)) : ((unsigned)m | ((unsigned)m << 16));
                    if (i < mid && !f) V.scr[b + (i - b) - (Si - Sb)] = (unsigned short)i;     // k-th misplaced point of the front part
                    else if (i >= mid && f) V.scr[e - 1 - (Se - Si - 1)] = (unsigned short)i;   // k-th misplaced point counted from the end
                }
            }
            team_sync<WG>();
            for (int k = 0; k < ept; k++) {
                const int r = tw + k * nw;
                if (r >= nrows) break;
                const int j = eb + r * 64 + lane;
                const bool valid = j < ee;
                const unsigned g = valid ? V.eseg[j] : 0u;
                const unsigned be = V.nbe[g];
                const int b = valid ? (int)(be & 0xffffu) : eb, e = valid ? (int)(be >> 16) : eb;
                const bool active = valid && e - b > kLeafMax;
                int Sb, Se, Sm;
                UH_KD_S(b, Sb);
                UH_KD_S(e, Se);
                const int mid = b + (Se - Sb);
                UH_KD_S(mid, Sm);
                const int nl = (mid - b) - (Sm - Sb);
                if (active && j - b < nl) {
                    const int iL = V.scr[j], iR = V.scr[e - 1 - (j - b)];
                    const float ax = V.px[iL], ay = V.py[iL], bx = V.px[iR], by = V.py[iR];
                    const unsigned short ao = V.ord[iL], bo = V.ord[iR];
                    V.px[iL] = bx; V.py[iL] = by; V.ord[iL] = bo;
                    V.px[iR] = ax; V.py[iR] = ay; V.ord[iR] = ao;
                }
            }
#undef UH_KD_S
            if (!need2) break;
        }
        UH_KD_STAMP(3);
        if (tid == 0) status[2] = 0;   // (read a barrier ago, set again two barriers ahead)
        // ---- where to split (picoflann.h:426-437) and the two children — straight behind the swaps: the limits were counted a barrier ago,
        // and only the std::sort fallback (below, behind the barrier) needs the points in their final places
        for (int q0 = 0; q0 < nL; q0 += nthr) {
            const int nd = q0 + tid;
            if (nd >= nL) continue;
            const int g = lvl_b + nd;
            const unsigned be = V.nbe[g];
            const int b = (int)(be & 0xffffu), e = (int)(be >> 16), c = e - b;
            if (c <= kLeafMax) continue;
            const unsigned lim = V.nlim[g];
            const int lim1 = (int)(lim & 0xffffu), lim2 = (int)(lim >> 16);
            int at = c / 2;
            if (lim1 > c / 2) at = lim1;
            else if (lim2 < c / 2) at = lim2;
            if (lim1 == c || lim2 == 0) at = c / 2;
            unsigned flag = V.nflag[g];
            if (at < kLeafMax || c - at < kLeafMax) {
                at = c / 2;
                flag |= 2;
                atomicOr(&status[0], 1u);
            }
            V.nflag[g] = (unsigned char)flag;
            const unsigned chd = atomicAdd(cursor, 2u);
            V.nchild[g] = (unsigned short)chd;
            V.nmid[g] = (unsigned short)(b + at);
            const unsigned dimbit = 4u << (flag & 1);
            V.nbe[chd] = (unsigned)b | ((unsigned)(b + at) << 16); V.nchild[chd] = 0; V.npar[chd] = (unsigned short)g;
            V.nflag[chd] = (unsigned char)((flag & 0xCu) | dimbit);          // lbox.hi[dim] = cut hides the left subtree's upper bound in dim
            V.nbe[chd + 1] = (unsigned)(b + at) | ((unsigned)e << 16); V.nchild[chd + 1] = 0; V.npar[chd + 1] = (unsigned short)g;
            V.nflag[chd + 1] = (unsigned char)(flag & 0xCu);
            const unsigned nsplit = (at > kLeafMax ? 1u : 0u) + (c - at > kLeafMax ? 1u : 0u);
            if (nsplit) atomicAdd(&status[1], nsplit);
            atomicMax(s_maxdepth, (unsigned)(depth + 1));
        }
        team_sync<WG>();
        UH_KD_STAMP(4);
        const bool anyfb = status[0] != 0;
        more = status[1] != 0;
        const int cur_b = lvl_b;   // (this level's nodes: the fallback below still works on them)
        lvl_b = (int)c0;
        lvl_e = (int)*cursor;
        ++depth;
        // ---- picoflann's std::sort fallback: libstdc++'s partitioning phase by one lane per node, then the insertion sort = a stable sort
        // of what that left: rank by (key, position), permute through tmp
        if (anyfb) {
            for (int q0 = 0; q0 < nL; q0 += nthr) {
                const int nd = q0 + tid;
                if (nd >= nL) continue;
                const int g = cur_b + nd;
                const unsigned fl = V.nflag[g];
                if (!(fl & 2)) continue;
                const unsigned be = V.nbe[g];
                LdsAcc acc{V.px, V.py, V.ord, (int)(fl & 1)};
                sort_phase(acc, (int)(be & 0xffffu), (int)(be >> 16));
            }
            team_sync<WG>();
            unsigned long long pm = 0;
            for (int k = 0; k < ept; k++) {
                const int r = tw + k * nw;
                if (r >= nrows) break;
                const int i = eb + r * 64 + lane;
                if (i >= ee) continue;
                const unsigned g = V.eseg[i];
                const unsigned fl = V.nflag[g];
                if (!(fl & 2) || V.nchild[g] == 0) continue;
                pm |= 1ull << k;
                const unsigned be = V.nbe[g];
                const int b = (int)(be & 0xffffu), e = (int)(be >> 16);
                const float* v = (fl & 1) ? V.py : V.px;
                const float ki = v[i];
                int rk = 0;
                for (int j = b; j < e; j++) { const float kj = v[j]; rk += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
                V.scr[i] = (unsigned short)(b + rk);
            }
            float* tf = reinterpret_cast<float*>(V.tmp);
#define UH_KD_EACH(body) for (int k = 0; k < ept; k++) if ((pm >> k) & 1ull) { const int i = eb + (tw + k * nw) * 64 + lane; body; }
            team_sync<WG>();
            UH_KD_EACH(tf[V.scr[i]] = V.px[i]);
            team_sync<WG>();
            UH_KD_EACH(V.px[i] = tf[i]);
            team_sync<WG>();
            UH_KD_EACH(tf[V.scr[i]] = V.py[i]);
            team_sync<WG>();
            UH_KD_EACH(V.py[i] = tf[i]);
            team_sync<WG>();
            UH_KD_EACH(V.tmp[V.scr[i]] = V.ord[i]);
            team_sync<WG>();
            UH_KD_EACH(V.ord[i] = (unsigned short)V.tmp[i]);
            team_sync<WG>();
#undef UH_KD_EACH
        }
        UH_KD_STAMP(5);
        // ---- every point learns its child; a fallback node's cut is the first point of its right half (picoflann.h:441-446)
        if constexpr (FAST) {
            unsigned chd[kKC], flg[kKC];
            int mid[kKC];
#pragma unroll
            for (int u = 0; u < kKC; u++) { chd[u] = V.nchild[fg[u]]; mid[u] = V.nmid[fg[u]]; flg[u] = V.nflag[fg[u]]; }
#pragma unroll
            for (int u = 0; u < kKC; u++) {
                if (!fvalid[u] || chd[u] == 0) continue;    // a leaf's points keep the leaf
                const int i = fi[u];
                if ((flg[u] & 2) && i == mid[u]) { const double cut = (double)((flg[u] & 1) ? V.py[i] : V.px[i]); V.ncut[fg[u]] = cut; V.ncutf[fg[u]] = (float)cut; }
                if (i < mid[u]) { fg[u] = chd[u]; fe[u] = mid[u]; } else { fg[u] = chd[u] + 1; fb[u] = mid[u]; }
                V.eseg[i] = (unsigned short)fg[u];
            }
        } else
        for (int k = 0; k < ept; k++) {
            const int r = tw + k * nw;
            if (r >= nrows) break;
            const int i = eb + r * 64 + lane;
            if (i >= ee) continue;
            const unsigned g = V.eseg[i];
            const unsigned chd = V.nchild[g];
            if (chd == 0) continue;    // a leaf's points keep the leaf
            const int mid = V.nmid[g];
            const unsigned flag = V.nflag[g];
            if ((flag & 2) && i == mid) { const double cut = (double)((flag & 1) ? V.py[i] : V.px[i]); V.ncut[g] = cut; V.ncutf[g] = (float)cut; }
            V.eseg[i] = (unsigned short)(i < mid ? chd : chd + 1);
        }
        UH_KD_STAMP(6);
    }
#undef UH_KD_STAMP
}

struct Node24 { float divlow, divhigh; int left, right; int leaf_begin; short leaf_count, col; };
static_assert(sizeof(Node24) == 24, "node layout");
struct Meta { unsigned long long word; int n, n_nodes, max_depth, m_used; double box[4]; };   // 56 bytes, in pinned host memory

// ---- the build over THREE launches (round 6, second form): the top levels by one workgroup (kd_top), one workgroup per subtree below them
// on its own compute unit (kd_sub: one wave per SIMD instead of two sharing one, a quarter of the rows per wave), a small join launch that
// numbers the nodes across subtrees.  What travels between them (HBM scratch of the frame object):
struct TopNode { unsigned nbe; unsigned short nchild, npar; unsigned flag; double cut; };
struct TopDump { int nsub, n, depth, ntop, lvl_b, lvl_e, exact, pad1; TopNode node[64]; };   // nsub = 0: kd_top built the whole tree itself (small / shallow clouds)
struct SubSum { int cnt_int, n_nodes, maxdepth, pad; float lo[2]; double rhi[2]; };       // a subtree as its parent sees it
constexpr int kSplitMin = 1200;   // below this one workgroup finishes sooner than three launches do (measured: 500 points 55 vs 64 us, 2000 points 102 vs 95 us)
constexpr int kSubMax = 8;      // subtrees = nodes of the level below log2(waves of kd_top) sweeps

enum : int { kFull = 0, kTop = 1, kSub = 2 };
struct SubArgs {            // kSub: the subtree this workgroup builds
    const float4* pts;      // the points as kd_top left them: {x, y, bits(keypoint), -} in position order
    int pos0;               // position of the subtree's first point
    unsigned root_flags;    // the "upper bound overridden" bits its root inherits
    int depth0;             // depth of its root
    int exact;              // the cloud passed the exact-sum test (sweep_levels)
    SubSum* sum;
};

// One workgroup (blockDim.x = 256 or 512 threads).  kFull: the whole build.  in[i] = {x, y, bits(octave), -} of keypoint i; nodes_out: room
// for 2 * (n / 5) + 1 nodes (uh_kd::node_cap covers it), leaf_out[i] = {x, y, bits(keypoint << 4 | octave), 0} in leaf order; thread 0 leaves
// {n, n_nodes, depth, root box} in *meta and, last, stores `word` into meta->word with system-scope release.  kTop: the same, unless the
// tree is deep enough to be shared out after the workgroup-wide levels — then the points, the top nodes and the subtree roots go to
// `dump` / `pts_out` and nothing else is written.  kSub: the subtree of SubArgs with picoflann's numbering LOCAL to it (root 0), leaf
// records at their final places, a SubSum for the join.
template <int MODE>
__device__ void build_workgroup(unsigned char* lds_base, const int n_cap, const float4* __restrict__ in, const int n, Node24* __restrict__ nodes_out,
                                float4* __restrict__ leaf_out, Meta* meta, const unsigned long long word, long long* clk, TopDump* dump, float4* pts_out,
                                const SubArgs sub) {
#define UH_KD_TOP(j) do { if (clk && threadIdx.x == 0) clk[j] = wall_clock64(); } while (0)
    UH_KD_TOP(0);
    __shared__ unsigned s_cursor[17], s_status[17][4], s_maxdepth;
    __shared__ double s_rhi[16][2];
    __shared__ int s_emin, s_emax;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const Lds V = carve(lds_base, n_cap, nwaves);
    const int pos0 = MODE == kSub ? sub.pos0 : 0;
    if (MODE != kSub && tid == 0) { s_emin = 1 << 20; s_emax = 0; }
    for (int g = tid; g < V.m_cap; g += nthr) V.npar[g] = kNoNode;
    int emin = 1 << 20, emax = 0;
    if (MODE != kSub) __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        if constexpr (MODE == kSub) {
            const float4 r = sub.pts[pos0 + i];
            V.px[i] = r.x; V.py[i] = r.y; V.ord[i] = (unsigned short)__float_as_uint(r.z);
        } else {
            const float4 r = in[i];
            V.px[i] = r.x; V.py[i] = r.y; V.ord[i] = (unsigned short)i;
            exp_range(r.x, emin, emax); exp_range(r.y, emin, emax);
        }
        V.eseg[i] = 0;
    }
    if constexpr (MODE != kSub) {   // the exponent range of the cloud, for sweep_levels' exact sums
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int a = __shfl_xor(emin, o), b = __shfl_xor(emax, o);
            emin = a < emin ? a : emin; emax = b > emax ? b : emax;
        }
        if (lane == 0) { atomicMin(&s_emin, emin); atomicMax(&s_emax, emax); }
    }
    if (tid < 17) { s_status[tid][0] = 0; s_status[tid][1] = 0; s_status[tid][2] = 0; }
    __syncthreads();
    const int depth_root = MODE == kSub ? sub.depth0 : 1;
    if (tid == 0) {
        V.nbe[0] = (unsigned)n << 16; V.nchild[0] = 0; V.npar[0] = kRootPar; V.nflag[0] = MODE == kSub ? (unsigned char)(sub.root_flags & 0xCu) : (unsigned char)0;
        s_cursor[16] = 1;
        s_maxdepth = n > 0 ? (unsigned)depth_root : 0u;
    }
    __syncthreads();
    UH_KD_TOP(1);
    const bool exact = MODE == kSub ? sub.exact != 0 : (s_emin >= 1 && s_emax - s_emin <= 10);   // (emin < 1: a denormal or a non-finite coordinate)
    int m_used = n > 0 ? 1 : 0;
    if (n > kLeafMax) {
        int lvl_b = 0, lvl_e = 1, depth = depth_root;
        const int k_wg = uh_sel::floor_log2(nwaves);
        if ((n + 63) / 64 <= kKC * nwaves) sweep_levels<true, true>(V, tid, nthr, 0, n, lvl_b, lvl_e, &s_cursor[16], s_status[16], V.bal, depth, k_wg, &s_maxdepth, clk ? clk + 16 : nullptr, exact);
        else sweep_levels<true, false>(V, tid, nthr, 0, n, lvl_b, lvl_e, &s_cursor[16], s_status[16], V.bal, depth, k_wg, &s_maxdepth, clk ? clk + 16 : nullptr, exact);
        __syncthreads();
        UH_KD_TOP(2);
        const int nL = lvl_e - lvl_b;
        if constexpr (MODE == kTop) {
            // deep enough to share out?  (every node of the level exists and at least one will split again; a shallow tree is finished here)
            if (n >= kSplitMin && nL == nwaves && nL <= kSubMax && s_status[16][1] != 0) {
                for (int i = tid; i < n; i += nthr) pts_out[i] = make_float4(V.px[i], V.py[i], __uint_as_float((unsigned)V.ord[i]), 0.f);
                const int ntop = (int)s_cursor[16];
                for (int g = tid; g < ntop && g < 64; g += nthr) {
                    TopNode t;
                    t.nbe = V.nbe[g]; t.nchild = V.nchild[g]; t.npar = V.npar[g]; t.flag = V.nflag[g]; t.cut = V.ncut[g];
                    dump->node[g] = t;
                }
                if (tid == 0) { dump->exact = exact ? 1 : 0; dump->nsub = nL; dump->n = n; dump->depth = depth; dump->ntop = ntop; dump->lvl_b = lvl_b; dump->lvl_e = lvl_e; }
                return;
            }
        }
        // hand-over: node lvl_b + w goes to wave w with a node region of its own (a subtree of c points holds at most 2c/5 nodes)
        unsigned base = s_cursor[16], mine = 0;
        int mb = 0, me = 0, before = 0;   // before: subtrees in front of mine (the level's nodes are in allocation order, not in point order)
        {
            const unsigned bw = V.nbe[lvl_b + (wave < nL ? wave : 0)];
            mb = (int)(bw & 0xffffu); me = (int)(bw >> 16);
        }
        for (int j = 0; j < nL; j++) {
            const unsigned be = V.nbe[lvl_b + j];
            const int b = (int)(be & 0xffffu), e = (int)(be >> 16);
            if (j == wave) mine = base;
            before += b < mb ? 1 : 0;
            base += (unsigned)(2 * (e - b) / 5 + 2);
        }
        m_used = (int)base;
        if (wave < nL && me - mb > kLeafMax) {
            if (lane == 0) s_cursor[wave] = mine;
            uh_sel::wave_mem_sync();
            int lb = lvl_b + wave, le = lb + 1, d = depth;
            if (me - mb <= 64 * kKC) sweep_levels<false, true>(V, lane, 64, mb, me, lb, le, &s_cursor[wave], s_status[wave], V.bal + (mb >> 6) + before, d, 1 << 20, &s_maxdepth,
                                                               clk && wave == 0 ? clk + 64 : nullptr, exact);
            else sweep_levels<false, false>(V, lane, 64, mb, me, lb, le, &s_cursor[wave], s_status[wave], V.bal + (mb >> 6) + before, d, 1 << 20, &s_maxdepth,
                                            clk && wave == 0 ? clk + 64 : nullptr, exact);
        }
        UH_KD_TOP(3);
        if (clk && lane == 0) { clk[128 + wave] = wall_clock64(); clk[144 + wave] = me - mb; }
        __syncthreads();
        UH_KD_TOP(4);
    }
    if constexpr (MODE == kTop) { if (tid == 0) dump->nsub = 0; }
    // ---- from the leaves up: tight lower bounds (divhigh of a parent = its right child's in the split dimension), counts of inner nodes;
    // the root's upper bounds from the leaves / cuts no ancestor overrides.  The second child to arrive at a parent goes on.
    for (int g = tid; g < m_used; g += nthr) V.nlim[g] = 0;
    __syncthreads();
    UH_KD_TOP(5);
    double rhi0 = -__builtin_huge_val(), rhi1 = -__builtin_huge_val();
    for (int g = tid; g < m_used; g += nthr) {
        if (V.npar[g] == kNoNode || V.nchild[g] != 0) continue;
        const unsigned be = V.nbe[g];
        const int b = (int)(be & 0xffffu), e = (int)(be >> 16);
        float lx = V.px[b], hx = lx, ly = V.py[b], hy = ly;
        for (int i = b + 1; i < e; i++) {
            const float x = V.px[i], y = V.py[i];
            lx = x < lx ? x : lx; hx = x > hx ? x : hx;
            ly = y < ly ? y : ly; hy = y > hy ? y : hy;
        }
        V.nlo[2 * g] = lx; V.nlo[2 * g + 1] = ly; V.ncnt[g] = 0;
        const unsigned fl = V.nflag[g];
        if (!(fl & 4) && (double)hx > rhi0) rhi0 = (double)hx;
        if (!(fl & 8) && (double)hy > rhi1) rhi1 = (double)hy;
        int X = g;
        for (;;) {
            const unsigned p = V.npar[X];
            if (p == kRootPar) break;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const unsigned old = atomicAdd(&V.nlim[p], 1u);
            if (old == 0) break;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const unsigned l = V.nchild[p], r = l + 1;
            const unsigned pf = V.nflag[p];
            const int dim = (int)(pf & 1);
            const float l0 = V.nlo[2 * l], l1 = V.nlo[2 * l + 1], r0 = V.nlo[2 * r], r1 = V.nlo[2 * r + 1];
            V.ndivhigh[p] = dim ? r1 : r0;
            V.nlo[2 * p] = l0 < r0 ? l0 : r0;
            V.nlo[2 * p + 1] = l1 < r1 ? l1 : r1;
            V.ncnt[p] = (unsigned short)(1 + V.ncnt[l] + V.ncnt[r]);
            const double cut = V.ncut[p];
            if (!(pf & (4u << dim))) { if (dim) { if (cut > rhi1) rhi1 = cut; } else if (cut > rhi0) rhi0 = cut; }
            X = (int)p;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = shfl_f64(rhi0, lane ^ o), b = shfl_f64(rhi1, lane ^ o);
        rhi0 = a > rhi0 ? a : rhi0; rhi1 = b > rhi1 ? b : rhi1;
    }
    if (lane == 0) { s_rhi[wave][0] = rhi0; s_rhi[wave][1] = rhi1; }
    __syncthreads();
    UH_KD_TOP(6);
    // ---- picoflann's node numbers (children of the r-th split in depth-first order: 2r + 1, 2r + 2) and the flattened records
    const int n_int = n > 0 && V.nchild[0] != 0 ? (int)V.ncnt[0] : 0;
    const int n_nodes = n > 0 ? 1 + 2 * n_int : 0;
    for (int g = tid; g < m_used; g += nthr) {
        if (V.npar[g] == kNoNode) continue;
        int acc = 0, stepg = 0, right = 0;
        for (int X = g;;) {
            const unsigned p = V.npar[X];
            if (p == kRootPar) break;
            const unsigned l = V.nchild[p];
            const int isr = l + 1 == (unsigned)X;
            const int s = 1 + (isr ? (int)V.ncnt[l] : 0);
            if (X == g) { stepg = s; right = isr; } else acc += s;
            X = (int)p;
        }
        const bool root = V.npar[g] == kRootPar;
        const int id = root ? 0 : 2 * acc + 1 + right;
        const int pre = root ? 0 : acc + stepg;
        const unsigned be = V.nbe[g];
        Node24 nd;
        if (V.nchild[g] != 0) {
            nd.divlow = (float)V.ncut[g]; nd.divhigh = V.ndivhigh[g];
            nd.left = 2 * pre + 1; nd.right = 2 * pre + 2; nd.leaf_begin = 0; nd.leaf_count = 0; nd.col = (short)(V.nflag[g] & 1);
        } else {
            nd.divlow = 0.f; nd.divhigh = 0.f; nd.left = -1; nd.right = -1;
            nd.leaf_begin = pos0 + (int)(be & 0xffffu); nd.leaf_count = (short)((be >> 16) - (be & 0xffffu)); nd.col = 0;
        }
        nodes_out[id] = nd;
    }
    UH_KD_TOP(7);
    for (int i = tid; i < n; i += nthr) {
        const unsigned id = V.ord[i];
        const unsigned oct = __float_as_uint(in[id].z);
        leaf_out[pos0 + i] = make_float4(V.px[i], V.py[i], __uint_as_float((id << 4) | (oct & 15u)), 0.f);
    }
    if (tid == 0) {
        double h0 = s_rhi[0][0], h1 = s_rhi[0][1];
        for (int w = 1; w < nwaves; w++) { h0 = s_rhi[w][0] > h0 ? s_rhi[w][0] : h0; h1 = s_rhi[w][1] > h1 ? s_rhi[w][1] : h1; }
        if constexpr (MODE == kSub) {
            SubSum sm;
            sm.cnt_int = n_int; sm.n_nodes = n_nodes; sm.maxdepth = (int)s_maxdepth; sm.pad = 0;
            sm.lo[0] = n > 0 ? V.nlo[0] : 0.f; sm.lo[1] = n > 0 ? V.nlo[1] : 0.f; sm.rhi[0] = h0; sm.rhi[1] = h1;
            *sub.sum = sm;
            UH_KD_TOP(8);
        } else {
            meta->n = n; meta->n_nodes = n_nodes; meta->max_depth = (int)s_maxdepth; meta->m_used = m_used;
            if (n > 0) { meta->box[0] = (double)V.nlo[0]; meta->box[1] = h0; meta->box[2] = (double)V.nlo[1]; meta->box[3] = h1; }
            else { meta->box[0] = meta->box[1] = meta->box[2] = meta->box[3] = 0.0; }
            UH_KD_TOP(8);
            __hip_atomic_store(&meta->word, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#undef UH_KD_TOP
}

// The join launch (one workgroup): the top nodes' counts, lower bounds and numbers from the subtrees' summaries (thread 0: <= 15 nodes), then every
// subtree's records to their final numbers (a local number L > 0 becomes L + 2 * the subtree root's depth-first rank, the local root takes the
// number its parent gives it), meta and the completion word.
__device__ void join_workgroup(const TopDump* __restrict__ dump, const SubSum* __restrict__ sums, const Node24* __restrict__ sub_nodes, const int node_stride,
                               Node24* __restrict__ nodes_out, Meta* meta, const unsigned long long word) {
    __shared__ int s_cnt[64], s_pre[64], s_id[64], s_base[kSubMax], s_rootid[kSubMax], s_nn[kSubMax];
    __shared__ float s_lo[64][2];
    __shared__ TopNode s_top[2 * kSubMax];
    __shared__ SubSum s_sum[kSubMax];
    const int nsub = dump->nsub;
    if (nsub == 0) return;   // kd_top finished the tree (and posted the word) itself
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int ntop = dump->ntop, lvl_b = dump->lvl_b, lvl_e = dump->lvl_e;
    // (one parallel read of what thread 0 walks through below: a dependent global load per node costs it a microsecond each)
    if (tid < ntop) s_top[tid] = dump->node[tid];
    if (tid >= 64 && tid < 64 + nsub) s_sum[tid - 64] = sums[tid - 64];
    __syncthreads();
    if (tid == 0) {
        double rhi0 = -__builtin_huge_val(), rhi1 = -__builtin_huge_val();
        int maxd = dump->depth;
        for (int g = ntop - 1; g >= 0; g--) {   // children carry larger numbers than their parents
            const TopNode t = s_top[g];
            if (g >= lvl_b && g < lvl_e) {
                const SubSum sm = s_sum[g - lvl_b];
                s_cnt[g] = sm.cnt_int; s_lo[g][0] = sm.lo[0]; s_lo[g][1] = sm.lo[1];
                rhi0 = sm.rhi[0] > rhi0 ? sm.rhi[0] : rhi0; rhi1 = sm.rhi[1] > rhi1 ? sm.rhi[1] : rhi1;
                maxd = sm.maxdepth > maxd ? sm.maxdepth : maxd;
            } else {
                const int l = t.nchild, r = l + 1, dim = (int)(t.flag & 1);
                s_cnt[g] = 1 + s_cnt[l] + s_cnt[r];
                s_lo[g][0] = s_lo[l][0] < s_lo[r][0] ? s_lo[l][0] : s_lo[r][0];
                s_lo[g][1] = s_lo[l][1] < s_lo[r][1] ? s_lo[l][1] : s_lo[r][1];
                if (!(t.flag & (4u << dim))) { if (dim) { if (t.cut > rhi1) rhi1 = t.cut; } else if (t.cut > rhi0) rhi0 = t.cut; }
            }
        }
        s_pre[0] = 0; s_id[0] = 0;
        for (int g = 0; g < ntop; g++) {
            if (g >= lvl_b && g < lvl_e) continue;
            const TopNode t = s_top[g];
            const int l = t.nchild, r = l + 1, dim = (int)(t.flag & 1);
            s_pre[l] = s_pre[g] + 1; s_pre[r] = s_pre[g] + 1 + s_cnt[l];
            s_id[l] = 2 * s_pre[g] + 1; s_id[r] = 2 * s_pre[g] + 2;
            Node24 nd;
            nd.divlow = (float)t.cut; nd.divhigh = s_lo[r][dim]; nd.left = s_id[l]; nd.right = s_id[r]; nd.leaf_begin = 0; nd.leaf_count = 0; nd.col = (short)dim;
            nodes_out[s_id[g]] = nd;
        }
        for (int j = 0; j < nsub; j++) {
            const int g = lvl_b + j;
            s_base[j] = s_pre[g]; s_rootid[j] = s_id[g]; s_nn[j] = s_sum[j].n_nodes;
        }
        meta->n = dump->n; meta->n_nodes = 1 + 2 * s_cnt[0]; meta->max_depth = maxd; meta->m_used = ntop;
        meta->box[0] = (double)s_lo[0][0]; meta->box[1] = rhi0; meta->box[2] = (double)s_lo[0][1]; meta->box[3] = rhi1;
    }
    __syncthreads();
    for (int j = 0; j < nsub; j++) {
        const int add = 2 * s_base[j], nn = s_nn[j];
        const Node24* src = sub_nodes + (size_t)j * node_stride;
        for (int L = tid; L < nn; L += nthr) {
            Node24 nd = src[L];
            if (nd.left >= 0) { nd.left += add; nd.right += add; }
            nodes_out[L == 0 ? s_rootid[j] : L + add] = nd;
        }
    }
    __syncthreads();   // (the records are for later launches of this stream; the word announces meta, which thread 0 wrote itself)
    if (tid == 0) __hip_atomic_store(&meta->word, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace uh_kd
#endif
