// Selection with the exact permutation of libstdc++'s std::nth_element (GCC 11 <bits/stl_algo.h>,
// __introselect: median-of-3 to first, unguarded Hoare partition, heap-select when the depth limit
// 2*floor(log2(n)) is exhausted, insertion sort once the window is <= 3 elements).
//
// Why it exists: the reference truncates cv::KeyPointsFilter::retainBest's output right after its
// std::nth_element (src/featureextractors/ORBextractor.cpp:1053-1055, :1071-1072), so WHICH tied
// keypoints survive, and their ORDER, are defined by that algorithm's data movement.  The ORB stage runs
// the same data movement on the GPU so that keypoint indices stay bit-exact without a host round trip.
//
// Elements are opaque 32-bit words; ordering is "greater response first" on key(e) = e >> 24
// (the FAST score), exactly cv::KeypointResponseGreater on integral responses.
// tests/test_introselect.py checks this header (compiled for the host) against std::nth_element.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define UH_HD __host__ __device__ __forceinline__
#else
#define UH_HD inline
#endif

namespace uh_sel {

UH_HD uint32_t key(uint32_t e) { return e >> 24; }
UH_HD bool before(uint32_t a, uint32_t b) { return key(a) > key(b); }   // comp(a,b): a has the larger response

template <typename P> UH_HD void swp(P a, int i, int j) { uint32_t t = a[i]; a[i] = a[j]; a[j] = t; }

// std::__move_median_to_first(result, a, b, c)
template <typename P> UH_HD void median_to_first(P v, int result, int a, int b, int c) {
    if (before(v[a], v[b])) {
        if (before(v[b], v[c])) swp(v, result, b);
        else if (before(v[a], v[c])) swp(v, result, c);
        else swp(v, result, a);
    } else if (before(v[a], v[c])) swp(v, result, a);
    else if (before(v[b], v[c])) swp(v, result, c);
    else swp(v, result, b);
}

// std::__unguarded_partition(first, last, pivot)
template <typename P> UH_HD int unguarded_partition(P v, int first, int last, int pivot) {
    const uint32_t pv = v[pivot];   // the pivot slot lies outside [first,last) and is never written here
    for (;;) {
        while (before(v[first], pv)) ++first;
        --last;
        while (before(pv, v[last])) --last;
        if (!(first < last)) return first;
        swp(v, first, last);
        ++first;
    }
}

// std::__adjust_heap + std::__push_heap on v[first .. first+len), comparator `before`
template <typename P> UH_HD void adjust_heap(P v, int first, int hole, int len, uint32_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (before(v[first + child], v[first + child - 1])) child--;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && before(v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

// std::__heap_select(first, middle, last)
template <typename P> UH_HD void heap_select(P v, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {   // std::__make_heap
        int parent = (len - 2) / 2;
        for (;;) {
            uint32_t value = v[first + parent];
            adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i) {
        if (before(v[i], v[first])) {   // std::__pop_heap(first, middle, i)
            uint32_t value = v[i];
            v[i] = v[first];
            adjust_heap(v, first, 0, len, value);
        }
    }
}

// std::__insertion_sort(first, last)
template <typename P> UH_HD void insertion_sort(P v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        uint32_t val = v[i];
        if (before(val, v[first])) {
            for (int j = i; j > first; --j) v[j] = v[j - 1];
            v[first] = val;
        } else {   // __unguarded_linear_insert
            int j = i;
            while (before(val, v[j - 1])) { v[j] = v[j - 1]; --j; }
            v[j] = val;
        }
    }
}

UH_HD int floor_log2(int n) { int r = 0; while (n > 1) { n >>= 1; ++r; } return r; }

// std::nth_element(v, v+nth, v+n, greater-response)
template <typename P> UH_HD void nth_element_desc(P v, int n, int nth) {
    if (n <= 0 || nth >= n) return;   // first == last || nth == last
    int first = 0, last = n;
    int depth_limit = 2 * floor_log2(n);
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select(v, first, nth + 1, last);
            swp(v, first, nth);
            return;
        }
        --depth_limit;
        int mid = first + (last - first) / 2;
        median_to_first(v, first, first + 1, mid, last - 1);
        int cut = unguarded_partition(v, first + 1, last, first);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    insertion_sort(v, first, last);
}

}  // namespace uh_sel

// Host emulation of the pairing formulation used by the wave-cooperative partition below (test hook).
#if !defined(__HIPCC__)
#include <vector>
namespace uh_sel {
inline int partition_pairing_host(uint32_t* v, int first, int last, int pivot) {
    const uint32_t kp = key(v[pivot]);
    std::vector<int> L, R;
    for (int i = first; i < last; i++) if (!(key(v[i]) > kp)) L.push_back(i);
    for (int i = last - 1; i >= first; i--) if (!(kp > key(v[i]))) R.push_back(i);
    int m = 0;
    while (m < (int)L.size() && m < (int)R.size() && L[m] < R[m]) m++;
    for (int i = 0; i < m; i++) swp(v, L[i], R[i]);
    int cut = 0x7fffffff;
    if (m < (int)L.size()) cut = L[m];
    if (m > 0 && R[m - 1] < cut) cut = R[m - 1];
    return cut;
}
inline void nth_element_desc_pairing_host(uint32_t* v, int n, int nth) {
    if (n <= 0 || nth >= n) return;
    int first = 0, last = n, depth_limit = 2 * floor_log2(n);
    while (last - first > 3) {
        if (depth_limit == 0) { heap_select(v, first, nth + 1, last); swp(v, first, nth); return; }
        --depth_limit;
        const int mid = first + (last - first) / 2;
        median_to_first(v, first, first + 1, mid, last - 1);
        const int cut = partition_pairing_host(v, first + 1, last, first);
        if (cut <= nth) first = cut; else last = cut;
    }
    insertion_sort(v, first, last);
}
}  // namespace uh_sel
#endif

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------------------------
// Wave-cooperative version (64 lanes work on ONE array).  The sequential Hoare partition of libstdc++ is equivalent to:
//   L = positions (ascending) whose element does NOT precede the pivot      (the left scan's stopping points)
//   R = positions (descending) which the pivot does NOT precede             (the right scan's stopping points)
//   swap the pairs (L_i, R_i) while L_i < R_i (m pairs);  cut = min(L_m, R_{m-1})
// (no position can belong to two swapped pairs, so the swaps are independent).  The equivalence, including ties and the
// crossing cases, is checked against std::nth_element by tests/test_introselect.py through the host emulation
// uh_sel::partition_pairing_host.  Median-of-3, the <=3-element insertion sort and the depth-limit heap-select stay
// sequential on lane 0 (a handful of elements).
namespace uh_sel {

__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The same partition for ranges of up to 512 entries (round 6): ONE round of loads for all eight chunks, both predicates from the same
// registers, the list offsets from the ballots' popcounts — instead of two chunk-by-chunk sweeps whose every step waited for its own LDS round
// trip (8 + 8 dependent round trips per pass at 500 entries: 1.4 us; the level-wide retainBest of level 0 takes ~9 passes).  Same lists
// (Lpos ascending, Rpos descending), same swaps, same cut.
template <int NC, typename P, typename S>   // NC chunks of 64 entries cover the range (1, 2, 4 or 8: the late passes of a selection and the per-cell selections are short)
__device__ __forceinline__ int wave_partition_nc(P v, int first, int last, int pivot, S Lpos, S Rpos, int lane) {
    const uint32_t kp = key(v[pivot]);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t e[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { const int pos = first + 64 * c + lane; e[c] = pos < last ? v[pos] : 0u; }
    unsigned long long mL[NC], mR[NC];
    int nL = 0, nR = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const bool in = first + 64 * c + lane < last;
        mL[c] = __ballot(in && !(key(e[c]) > kp));
        mR[c] = __ballot(in && !(kp > key(e[c])));
        nR += __popcll(mR[c]);
    }
    int preR = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int pos = first + 64 * c + lane;
        if ((mL[c] >> lane) & 1ull) Lpos[nL + __popcll(mL[c] & below)] = pos;
        if ((mR[c] >> lane) & 1ull) Rpos[nR - 1 - (preR + __popcll(mR[c] & below))] = pos;   // rank counted from the top of the range
        nL += __popcll(mL[c]);
        preR += __popcll(mR[c]);
    }
    wave_mem_sync();
    constexpr int NP = NC >= 2 ? NC / 2 : 1;   // chunks of pairs (np <= 32 NC)
    const int np = nL < nR ? nL : nR;
    int a4[NP], b4[NP], m = 0;
#pragma unroll
    for (int u = 0; u < NP; u++) {
        const int i = 64 * u + lane;
        a4[u] = i < np ? Lpos[i] : 0; b4[u] = i < np ? Rpos[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < NP; u++) m += __popcll(__ballot(64 * u + lane < np && a4[u] < b4[u]));   // (L ascending / R descending: the pairs that still cross are a prefix)
    uint32_t va[NP], vb[NP];
#pragma unroll
    for (int u = 0; u < NP; u++) { const bool on = 64 * u + lane < m; va[u] = on ? v[a4[u]] : 0u; vb[u] = on ? v[b4[u]] : 0u; }
#pragma unroll
    for (int u = 0; u < NP; u++) if (64 * u + lane < m) { v[a4[u]] = vb[u]; v[b4[u]] = va[u]; }
    int cut = 0x7fffffff;
    if (m < nL) cut = Lpos[m];
    if (m > 0) { const int r = Rpos[m - 1]; cut = r < cut ? r : cut; }
    wave_mem_sync();
    return cut;
}

template <typename P, typename S>
__device__ __forceinline__ int wave_partition(P v, int first, int last, int pivot, S Lpos, S Rpos, int lane) {
    {
        const int len = last - first;
        if (len <= 64) return wave_partition_nc<1>(v, first, last, pivot, Lpos, Rpos, lane);
        if (len <= 128) return wave_partition_nc<2>(v, first, last, pivot, Lpos, Rpos, lane);
        if (len <= 256) return wave_partition_nc<4>(v, first, last, pivot, Lpos, Rpos, lane);
        if (len <= 512) return wave_partition_nc<8>(v, first, last, pivot, Lpos, Rpos, lane);
    }
    const uint32_t kp = key(v[pivot]);
    int nL = 0;
    for (int s = first; s < last; s += 64) {
        const int pos = s + lane;
        const bool in = pos < last;
        const uint32_t e = in ? v[pos] : 0u;
        const bool isL = in && !(key(e) > kp);
        const unsigned long long m = __ballot(isL);
        if (isL) Lpos[nL + __popcll(m & ((1ull << lane) - 1ull))] = pos;
        nL += __popcll(m);
    }
    int nR = 0;
    for (int s = last; s > first; s -= 64) {
        const int pos = s - 64 + lane;
        const bool in = pos >= first;
        const uint32_t e = in ? v[pos] : 0u;
        const bool isR = in && !(kp > key(e));
        const unsigned long long m = __ballot(isR);
        if (isR) Rpos[nR + __popcll((m >> lane) >> 1)] = pos;   // rank counted from the top of the chunk
        nR += __popcll(m);
    }
    wave_mem_sync();
    const int np = nL < nR ? nL : nR;
    int m = 0;
    for (int i0 = 0; i0 < np; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < np && Lpos[i] < Rpos[i];
        const unsigned long long mk = __ballot(ok);
        m += __popcll(mk);
        if (mk != ~0ull) break;     // L ascending / R descending: once a pair fails all later pairs fail
    }
    for (int i = lane; i < m; i += 64) {
        const int a = Lpos[i], b = Rpos[i];
        const uint32_t va = v[a], vb = v[b];
        v[a] = vb;
        v[b] = va;
    }
    int cut = 0x7fffffff;
    if (m < nL) cut = Lpos[m];
    if (m > 0) { const int r = Rpos[m - 1]; cut = r < cut ? r : cut; }
    wave_mem_sync();
    return cut;
}

// std::nth_element(v, v+nth, v+n, greater-response), executed by a whole wave; Lpos/Rpos: scratch of >= n ints each
template <typename P, typename S>
__device__ __forceinline__ void wave_nth_element_desc(P v, int n, int nth, S Lpos, S Rpos, int lane) {
    if (n <= 0 || nth >= n) return;
    int first = 0, last = n;
    int depth_limit = 2 * floor_log2(n);
    while (last - first > 3) {
        if (depth_limit == 0) {
            if (lane == 0) { heap_select(v, first, nth + 1, last); swp(v, first, nth); }
            wave_mem_sync();
            return;
        }
        --depth_limit;
        const int mid = first + (last - first) / 2;
        if (lane == 0) median_to_first(v, first, first + 1, mid, last - 1);
        wave_mem_sync();
        const int cut = wave_partition(v, first + 1, last, first, Lpos, Rpos, lane);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) insertion_sort(v, first, last);
    wave_mem_sync();
}

}  // namespace uh_sel
#endif  // __HIPCC__
