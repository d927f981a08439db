// Projection matcher on MI355X (gfx950) behind the C ABI (uh_projmatch_*): map points of the neighbouring keyframes are
// projected into the current frame and matched to its keypoints inside a scale-dependent disc.
//
// Semantic contract (reference file:line):
//   src/map.cpp:651-770                 Map::matchFrameToMapPoints: view-cosine >= 0.5, depth >= 0, distance-invariance window,
//                                       in-image test, predicted octave, disc radius scale*maxRepjDist (x1.6 if cos < 0.98),
//                                       octaves [p-1, p], best / second-best Hamming bookkeeping IN CANDIDATE ORDER, the 0.8
//                                       ratio rule when both have the same octave, then filter_ambiguous_query
//   src/map_types/frame.cpp:102-115     Frame::getKeyPointsInRegion (kd-tree radius search, unsorted, octave filter)
//   src/map_types/frame.h:129-136       Frame::predictScale
//   src/map_types/mappoint.h:99,146-177 getViewCos, getHammDescDistance_2 (float result)
//   src/basictypes/picoflann.h          KdTreeIndex<2>: build :150-163,238-345 (mean/variance split on <=100 samples,
//                                       planeSplit, std::sort fallback, leaves <= 10) and searchExactLevel :545-590
//
// The candidate ORDER is observable (a candidate that fails "d < best" updates the second-best, one that succeeds does not
// demote the old best), so the disc search cannot be replaced by a grid: the kernel walks the SAME kd-tree in the SAME
// order.  Division of work:
//   host   (uh_projmatch_set_frame)  builds the tree exactly like picoflann does — the reference also builds it on the CPU,
//          once per frame (Frame::create_kdtree, frame.h:124) — and uploads it flattened (24-byte nodes + leaf index list)
//   device (projmatch_kernel)        one lane per map point: visibility tests, predicted octave, iterative tree walk with an
//          explicit stack (the recursion's after-best-child / restore steps are stack records), Hamming distances against the
//          frame's descriptors, best / second-best rule.  ~10^3..10^4 map points x a few dozen candidates each: the kernel is
//          bound by dependent L2 round trips of the walk, not by bandwidth.
//   host   orders the per-point results into the DMatch list and runs filter_ambiguous_query (matcher.hip).
// Floating-point conventions are those of oracle/proj_oracle.cpp (float ops in source order, no contraction; cv::norm in
// double); predictScale's logf is libm's, reproduced on the device by glibc_sincosf.hpp.
#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "devframe.hpp"
#include "glibc_sincosf.hpp"

extern "C" int uh_filter_ambiguous(uh_dmatch* matches, int n, int by_train);

namespace {

constexpr int kLeafMax = 10;          // picoflann _maxLeafSize
constexpr int kMaxDepth = 90;         // deeper trees (pathological inputs) are refused: the walk stacks live in LDS
constexpr size_t kLdsBudget = 150 * 1024;   // of the CU's 160 KB
constexpr int kPmThreads = 768;       // at most twelve waves share the LDS copy of the frame (the launch picks 4 .. 12: see uh_projmatch_match; 170 registers per lane)
constexpr int kGroup = 64;            // lanes per map point: ONE point per wave (round 5; a leaf holds <= 10 keypoints, the drain uses all 64)
constexpr int kGroupsMax = kPmThreads / kGroup;   // groups (map points in flight) per workgroup, at most
constexpr int kPmMaxBlocks = 768;     // workgroups loop over chunks of 16 map points: the frame is staged once per workgroup
constexpr int kCandCap = 128;         // per-point list of disc hits between two drains (a leaf adds at most 10)

struct KdNodeDev {
    float divlow, divhigh;
    int left, right;        // -1/-1 = leaf
    int leaf_begin;
    short leaf_count, col;
};
static_assert(sizeof(KdNodeDev) == 24, "node layout");

// ---------------------------------------------------------------------------------------------------------- host: tree build
// Flat-array restatement of picoflann::KdTreeIndex<2>::build for (x,y) float points.  std::sort is used exactly where the
// reference uses it (same libstdc++ => same permutation of equal keys).
class KdBuilder {
public:
    std::vector<KdNodeDev> nodes;
    std::vector<uint32_t> leaf_idx;
    double root_box[4] = {0, 0, 0, 0};   // x.min x.max y.min y.max
    int max_depth = 0;

    void build(const float* xy, int n) {
        xy_ = xy;
        nodes.clear();
        leaf_idx.clear();
        leaf_idx.reserve(n);
        order_.resize(n);
        px_.resize(n); py_.resize(n);
        tmp_l_.resize(n + 1); tmp_r_.resize(n + 1);
        // (the same loop finds the exponent spread of the non-zero coordinates — see sample_moments; branch-free: a zero counts as neither
        // bound, a non-finite value shows as exponent 255, a denormal as exponent 0)
        uint32_t emin = 255, emax = 0;
        for (int i = 0; i < n; i++) {
            order_[i] = (uint32_t)i;
            const float x = xy[2 * (size_t)i], y = xy[2 * (size_t)i + 1];
            px_[i] = x; py_[i] = y;
            uint32_t bx, by;
            std::memcpy(&bx, &x, 4); std::memcpy(&by, &y, 4);
            const uint32_t ex = (bx >> 23) & 0xffu, ey = (by >> 23) & 0xffu;
            const bool zx = (bx & 0x7fffffffu) == 0, zy = (by & 0x7fffffffu) == 0;
            emin = std::min(emin, std::min(zx ? 255u : ex, zy ? 255u : ey));
            emax = std::max(emax, std::max(zx ? 0u : ex, zy ? 0u : ey));
        }
        max_depth = 0;
        if (n == 0) return;
        exact_sums_ = emax != 255 /* finite */ && emin >= 1 /* no denormals */ && emax - emin <= 10;   // (a cloud of zeros only: the unsigned difference wraps and the ordered chains stay — equally right)
        Box root;
        bounds(0, n, root);
        nodes.reserve(2 * (size_t)n + 2);
        nodes.push_back(KdNodeDev{0, 0, -1, -1, 0, 0, 0});
        split(0, 0, n, root, 1);
        root_box[0] = root.lo[0]; root_box[1] = root.hi[0]; root_box[2] = root.lo[1]; root_box[3] = root.hi[1];
    }

private:
    struct Box { double lo[2], hi[2]; };
    const float* xy_ = nullptr;
    // the index permutation the reference partitions in place, with the points' coordinates carried along in the same order: every
    // scan below reads contiguous floats instead of gathering through the permutation
    std::vector<uint32_t> order_;
    std::vector<float> px_, py_;
    std::vector<int> tmp_l_, tmp_r_;
    float* coords(int d) { return d == 0 ? px_.data() : py_.data(); }

    void bounds(int b, int e, Box& box) const {
        float lx = px_[b], hx = px_[b], ly = py_[b], hy = py_[b];
        for (int k = b + 1; k < e; k++) {
            const float x = px_[k], y = py_[k];
            lx = x < lx ? x : lx; hx = x > hx ? x : hx;
            ly = y < ly ? y : ly; hy = y > hy ? y : hy;
        }
        box.lo[0] = lx; box.hi[0] = hx; box.lo[1] = ly; box.hi[1] = hy;
    }

    // picoflann.h:362-391: mean / variance over at most ~100 evenly spaced samples (float squares, double sums IN SAMPLE ORDER: four
    // dependent addition chains, ~a third of the build).  Round 6: when every coordinate of the cloud lies within a factor 2^10 of every
    // other (exponent spread <= 10: any extractor output — pixels 19 .. 4095 — does), no partial sum of <= 256 samples is ever rounded
    // (x: multiples of 2^(e_min-23) below 2^(e_max+9); fl(x*x): multiples of 2^(2 e_min-23) below 2^(2 e_max+10) — both fit 53 bits), so
    // the sums are the exact real sums in ANY order and four partial accumulators per sum give the same bits; otherwise the chains stay.
    bool exact_sums_ = false;
    void sample_moments(int b, int e, double mean[2], double var[2]) const {
        double s1[2] = {0, 0}, s2[2] = {0, 0};
        int step = 1, cnt = 0;
        if (e - b >= 200) step = (e - b) / 100;
        if (exact_sums_ && avx512_ && step == 1) {
            cnt = e - b;
            moments_avx512(px_.data() + b, py_.data() + b, cnt, s1, s2);
        } else if (exact_sums_) {
            double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
            cnt = (e - b + step - 1) / step;
            const float* qx = px_.data() + b; const float* qy = py_.data() + b;
            int k = 0;
            for (; k + 4 <= cnt; k += 4) {
#pragma GCC unroll 4
                for (int u = 0; u < 4; u++) {
                    const float x = qx[(size_t)(k + u) * step], y = qy[(size_t)(k + u) * step];
                    a1[u] += x; a2[u] += x * x; c1[u] += y; c2[u] += y * y;
                }
            }
            for (; k < cnt; k++) { const float x = qx[(size_t)k * step], y = qy[(size_t)k * step]; a1[0] += x; a2[0] += x * x; c1[0] += y; c2[0] += y * y; }
            s1[0] = (a1[0] + a1[1]) + (a1[2] + a1[3]); s2[0] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
            s1[1] = (c1[0] + c1[1]) + (c1[2] + c1[3]); s2[1] = (c2[0] + c2[1]) + (c2[2] + c2[3]);
        } else {
            for (int i = b; i < e; i += step, cnt++) {
                const float x = px_[i], y = py_[i];
                s1[0] += x; s2[0] += x * x;
                s1[1] += y; s2[1] += y * y;
            }
        }
        const double inv = 1. / double(cnt);
        for (int d = 0; d < 2; d++) {
            mean[d] = s1[d] * inv;
            var[d] = s2[d] * inv - mean[d] * mean[d];
        }
    }

    // exact sums (see above) of cnt contiguous samples, eight doubles per accumulator: fl(x), fl(x * x) in float as picoflann squares it
    __attribute__((target("avx512f,avx512vl"))) static void moments_avx512(const float* qx, const float* qy, int cnt, double s1[2], double s2[2]) {
        __m512d a1 = _mm512_setzero_pd(), a2 = a1, c1 = a1, c2 = a1;
        int k = 0;
        for (; k + 8 <= cnt; k += 8) {
            const __m256 x = _mm256_loadu_ps(qx + k), y = _mm256_loadu_ps(qy + k);
            a1 = _mm512_add_pd(a1, _mm512_cvtps_pd(x)); a2 = _mm512_add_pd(a2, _mm512_cvtps_pd(_mm256_mul_ps(x, x)));
            c1 = _mm512_add_pd(c1, _mm512_cvtps_pd(y)); c2 = _mm512_add_pd(c2, _mm512_cvtps_pd(_mm256_mul_ps(y, y)));
        }
        if (k < cnt) {
            const __mmask8 m = (__mmask8)((1u << (cnt - k)) - 1u);
            const __m256 x = _mm256_maskz_loadu_ps(m, qx + k), y = _mm256_maskz_loadu_ps(m, qy + k);   // (zeros add nothing)
            a1 = _mm512_add_pd(a1, _mm512_cvtps_pd(x)); a2 = _mm512_add_pd(a2, _mm512_cvtps_pd(_mm256_mul_ps(x, x)));
            c1 = _mm512_add_pd(c1, _mm512_cvtps_pd(y)); c2 = _mm512_add_pd(c2, _mm512_cvtps_pd(_mm256_mul_ps(y, y)));
        }
        s1[0] = _mm512_reduce_add_pd(a1); s2[0] = _mm512_reduce_add_pd(a2);
        s1[1] = _mm512_reduce_add_pd(c1); s2[1] = _mm512_reduce_add_pd(c2);
    }

    void swap_items(int i, int j) {
        std::swap(order_[i], order_[j]); std::swap(px_[i], px_[j]); std::swap(py_[i], py_[j]);
    }

    // One Hoare pass of picoflann.h:403-424 over [b, e) with the predicate "belongs left" = (v < cut) or (v <= cut): the reference
    // walks l up to the first element that belongs right, r down to the last that belongs left, swaps them and repeats until the
    // two meet.  That is: with m = the number of elements that belong left, the misplaced elements in front of b + m (ascending)
    // are exchanged pairwise with the misplaced elements from b + m on (descending) — the same permutation, obtained here with a
    // count and two branch-free compactions instead of two data-dependent scans (the scans mispredict on every other element).
    template <bool INCLUSIVE>
    int hoare_pass(int b, int e, const float* v, float cut) {
        if (avx512_) return hoare_pass_avx512<INCLUSIVE>(b, e, v, cut);
        int m = 0;
        for (int i = b; i < e; i++) m += INCLUSIVE ? (v[i] <= cut) : (v[i] < cut);
        const int mid = b + m;
        if (m == 0 || mid == e) return mid;   // one side is empty: nothing is misplaced (the usual outcome of the second pass: no point EQUALS the cut)
        int* L = tmp_l_.data();
        int* R = tmp_r_.data();
        int nl = 0, nr = 0;
        for (int i = b; i < mid; i++) { L[nl] = i; nl += INCLUSIVE ? !(v[i] <= cut) : !(v[i] < cut); }
        for (int j = e - 1; j >= mid; j--) { R[nr] = j; nr += INCLUSIVE ? (v[j] <= cut) : (v[j] < cut); }
        for (int k = 0; k < nl; k++) swap_items(L[k], R[k]);   // (nl == nr)
        return mid;
    }
    // The same pass with AVX-512 where the host has it (round 6; the count and the two index lists are ~half of a build): sixteen
    // predicates per compare, the misplaced positions leave through vpcompressd.  Same permutation: the k-th misplaced position of the front
    // part (ascending) meets the k-th of the back part counted from the end.
    bool avx512_ = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && !getenv("UH_KD_NO_AVX512");
    template <bool INCLUSIVE>
    __attribute__((target("avx512f,avx512vl,popcnt"))) int hoare_pass_avx512(int b, int e, const float* v, float cut) {
        const __m512 vc = _mm512_set1_ps(cut);
#define left_mask(i_, n_) _mm512_mask_cmp_ps_mask((__mmask16)((1u << (n_)) - 1u), _mm512_maskz_loadu_ps((__mmask16)((1u << (n_)) - 1u), v + (i_)), vc, INCLUSIVE ? _CMP_LE_OQ : _CMP_LT_OQ)   /* "belongs left" of v[i .. i + n), n <= 16 */
        int m = 0;
        for (int i = b; i < e; i += 16) m += __builtin_popcount(left_mask(i, std::min(16, e - i)));
        const int mid = b + m;
        if (m == 0 || mid == e) return mid;
        int* L = tmp_l_.data();
        int* R = tmp_r_.data();
        int nl = 0, nr = 0;
        const __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        for (int i = b; i < mid; i += 16) {
            const int n = std::min(16, mid - i);
            const __mmask16 mis = (__mmask16)(~left_mask(i, n) & ((1u << n) - 1u));
            _mm512_mask_compressstoreu_epi32(L + nl, mis, _mm512_add_epi32(lane, _mm512_set1_epi32(i)));
            nl += __builtin_popcount(mis);
        }
        for (int i = mid; i < e; i += 16) {   // ascending here; paired from the end below
            const int n = std::min(16, e - i);
            const __mmask16 mis = left_mask(i, n);
            _mm512_mask_compressstoreu_epi32(R + nr, mis, _mm512_add_epi32(lane, _mm512_set1_epi32(i)));
            nr += __builtin_popcount(mis);
        }
        for (int k = 0; k < nl; k++) swap_items(L[k], R[nr - 1 - k]);   // (nl == nr)
#undef left_mask
        return mid;
    }

    // picoflann.h:238-345.  `box` is in/out: on return it is the tight box of the subtree.
    void split(int node, int b, int e, Box& box, int depth) {
        max_depth = std::max(max_depth, depth);
        const int count = e - b;
        if (count <= kLeafMax) {
            nodes[node].leaf_begin = (int)leaf_idx.size();
            nodes[node].leaf_count = (short)count;
            for (int i = b; i < e; i++) leaf_idx.push_back(order_[i]);
            bounds(b, e, box);
            return;
        }
        const int lch = (int)nodes.size();
        nodes.push_back(KdNodeDev{0, 0, -1, -1, 0, 0, 0});
        nodes.push_back(KdNodeDev{0, 0, -1, -1, 0, 0, 0});
        double mean[2], var[2];
        sample_moments(b, e, mean, var);
        const int dim = var[1] > var[0] ? 1 : 0;
        double cut = mean[dim];
        // picoflann.h:403-424: two Hoare passes -> [< cut | == cut | > cut]
        const int lim1 = hoare_pass<false>(b, e, coords(dim), (float)cut) - b;
        const int lim2 = hoare_pass<true>(b + lim1, e, coords(dim), (float)cut) - b;
        int at = count / 2;
        if (lim1 > count / 2) at = lim1;
        else if (lim2 < count / 2) at = lim2;
        if (lim1 == count || lim2 == 0) at = count / 2;
        if (at < kLeafMax || count - at < kLeafMax) {
            // std::sort exactly where the reference uses it (same libstdc++ => same permutation of equal keys).  Up to 16 elements — most of
            // the nodes that come here hold 11..19 points — libstdc++'s std::sort IS its insertion sort (introsort's loop does nothing below
            // _S_threshold = 16, __final_insertion_sort is one guarded __insertion_sort): done here on the carried coordinates, no gathers
            // through the permutation in the comparisons and none behind them.
            if (count <= 16) {
                float* key = coords(dim); float* oth = coords(dim ^ 1);
                for (int i = b + 1; i < e; i++) {
                    const float k = key[i], o = oth[i];
                    const uint32_t id = order_[i];
                    int j = i;
                    if (k < key[b]) {   // (libstdc++ moves the whole prefix when the element is smaller than the first one: same result)
                        for (; j > b; j--) { key[j] = key[j - 1]; oth[j] = oth[j - 1]; order_[j] = order_[j - 1]; }
                    } else {
                        for (; k < key[j - 1]; j--) { key[j] = key[j - 1]; oth[j] = oth[j - 1]; order_[j] = order_[j - 1]; }
                    }
                    key[j] = k; oth[j] = o; order_[j] = id;
                }
            } else {
                const float* xy = xy_;
                std::sort(order_.begin() + b, order_.begin() + e, [xy, dim](const uint32_t& p, const uint32_t& q) { return xy[2 * (size_t)p + dim] < xy[2 * (size_t)q + dim]; });
                for (int i = b; i < e; i++) { px_[i] = xy[2 * (size_t)order_[i]]; py_[i] = xy[2 * (size_t)order_[i] + 1]; }
            }
            at = count / 2;
            cut = coords(dim)[b + at];
        }
        Box lbox = box, rbox = box;
        lbox.hi[dim] = cut;
        split(lch, b, b + at, lbox, depth + 1);
        lbox.hi[dim] = cut;
        rbox.lo[dim] = cut;
        split(lch + 1, b + at, e, rbox, depth + 1);
        KdNodeDev& nd = nodes[node];
        nd.left = lch; nd.right = lch + 1; nd.col = (short)dim;
        nd.divlow = (float)lbox.hi[dim];
        nd.divhigh = (float)rbox.lo[dim];
        for (int d = 0; d < 2; d++) { box.lo[d] = std::min(lbox.lo[d], rbox.lo[d]); box.hi[d] = std::max(lbox.hi[d], rbox.hi[d]); }
    }
};

// ---------------------------------------------------------------------------------------------------------- device
struct PmFrame {
    const uint64_t* kp_desc;        // n_kpts x 4
    const KdNodeDev* nodes;
    const float4* leaf_rec;         // n_kpts records in LEAF order: {x, y, bits(keypoint << 4 | octave), 0} — what a leaf visit needs, in one 16-byte read
                                    // (round 3 chased leaf index -> coordinates -> octave through three dependent LDS reads per visit)
    double box[4];
    const float* scale; int n_levels; int n_kpts;
    float fx, fy, cx, cy, min_x, min_y, max_x, max_y;
    float log_scale;
};

struct PmPoints {
    int n;
    // one 64-byte record per candidate, in the pinned staging block (read by the kernel in place): position (3 floats), then normal (3),
    // min / max distance — or, PREV mode, the previous frame's keypoint octave in word 3 —, then the 32-byte descriptor.  Five separate
    // arrays (rounds 3-5a) were five small reads over the host link per candidate — 15 000 requests for 3000 map points, ~6 us before a walk
    // could start; one record is one 64-byte request.
    const uint4* rec;
    int* best_kp; float* best_dist; unsigned char* visible;
    float4* pos_out;   // NULL, or HBM: every candidate's position again (uh_track_pose's look-ups gather from it instead of from pinned memory)
    const uint4* aux_in; uint4* aux_out;   // with pos_out: a 16-byte record per candidate the caller staged in pinned memory ({id, map row, weight, -}), moved to HBM likewise
};

struct PmPose { float T[12]; float cc[3]; };
struct PmDyn { PmPose ps; float radius; int skip; };   // a search whose pose / radius an earlier launch of the same stream decides (uh_track_pose); skip: nothing to do
// The call's hand-over, done by the kernel itself (round 5; rounds 3-4: a copy launch in front and a publish launch behind, ~6 us of kernel
// and a launch gap each): the candidates' arrays are read from the pinned staging block in place, the per-point results go to HBM with
// write-through (agent-scope) stores, every workgroup takes a ticket behind its acknowledged stores, and the workgroup that draws the last
// one copies the result block into pinned memory 8 bytes a lane, clears the ticket and the walk-overflow word for the next call and posts
// the completion word (one system-scope release).
struct PmPublish {
    unsigned* ticket;                          // HBM, 0 between calls
    const unsigned long long* dev_out; unsigned long long* host_out; unsigned n8;   // the result block in HBM, its pinned twin, 8-byte words
    unsigned long long* host_done; unsigned long long word;                        // pinned: completion word; word [1] low half receives the overflow flag
};

__device__ __forceinline__ float logf_cr(float x) { return uh_sincosf::logf_glibc(x); }   // libm's logf, bit for bit
__device__ __forceinline__ double readlane_f64(double v, int lane_uniform) {   // lane index must be wave-uniform
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_uniform);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_uniform);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Walk records (one per tree level at most): (other child << 2) | (col << 1) | state, with the other child's mindistsq, the
// cut distance and the float dists[col] to restore; state 0 = other child still to be searched, 1 = being searched (restore
// dists[col] when it is done).  The node to visit next is carried in registers (cur / cur_m), not pushed.
//
// Work decomposition: kGroup = 16 lanes per map point, 4 points per wave, 4 waves per workgroup; a workgroup stages the frame
// in LDS once and then loops over chunks of 16 map points.  The lanes of a group walk
// the tree together (identical control flow); in a leaf each lane tests ONE of its <= 10 keypoints, the hits are appended in
// leaf order (group ballot + prefix popcount) to the point's candidate list.  The list is drained by the 16 lanes in parallel
// (descriptor fetch + Hamming), then scanned in order for the best / second-best rule.  Compared with one lane per point this
// cuts the divergence (max over 4 instead of 64 walks per wave), parallelises the leaf and the L2 fetches, and fills 16x
// more SIMDs.
//
// LDS layout of a workgroup: [nodes | leaf records] when the frame fits (IN_LDS), then per group the
// walk stack (double m, int rec per level) and the candidate list ((keypoint << 4) | octave, then the Hamming distance).
// A frame of 2000 keypoints is ~40 KB, of 4000 ~80 KB: every node / leaf / coordinate access of the walk is an LDS access;
// only descriptors of disc hits come from L2.
// PREV = false: Map::matchFrameToMapPoints (map.cpp:689-759).  PREV = true: the tracker's search against the previous frame's
// keypoints (system.cpp:5930-6460): Frame::project visibility, the keypoint's own octave as the only admissible level, radius
// maxRepjDist * scaleFactors[octave], best/second without demotion from (float)(minDescDist + 0.01), accept best < 0.7 * second.
// LANE_STACK (levels <= 64, i.e. always but for pathological trees): the walk records live across the LANES of the point's wave — record i in
// lane i, pushed by a predicated move, popped by v_readlane with the (wave-uniform) stack pointer — instead of in LDS: a pop was three to four
// dependent LDS round trips of an iteration that has five.
template <bool IN_LDS, bool PREV, bool LANE_STACK>
__global__ __launch_bounds__(kPmThreads) void projmatch_kernel(PmFrame f, PmPoints mp, PmPose ps, float minDescDist, float maxRepjDist,
                                                               int n_nodes, int levels, int* overflow, PmPublish pub, const PmDyn* __restrict__ dyn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (dyn) { ps = dyn->ps; maxRepjDist = dyn->radius; }   // (uniform: scalar loads)
    const int lane = threadIdx.x, g = lane / kGroup, gl = lane % kGroup, gw = (lane & 63) / kGroup;   // gw: group within its wave
    const int nthr = (int)blockDim.x, gpw = nthr / kGroup;   // threads and groups of THIS launch's workgroups
    // measurement (UH_PM_CLK): shader-clock stamps of workgroup 0, thread 0 in the status block behind the overflow word
    long long* const clk = (overflow[1] == 0x434c4b && blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<long long*>(overflow + 4) : nullptr;
    if (clk) clk[0] = __builtin_readcyclecounter();
    const int n = f.n_kpts;
    size_t off = 0;
    KdNodeDev* s_nodes = reinterpret_cast<KdNodeDev*>(smem);
    if (IN_LDS) off += ((size_t)n_nodes * sizeof(KdNodeDev) + 15) & ~(size_t)15;
    float4* s_rec = reinterpret_cast<float4*>(smem + off);
    if (IN_LDS) off += (size_t)n * 16;
    double* st_m = reinterpret_cast<double*>(smem + off) + (size_t)g * levels;
    off += (size_t)levels * gpw * 8;
    double* st_c = reinterpret_cast<double*>(smem + off) + (size_t)g * levels;
    off += (size_t)levels * gpw * 8;
    int* st_rec = reinterpret_cast<int*>(smem + off) + (size_t)g * levels;
    off += (size_t)levels * gpw * 4;
    float* st_d = reinterpret_cast<float*>(smem + off) + (size_t)g * levels;
    off += (size_t)levels * gpw * 4;
    unsigned int* s_cand = reinterpret_cast<unsigned int*>(smem + off) + (size_t)g * kCandCap;
    off += (size_t)kCandCap * gpw * 4;
    int* s_hd = reinterpret_cast<int*>(smem + off) + (size_t)g * kCandCap;
    // The candidate of this wave's FIRST chunk is requested before the frame is staged: it lies in pinned host memory (uh_projmatch_match
    // hands the points over in place), a host-link round trip that used to start only behind the staging's barrier.
    struct Cand { float P0, P1, P2, n0, n1, n2, mind, maxd; int oct; uint64_t q0, q1, q2, q3; };
    auto load_cand = [&](int mc) {
        const uint4* r = mp.rec + 4 * (size_t)mc;
        const uint4 a = r[0], b = r[1], d0 = r[2], d1 = r[3];
        Cand c;
        c.P0 = __uint_as_float(a.x); c.P1 = __uint_as_float(a.y); c.P2 = __uint_as_float(a.z);
        c.oct = (int)a.w; c.n0 = __uint_as_float(a.w); c.n1 = __uint_as_float(b.x); c.n2 = __uint_as_float(b.y);
        c.mind = __uint_as_float(b.z); c.maxd = __uint_as_float(b.w);
        c.q0 = ((uint64_t)d0.y << 32) | d0.x; c.q1 = ((uint64_t)d0.w << 32) | d0.z; c.q2 = ((uint64_t)d1.y << 32) | d1.x; c.q3 = ((uint64_t)d1.w << 32) | d1.z;
        return c;
    };
    const int m_first = (int)blockIdx.x * gpw + g;
    const Cand c_first = load_cand(m_first < mp.n ? m_first : 0);
    if (IN_LDS) {
        // 16-byte loads, eight in flight per lane and round (the node array is 24 n bytes in a 256-byte aligned block: the last chunk may
        // run 8 bytes into its padding, which the LDS area's rounding covers)
        const uint4* gn = reinterpret_cast<const uint4*>(f.nodes);
        uint4* sn = reinterpret_cast<uint4*>(s_nodes);
        constexpr int U = 8;
        const int n16 = (n_nodes * (int)sizeof(KdNodeDev) + 15) / 16;
        for (int i0 = lane; i0 < n16; i0 += U * nthr) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; v[u] = gn[i < n16 ? i : 0]; }
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; if (i < n16) sn[i] = v[u]; }
        }
        for (int i0 = lane; i0 < n; i0 += U * nthr) {
            float4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; a[u] = f.leaf_rec[i < n ? i : 0]; }
#pragma unroll
            for (int u = 0; u < U; u++) { const int i = i0 + u * nthr; if (i < n) s_rec[i] = a[u]; }
        }
        __syncthreads();
    }
    if (clk) clk[1] = __builtin_readcyclecounter();
  for (int chunk = blockIdx.x; chunk * gpw < mp.n; chunk += gridDim.x) {
    const int m = chunk * gpw + g;
    const bool live = m < mp.n;
    const int mc = live ? m : 0;
    const Cand cd = chunk == (int)blockIdx.x ? c_first : load_cand(mc);
    // ---- visibility tests and search parameters (identical in the lanes of the group)
    bool vis = false;
    float px = 0, py = 0;
    int predicted = 0;
    double worst = 0;
    if constexpr (PREV) {
        const float P0 = cd.P0, P1 = cd.P1, P2 = cd.P2;
        const float* T = ps.T;
        const float z = P0 * T[8] + P1 * T[9] + P2 * T[10] + T[11];       // Frame::project (frame.h:140-161)
        const float x = P0 * T[0] + P1 * T[1] + P2 * T[2] + T[3];
        const float y = P0 * T[4] + P1 * T[5] + P2 * T[6] + T[7];
        const float iz = (float)(1. / z);
        px = ((f.fx * x) * iz) + f.cx; py = ((f.fy * y) * iz) + f.cy;
        vis = live && !(z < 0) && (px >= f.min_x && py >= f.min_y && px < f.max_x && py < f.max_y);
        predicted = cd.oct;
        if (vis) {
            const double radius = (double)(maxRepjDist * f.scale[predicted]);
            worst = radius * radius;
        }
    } else {
        const float P0 = cd.P0, P1 = cd.P1, P2 = cd.P2;
        float v0 = ps.cc[0] - P0, v1 = ps.cc[1] - P1, v2 = ps.cc[2] - P2;   // getViewCos
        const double s = 1. / sqrt((double)v0 * v0 + (double)v1 * v1 + (double)v2 * v2);
        v0 = (float)(v0 * s); v1 = (float)(v1 * s); v2 = (float)(v2 * s);
        const float viewCos = v0 * cd.n0 + v1 * cd.n1 + v2 * cd.n2;
        const float* T = ps.T;
        const float x = T[0] * P0 + T[1] * P1 + T[2] * P2 + T[3];
        const float y = T[4] * P0 + T[5] * P1 + T[6] * P2 + T[7];
        const float z = T[8] * P0 + T[9] * P1 + T[10] * P2 + T[11];
        const float dist = (float)sqrt((double)x * x + (double)y * y + (double)z * z);
        const float maxd = cd.maxd;
        const float iz = (float)(1. / z);
        px = x * f.fx * iz + f.cx; py = y * f.fy * iz + f.cy;
        vis = live && !(viewCos < 0.5) && !(z < 0) && (0.8f * cd.mind < dist && dist < 1.2f * maxd) &&
              (px > f.min_x && py > f.min_y && px < f.max_x && py < f.max_y);
        if (vis) {
            const int ns = (int)ceilf(logf_cr(maxd / dist) / f.log_scale);
            predicted = ns < 0 ? 0 : (ns >= f.n_levels ? f.n_levels - 1 : ns);
            float radius_scale = f.scale[predicted];
            if (viewCos < 0.98) radius_scale = (float)(radius_scale * 1.6);
            const double radius = (double)(radius_scale * maxRepjDist);
            worst = radius * radius;
        }
    }
    int best_kp = -1;
    float best_d = PREV ? (float)(minDescDist + 0.01) : 3.402823466e+38f, second_d = 3.402823466e+38f;
    const int oct_lo = PREV ? predicted : predicted - 1;
    int bestLevel = 0, bestLevel2 = -1;
    bool ovf = false;
    const uint64_t q0 = cd.q0, q1 = cd.q1, q2 = cd.q2, q3 = cd.q3;
    int ncand = 0;
    // drain: Hamming distances of the listed hits (16 lanes, one hit each per round), then the in-order best / second rule
    auto drain = [&](int cnt) {
        for (int k = gl; k < cnt; k += kGroup) {
            const uint64_t* kd = f.kp_desc + 4 * (size_t)(s_cand[k] >> 4);
            s_hd[k] = __popcll(q0 ^ kd[0]) + __popcll(q1 ^ kd[1]) + __popcll(q2 ^ kd[2]) + __popcll(q3 ^ kd[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = 0; k < cnt; k++) {
            const float hd = (float)s_hd[k];
            const unsigned int e = s_cand[k];
            const int oc = (int)(e & 15u);
            const bool lt = PREV ? true : hd < minDescDist;
            const bool isbest = lt && hd < best_d, issecond = lt && !isbest && hd < second_d;
            best_d = isbest ? hd : best_d; best_kp = isbest ? (int)(e >> 4) : best_kp; bestLevel = isbest ? oc : bestLevel;
            second_d = issecond ? hd : second_d; bestLevel2 = issecond ? oc : bestLevel2;
        }
        __builtin_amdgcn_wave_barrier();   // the list is rewritten after this
    };
    if (vis && n > 0) {
        // computeInitialDistances (float accumulator)
        double dd0 = 0, dd1 = 0;
        float distsq = 0.f;
        {
            const double ex = px, ey = py;
            if (ex < f.box[0]) { const double d = ex - f.box[0]; dd0 = d * d; distsq += dd0; }
            if (ex > f.box[1]) { const double d = ex - f.box[1]; dd0 = d * d; distsq += dd0; }
            if (ey < f.box[2]) { const double d = ey - f.box[2]; dd1 = d * d; distsq += dd1; }
            if (ey > f.box[3]) { const double d = ey - f.box[3]; dd1 = d * d; distsq += dd1; }
        }
        // Iterative searchExactLevel.  At an internal node everything the recursion does AFTER its best child returns is
        // already known when the node is entered: dst = (float)dists[col] (deeper levels only ever leave dists[col] rounded
        // to float, which the (float) cast hides), hence mindistsq' = mindistsq + cut - dst and whether the other child
        // will be searched at all.  Only nodes whose other child WILL be searched leave a record (most do not); the record
        // is turned into "restore dists[col]" in place when the other child is entered.
        int n_iter = 0, n_leaf = 0;
        // ---- Round 5: the walk of ONE point spread over the lanes of its wave.  This is a RADIUS search (`worst` never changes), and what a
        // node sees — mindistsq and the two dists[] entries — depends on the path from the root only (entering an "other" child sets
        // dists[col] = cut, a best child inherits everything; what a finished subtree leaves behind is the old value rounded to float, which
        // every later use casts to float anyway), so the subtrees can be expanded side by side: lane 0 starts at the root, every lane follows
        // its best-child chain down to a leaf, and whenever the other child must be searched too it is handed to a free lane (all of a level's
        // hand-overs at once: ballot, prefix count, ds_bpermute of the child's state).  A lane's path is its KEY — bit (63 - depth) set
        // where it was the "other" child — and the leaves' depth-first order, which the candidate list must keep (best / second without
        // demotion is order-dependent), is the order of the keys.  ~tree-depth steps of ~60 instructions instead of one ~90-instruction
        // step per visited node; the launch ended with its heaviest point (radius 15 px x scale x 1.6: 60+ visited nodes).
        // More than 64 leaves (or levels): the serial walk below.
        bool walked = false;
        if constexpr (LANE_STACK) {
            int node = gl == 0 ? 0 : -1;      // >= 0: node to expand; -1: free lane; -2: this lane has reached its leaf
            double pm = (double)distsq, pd0 = dd0, pd1 = dd1;
            unsigned long long key = 0;
            int lbeg = 0, lcnt = 0;
            int nl = 1;                       // lanes in use (wave-uniform)
            bool fits = true;
            for (int depth = 0;; ++depth) {
                const bool act = node >= 0;
                if (__ballot(act) == 0) break;
                if (depth >= 64) { fits = false; break; }
                const int nsafe = act ? node : 0;
                const uint2* np = reinterpret_cast<const uint2*>(IN_LDS ? s_nodes + nsafe : f.nodes + nsafe);
                uint2 w0 = np[0], w1 = np[1], w2 = np[2];
                asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(w1.x), "+v"(w1.y), "+v"(w2.x), "+v"(w2.y));
                const float divlow = __uint_as_float(w0.x), divhigh = __uint_as_float(w0.y);
                const int left = (int)w1.x, right = (int)w1.y;
                const int col = (int)(w2.y >> 16);
                const bool isleaf = left < 0;
                if (act && isleaf) { lbeg = (int)w2.x; lcnt = (int)(w2.y & 0xFFFFu); node = -2; }
                const bool inner = act && !isleaf;
                const double val = col == 0 ? px : py;
                const double diff1 = val - divlow, diff2 = val - divhigh;
                const bool go_left = diff1 + diff2 < 0;
                const double cut = go_left ? diff2 * diff2 : diff1 * diff1;
                const float dst = (float)(col == 0 ? pd0 : pd1);
                const double m2 = pm + cut - dst;
                const bool wantB = inner && (m2 * 1.0 <= worst);
                const unsigned long long mb = __ballot(wantB);
                const int cntB = __popcll(mb);
                if (nl + cntB > kGroup) { fits = false; break; }
                if (cntB) {
                    int src = gl, k = 0;      // receiving lane nl + k pulls from the k-th lane that hands a child over
                    for (unsigned long long t = mb; t; t &= t - 1, ++k) { const int b = __builtin_ctzll(t); src = gl == nl + k ? b : src; }
                    const int bnode = go_left ? right : left;
                    const double bd0 = col == 0 ? cut : pd0, bd1 = col == 0 ? pd1 : cut;
                    const unsigned long long bkey = key | (1ull << (63 - depth));
                    const int sa = src << 2;
                    auto pull64 = [&](unsigned long long v) {
                        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(sa, (int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_ds_bpermute(sa, (int)(unsigned)(v >> 32));
                        return ((unsigned long long)hi << 32) | lo;
                    };
                    const int r_node = __builtin_amdgcn_ds_bpermute(sa, bnode);
                    const double r_m = __longlong_as_double((long long)pull64((unsigned long long)__double_as_longlong(m2)));
                    const double r_d0 = __longlong_as_double((long long)pull64((unsigned long long)__double_as_longlong(bd0)));
                    const double r_d1 = __longlong_as_double((long long)pull64((unsigned long long)__double_as_longlong(bd1)));
                    const unsigned long long r_key = pull64(bkey);
                    if (gl >= nl && gl < nl + cntB) { node = r_node; pm = r_m; pd0 = r_d0; pd1 = r_d1; key = r_key; }
                    nl += cntB;
                }
                if (inner) node = go_left ? left : right;   // the best child: same mindistsq, same dists, same key
                ++n_iter;
            }
            if (fits) {
                walked = true;
                n_leaf = nl;
                // every used lane holds one leaf.  Its slots' place in the depth-first order: the leaf counts of all leaves with a smaller key
                int off = 0, total = 0;
                for (int j = 0; j < nl; j++) {
                    const unsigned long long kj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), j) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, j);
                    const int cj = __builtin_amdgcn_readlane(lcnt, j);
                    off += kj < key ? cj : 0;
                    total += cj;
                }
                for (int s0 = 0; s0 < total; s0 += kGroup) {
                    const int sidx = s0 + gl;
                    int kp = -1;
                    for (int j = 0; j < nl; j++) {
                        const int oj = __builtin_amdgcn_readlane(off, j), cj = __builtin_amdgcn_readlane(lcnt, j), bj = __builtin_amdgcn_readlane(lbeg, j);
                        kp = (sidx >= oj && sidx < oj + cj) ? bj + (sidx - oj) : kp;
                    }
                    bool hit = false;
                    unsigned int id = 0;
                    int oc = 0;
                    if (kp >= 0) {
                        const float4 c = IN_LDS ? s_rec[kp] : f.leaf_rec[kp];
                        const unsigned io = __float_as_uint(c.z);
                        id = io >> 4; oc = (int)(io & 15u);
                        const double dx = px - c.x;
                        double sqd = dx * dx;
                        if (!(sqd > worst)) { const double dy = py - c.y; sqd += dy * dy; }
                        hit = sqd < worst && oc >= oct_lo && oc <= predicted;
                    }
                    const unsigned long long hm = __ballot(hit);
                    if (hit) s_cand[ncand + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u))] = (id << 4) | (unsigned int)oc;
                    ncand += __popcll(hm);
                    if (ncand > kCandCap - kGroup) { drain(ncand); ncand = 0; }
                }
            }
        }
        int sp = 0;
        int l_rec = 0; double l_m = 0, l_c = 0; float l_d = 0;   // LANE_STACK: this lane's walk record (lane = stack slot)
        int cur = walked ? -1 : 0;   // node to visit (-1: take the top record; walked: nothing left to do)
        double cur_m = (double)distsq;
        for (;;) {
            if (walked) break;
            ++n_iter;
            if (cur < 0) {
                if (sp == 0) break;
                const int top = __builtin_amdgcn_readfirstlane(sp) - 1;   // (one point per wave: the stack pointer is wave-uniform)
                const int rec = LANE_STACK ? __builtin_amdgcn_readlane(l_rec, top) : st_rec[sp - 1];
                const int col = (rec >> 1) & 1;
                if (rec & 1) {       // the other child's subtree is done: dists[col] = dst
                    const double dstd = (double)(LANE_STACK ? __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(l_d), top)) : st_d[sp - 1]);
                    if (col == 0) dd0 = dstd; else dd1 = dstd;
                    --sp;
                    continue;
                }
                const double cutv = LANE_STACK ? readlane_f64(l_c, top) : st_c[sp - 1];   // enter the other child: dists[col] = cut
                if (col == 0) dd0 = cutv; else dd1 = cutv;
                cur = rec >> 2;
                cur_m = LANE_STACK ? readlane_f64(l_m, top) : st_m[sp - 1];
                if constexpr (LANE_STACK) { if (gl == top) l_rec = rec | 1; }
                else if (gl == 0) st_rec[sp - 1] = rec | 1;
            }
            // the whole 24-byte node in ONE round trip: three 8-byte reads issued together (left as `nd = s_nodes[cur]` the compiler fetched the
            // fields where they are used — left / right, then the planes, then the column, then the leaf range: four to five dependent LDS
            // round trips per tree step, ~500 of its ~650 clocks)
            KdNodeDev nd;
            {
                const uint2* np = reinterpret_cast<const uint2*>(IN_LDS ? s_nodes + cur : f.nodes + cur);
                uint2 w0 = np[0], w1 = np[1], w2 = np[2];
                asm volatile("" : "+v"(w0.x), "+v"(w0.y), "+v"(w1.x), "+v"(w1.y), "+v"(w2.x), "+v"(w2.y));   // (all six words are needed HERE)
                nd.divlow = __uint_as_float(w0.x); nd.divhigh = __uint_as_float(w0.y);
                nd.left = (int)w1.x; nd.right = (int)w1.y; nd.leaf_begin = (int)w2.x;
                nd.leaf_count = (short)(w2.y & 0xFFFFu); nd.col = (short)(w2.y >> 16);
            }
            if (nd.left < 0) {   // leaf: lane i of the group tests keypoint i; hits are appended in leaf order
                ++n_leaf;
                bool hit = false;
                unsigned int id = 0;
                int oc = 0;
                if (gl < nd.leaf_count) {
                    const float4 c = IN_LDS ? s_rec[nd.leaf_begin + gl] : f.leaf_rec[nd.leaf_begin + gl];
                    const unsigned io = __float_as_uint(c.z);
                    id = io >> 4; oc = (int)(io & 15u);
                    const double dx = px - c.x;
                    double sqd = dx * dx;
                    if (!(sqd > worst)) { const double dy = py - c.y; sqd += dy * dy; }
                    hit = sqd < worst && oc >= oct_lo && oc <= predicted;
                }
                // (only lanes gl < leaf_count <= 10 can hit: the group's hit mask fits 32 bits whatever kGroup is)
                const unsigned int gm = (unsigned int)((__ballot(hit) >> (gw * (kGroup & 63))) & (kGroup >= 32 ? 0xFFFFFFFFull : ((1ull << (kGroup & 31)) - 1ull)));
                if (hit) s_cand[ncand + __popc(gm & ((1u << (gl & 31)) - 1u))] = (id << 4) | (unsigned int)oc;
                ncand += __popc(gm);
                if (ncand > kCandCap - kLeafMax) { drain(ncand); ncand = 0; }
                cur = -1;
                continue;
            }
            const double val = nd.col == 0 ? px : py;
            const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
            const bool go_left = diff1 + diff2 < 0;
            const double cut = go_left ? diff2 * diff2 : diff1 * diff1;
            const float dst = (float)(nd.col == 0 ? dd0 : dd1);
            const double m2 = cur_m + cut - dst;
            if (m2 * 1.0 <= worst) {
                if (sp + 1 > levels) { ovf = true; break; }
                const int nrec = ((go_left ? nd.right : nd.left) << 2) | ((int)nd.col << 1);
                if constexpr (LANE_STACK) { if (gl == sp) { l_rec = nrec; l_m = m2; l_c = cut; l_d = dst; } }
                else if (gl == 0) { st_rec[sp] = nrec; st_m[sp] = m2; st_c[sp] = cut; st_d[sp] = dst; }
                sp++;
            }
            cur = go_left ? nd.left : nd.right;   // best child, same mindistsq
        }
        if (clk) { clk[6] = n_iter; clk[7] = n_leaf; }
    }
    if (clk) { clk[2] = __builtin_readcyclecounter(); clk[5] = ncand; }
    if (!ovf) drain(ncand);
    if (clk) clk[3] = __builtin_readcyclecounter();
    if (ovf) { if (gl == 0) __hip_atomic_store(overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); best_kp = -1; }
    if constexpr (PREV) { if (best_kp != -1 && !(best_d < 0.7 * second_d)) best_kp = -1; }
    else if (best_kp != -1 && bestLevel2 == bestLevel && best_d > 0.8 * second_d) best_kp = -1;
    if (live && gl == 0) {   // (write-through: the publishing workgroup may sit on another XCD)
        __hip_atomic_store(mp.best_kp + m, best_kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mp.best_dist + m, (PREV && best_kp < 0) ? 3.402823466e+38f : best_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mp.visible) __hip_atomic_store(mp.visible + m, (unsigned char)(vis ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mp.pos_out) { mp.pos_out[m] = make_float4(cd.P0, cd.P1, cd.P2, 0.f); mp.aux_out[m] = mp.aux_in[m]; }
    }
    if (clk) clk[4] = __builtin_readcyclecounter();
  }
    if (!pub.ticket) return;
    // ---- hand-over: ticket behind this workgroup's acknowledged stores; the last workgroup publishes
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(pub.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    for (unsigned i = threadIdx.x; i < pub.n8; i += 4 * (unsigned)nthr) {   // four words in flight per lane
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i + u * nthr < pub.n8) v[u] = __hip_atomic_load(pub.dev_out + i + u * nthr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 4; u++) if (i + u * nthr < pub.n8) __hip_atomic_store(pub.host_out + i + u * nthr, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
        const int ov = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<unsigned*>(pub.host_done)[2] = (unsigned)ov;      // the status word travels in the header, behind the completion word
        __hip_atomic_store(overflow, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pub.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every thread: its stores into pinned memory before the word
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub.host_done, pub.word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct uh_projmatch {
    uh_ctx* ctx = nullptr;
    bool have_frame = false;
    int n_kpts = 0, n_levels = 0;
    PmFrame fr{};
    uh::DevBuf d_frame;    // kp_desc | nodes | leaf records | scale
    // pinned, device-visible staging: the frame block (set_frame), the points of a match call, its results + completion word.
    // Everything moves as 16-byte-wide launches on the context stream (uh::copy16 / publish16); the host never synchronises the stream.
    uh::MappedBuf h_frame;
    // the buffers of a match call: slot 0 serves uh_projmatch_match / _match_prev; uh_track_pose (track.hpp) enqueues its two searches
    // back to back without waiting in between and gives the second one slot 1
    struct Slot {
        uh::DevBuf d_points;          // status block | best_kp | best_dist | visible
        uh::MappedBuf h_in, h_out;    // the candidates' records (read by the kernel in place); results + completion word
        bool ovf_zeroed = false;
        unsigned ovf_gen = 0;
    } slot[2];
    unsigned long long seq = 0, frame_word = 0;   // completion words: of the last match call / of the last frame upload (posted only on demand, see upload_pending)
    // The frame upload (a copy launch reading the pinned staging block) needs no completion word of its own: every match call that follows
    // runs behind it on the same stream and the host waits for THAT call's word — which also proves the upload has been read.  Only two
    // set_frame calls with no match in between post and wait for a word (round 6: the word launch sat between the upload and the first
    // match of every frame, 4 us of dispatch on the tracker's critical path).
    bool upload_pending = false;
    std::vector<float> xy;
    std::vector<int> oct;
    std::vector<uh_dmatch> mm;
    KdBuilder kd;
    int n_nodes = 0, max_depth = 0;     // of the current frame's tree (built on the host by set_frame, on the device for set_frame_dev)
    uh_dev_frame* dev = nullptr;        // the current frame is this device-resident one (uh_projmatch_set_frame_dev)
    uh::DevBuf d_scale;                 // set_frame_dev: the scale factors (uploaded when they change)
    std::vector<float> scale_host;
    bool attr_set = false;
    struct uh_track_state* track = nullptr;   // uh_track_pose's buffers (track.hpp), created on first use
    ~uh_projmatch();
};

extern "C" {

int uh_projmatch_create(uh_ctx* ctx, uh_projmatch** out) {
    UH_REQUIRE(ctx && out, "uh_projmatch_create: NULL argument");
    uh_projmatch* h = new uh_projmatch();
    h->ctx = ctx;
    *out = h;
    return UH_OK;
}

void uh_projmatch_destroy(uh_projmatch* h) { delete h; }

// before the pinned staging block of the frame upload is written again: make sure the previous upload has been read (see upload_pending)
static int pm_staging_free(uh_projmatch* h, const char* what) {
    if (!h->upload_pending) return UH_OK;
    int rc;
    h->frame_word = ++h->seq;
    if ((rc = uh::post_host_word(h->ctx, h->h_frame.dev<unsigned long long>(), h->frame_word))) return rc;
    if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(h->h_frame.host<char>()), h->frame_word, h->ctx->stream, what))) return rc;
    h->upload_pending = false;
    return UH_OK;
}

int uh_projmatch_set_frame(uh_projmatch* h, const uh_proj_frame* f) {
    UH_REQUIRE(h && f, "uh_projmatch_set_frame: NULL argument");
    UH_REQUIRE(f->n_kpts >= 0 && (f->n_kpts == 0 || (f->und_kpts && f->desc)), "uh_projmatch_set_frame: keypoints / descriptors missing");
    UH_REQUIRE(f->n_levels >= 1 && f->scale_factors, "uh_projmatch_set_frame: scale factors missing");
    UH_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    const int n = f->n_kpts;
    std::vector<float>& xy = h->xy;
    std::vector<int>& oct = h->oct;
    xy.resize(2 * (size_t)std::max(n, 1));
    oct.resize(std::max(n, 1));
    for (int i = 0; i < n; i++) {
        // the kernel packs a candidate's octave into 4 bits and an int8: anything outside [0,16) would silently corrupt bestLevel / bestLevel2
        UH_REQUIRE(f->und_kpts[i].octave >= 0 && f->und_kpts[i].octave < 16, "uh_projmatch_set_frame: octave %d of keypoint %d outside [0,16)", f->und_kpts[i].octave, i);
        xy[2 * i] = f->und_kpts[i].x; xy[2 * i + 1] = f->und_kpts[i].y; oct[i] = f->und_kpts[i].octave;
    }
    const auto t0 = std::chrono::steady_clock::now();
    h->kd.build(xy.data(), n);
    if (getenv("UH_PM_TIMING")) fprintf(stderr, "kd build: %.1f us (n=%d, depth %d, nodes %zu)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), n, h->kd.max_depth, h->kd.nodes.size());
    UH_REQUIRE(h->kd.max_depth <= kMaxDepth, "uh_projmatch_set_frame: kd-tree depth %d exceeds the walk stack (%d levels)", h->kd.max_depth, kMaxDepth);
    const size_t nn = h->kd.nodes.size();
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_desc = 0, o_nodes = al(o_desc + 32 * (size_t)n);
    const size_t o_leaf = al(o_nodes + sizeof(KdNodeDev) * nn), o_scale = al(o_leaf + 16 * (size_t)n), total = al(o_scale + 4 * (size_t)f->n_levels);
    int rc = h->d_frame.reserve(total + 256);
    if (rc) return rc;
    char* base = h->d_frame.as<char>();
    {   // one pinned staging block, one 16-byte-wide copy launch, no synchronisation: the block is only reused by the NEXT set_frame,
        // which first makes sure this upload has landed (its completion word — long since posted unless two set_frame calls follow
        // each other with nothing in between)
        if ((rc = pm_staging_free(h, "uh_projmatch_set_frame"))) return rc;
        if ((rc = h->h_frame.reserve(total + 64))) return rc;
        char* hi = h->h_frame.host<char>() + 64;   // (the first 64 bytes hold the completion word)
        if (n) {
            std::memcpy(hi + o_desc, f->desc, 32 * (size_t)n);
            std::memcpy(hi + o_nodes, h->kd.nodes.data(), sizeof(KdNodeDev) * nn);
            float* lr = reinterpret_cast<float*>(hi + o_leaf);   // the leaf records, in leaf order
            for (int i = 0; i < n; i++) {
                const uint32_t id = h->kd.leaf_idx[i], io = (id << 4) | (uint32_t)oct[id];
                lr[4 * i] = xy[2 * (size_t)id]; lr[4 * i + 1] = xy[2 * (size_t)id + 1];
                std::memcpy(lr + 4 * i + 2, &io, 4); lr[4 * i + 3] = 0.f;
            }
        }
        std::memcpy(hi + o_scale, f->scale_factors, 4 * (size_t)f->n_levels);
        std::atomic_thread_fence(std::memory_order_release);
        if ((rc = uh::copy16(h->ctx, base, h->h_frame.dev<char>() + 64, total))) return rc;
        h->upload_pending = true;
    }
    if (getenv("UH_PM_TIMING")) fprintf(stderr, "set_frame total: %.1f us (bytes %zu)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), total);
    PmFrame& d = h->fr;
    d.kp_desc = (const uint64_t*)(base + o_desc);
    d.nodes = (const KdNodeDev*)(base + o_nodes); d.leaf_rec = (const float4*)(base + o_leaf); d.scale = (const float*)(base + o_scale);
    for (int i = 0; i < 4; i++) d.box[i] = h->kd.root_box[i];
    d.n_levels = f->n_levels; d.n_kpts = n;
    d.fx = f->fx; d.fy = f->fy; d.cx = f->cx; d.cy = f->cy;
    d.min_x = (float)f->min_x; d.min_y = (float)f->min_y; d.max_x = (float)f->max_x; d.max_y = (float)f->max_y;
    d.log_scale = f->n_levels > 1 ? std::log(f->scale_factors[1]) : 1.f;   // float overload = libm logf, as Frame::predictScale
    h->n_kpts = n; h->n_levels = f->n_levels;
    h->n_nodes = (int)nn; h->max_depth = h->kd.max_depth; h->dev = nullptr;
    h->have_frame = true;
    return UH_OK;
}

// The frame the extractor left on the device (uh_orb_extract_frame_dev: descriptors, undistorted keypoints in leaf order, kd-tree built by
// kdbuild.hip) becomes the matcher's frame: no keypoints or descriptors cross the host link, no tree is built on the host.  `f` carries the
// camera / scale / image-bounds fields only (und_kpts, n_kpts, desc are ignored).
int uh_projmatch_set_frame_dev(uh_projmatch* h, uh_dev_frame* fr, const uh_proj_frame* f) {
    UH_REQUIRE(h && fr && f, "uh_projmatch_set_frame_dev: NULL argument");
    UH_REQUIRE(fr->ctx == h->ctx, "uh_projmatch_set_frame_dev: the device frame belongs to another context");
    UH_REQUIRE(f->n_levels >= 1 && f->n_levels <= 16 && f->scale_factors, "uh_projmatch_set_frame_dev: scale factors missing");
    UH_HIP_CHECK(hipSetDevice(h->ctx->device));
    int rc;
    if (h->scale_host.size() != (size_t)f->n_levels || std::memcmp(h->scale_host.data(), f->scale_factors, 4 * (size_t)f->n_levels) != 0) {
        h->scale_host.assign(f->scale_factors, f->scale_factors + f->n_levels);
        if ((rc = h->d_scale.reserve(64))) return rc;
        UH_HIP_CHECK(hipMemcpyAsync(h->d_scale.p, h->scale_host.data(), 4 * (size_t)f->n_levels, hipMemcpyHostToDevice, h->ctx->stream));
    }
    if (fr->host_tree) {
        // the tree by this core (KdBuilder, as uh_projmatch_set_frame), nodes and leaf records uploaded into the frame object; the descriptors
        // and everything else stay where the extractor left them
        const int n = f->n_kpts;
        UH_REQUIRE(n >= 0 && n <= fr->n_cap && (n == 0 || f->und_kpts), "uh_projmatch_set_frame_dev: %d keypoints for a device frame of capacity %d (host-built tree)", n, fr->n_cap);
        std::vector<float>& xy = h->xy;
        std::vector<int>& oct = h->oct;
        xy.resize(2 * (size_t)std::max(n, 1));
        oct.resize(std::max(n, 1));
        for (int i = 0; i < n; i++) {
            UH_REQUIRE(f->und_kpts[i].octave >= 0 && f->und_kpts[i].octave < 16, "uh_projmatch_set_frame_dev: octave %d of keypoint %d outside [0,16)", f->und_kpts[i].octave, i);
            xy[2 * i] = f->und_kpts[i].x; xy[2 * i + 1] = f->und_kpts[i].y; oct[i] = f->und_kpts[i].octave;
        }
        h->kd.build(xy.data(), n);
        UH_REQUIRE(h->kd.max_depth <= kMaxDepth, "uh_projmatch_set_frame_dev: kd-tree depth %d exceeds the walk stack (%d levels)", h->kd.max_depth, kMaxDepth);
        const size_t nn = h->kd.nodes.size(), gap = fr->o_leaf - fr->o_nodes, span = gap + 16 * (size_t)n;
        UH_REQUIRE(sizeof(KdNodeDev) * nn <= gap, "uh_projmatch_set_frame_dev: %zu nodes exceed the device frame's node block", nn);
        hipStream_t st = h->ctx->stream;
        if ((rc = pm_staging_free(h, "uh_projmatch_set_frame_dev"))) return rc;
        if ((rc = h->h_frame.reserve(span + 64))) return rc;
        char* hi = h->h_frame.host<char>() + 64;   // (the first 64 bytes hold the completion word)
        if (n) {
            std::memcpy(hi, h->kd.nodes.data(), sizeof(KdNodeDev) * nn);
            float* lr = reinterpret_cast<float*>(hi + gap);   // the leaf records, in leaf order
            for (int i = 0; i < n; i++) {
                const uint32_t id = h->kd.leaf_idx[i], io = (id << 4) | (uint32_t)oct[id];
                lr[4 * i] = xy[2 * (size_t)id]; lr[4 * i + 1] = xy[2 * (size_t)id + 1];
                std::memcpy(lr + 4 * i + 2, &io, 4); lr[4 * i + 3] = 0.f;
            }
            std::atomic_thread_fence(std::memory_order_release);
            // (one launch over the node block and the leaf records: the unused tail of the node block travels with them — a few KB against a second launch)
            if ((rc = uh::copy16(h->ctx, reinterpret_cast<char*>(fr->nodes()), h->h_frame.dev<char>() + 64, span))) return rc;
            h->upload_pending = true;
        }
        PmFrame& d = h->fr;
        d.kp_desc = reinterpret_cast<const uint64_t*>(fr->desc());
        d.nodes = reinterpret_cast<const KdNodeDev*>(fr->nodes()); d.leaf_rec = fr->leaf(); d.scale = h->d_scale.as<float>();
        for (int i = 0; i < 4; i++) d.box[i] = h->kd.root_box[i];
        d.n_levels = f->n_levels; d.n_kpts = n;
        d.fx = f->fx; d.fy = f->fy; d.cx = f->cx; d.cy = f->cy;
        d.min_x = (float)f->min_x; d.min_y = (float)f->min_y; d.max_x = (float)f->max_x; d.max_y = (float)f->max_y;
        d.log_scale = f->n_levels > 1 ? std::log(f->scale_factors[1]) : 1.f;
        h->n_kpts = n; h->n_levels = f->n_levels;
        h->n_nodes = (int)nn; h->max_depth = h->kd.max_depth; h->dev = fr;
        h->have_frame = true;
        return UH_OK;
    }
    const uh_kd::Meta* m = nullptr;
    if ((rc = uh::dev_frame_wait(fr, &m, "uh_projmatch_set_frame_dev"))) return rc;   // (the build runs behind the extractor's completion word: normally done)
    UH_REQUIRE(m->max_depth <= kMaxDepth, "uh_projmatch_set_frame_dev: kd-tree depth %d exceeds the walk stack (%d levels)", m->max_depth, kMaxDepth);
    PmFrame& d = h->fr;
    d.kp_desc = reinterpret_cast<const uint64_t*>(fr->desc());
    d.nodes = reinterpret_cast<const KdNodeDev*>(fr->nodes()); d.leaf_rec = fr->leaf(); d.scale = h->d_scale.as<float>();
    for (int i = 0; i < 4; i++) d.box[i] = m->box[i];
    d.n_levels = f->n_levels; d.n_kpts = m->n;
    d.fx = f->fx; d.fy = f->fy; d.cx = f->cx; d.cy = f->cy;
    d.min_x = (float)f->min_x; d.min_y = (float)f->min_y; d.max_x = (float)f->max_x; d.max_y = (float)f->max_y;
    d.log_scale = f->n_levels > 1 ? std::log(f->scale_factors[1]) : 1.f;
    h->n_kpts = m->n; h->n_levels = f->n_levels;
    h->n_nodes = m->n_nodes; h->max_depth = m->max_depth; h->dev = fr;
    h->have_frame = true;
    return UH_OK;
}

// host-only test hook (no device call): the tree set_frame would build for these points.  nodes24_out: 2n+2 nodes of room.
int uh_kdtree_build_host(const float* xy, int32_t n, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out, double* root_box4, int32_t* max_depth) {
    UH_REQUIRE(n >= 0 && (n == 0 || xy) && n_nodes && nodes24_out && leaf_idx_out && root_box4, "uh_kdtree_build_host: bad arguments");
    KdBuilder kd;
    kd.build(xy, n);
    *n_nodes = (int32_t)kd.nodes.size();
    if (!kd.nodes.empty()) std::memcpy(nodes24_out, kd.nodes.data(), sizeof(KdNodeDev) * kd.nodes.size());
    if (n) std::memcpy(leaf_idx_out, kd.leaf_idx.data(), 4 * (size_t)n);
    std::memcpy(root_box4, kd.root_box, 32);
    if (max_depth) *max_depth = kd.max_depth;
    return UH_OK;
}

// flattened tree of the current frame (for tests: compared with the oracle's / the real picoflann's build)
int uh_projmatch_debug_tree(uh_projmatch* h, int32_t* n_nodes, const void** nodes24, const uint32_t** leaf_idx, double* root_box4, int32_t* max_depth) {
    UH_REQUIRE(h && h->have_frame, "uh_projmatch_debug_tree: no frame set");
    UH_REQUIRE(!h->dev, "uh_projmatch_debug_tree: the current frame is device-resident (use uh_dev_frame_tree)");
    if (n_nodes) *n_nodes = (int32_t)h->kd.nodes.size();
    if (nodes24) *nodes24 = h->kd.nodes.data();
    if (leaf_idx) *leaf_idx = h->kd.leaf_idx.data();
    if (root_box4) std::memcpy(root_box4, h->kd.root_box, 32);
    if (max_depth) *max_depth = h->kd.max_depth;
    return UH_OK;
}

}  // extern "C"

namespace {

// What a search leaves behind until its results are collected: where they lie (HBM, pinned twin) and the call's completion word.
struct PmPending {
    int n = 0, slot = 0;
    bool prev = false;
    unsigned long long word = 0;
    const int* d_best_kp = nullptr; const float* d_best_dist = nullptr;   // HBM (for launches behind this one on the same stream)
    const float* d_rec = nullptr;                                        // the candidates' 64-byte records (pinned, device address)
    size_t o_bk = 0, o_bd = 0, o_vis = 0;
};

// Stage the candidates and enqueue the search (no waiting).  octave == nullptr: Map::matchFrameToMapPoints; normal / min / max == nullptr: the
// previous-frame search.  dyn != nullptr: pose and radius are read from device memory at launch time (written by an earlier launch of this stream).
int match_enqueue(uh_projmatch* h, int slot, const float* pose_f2g, const PmDyn* dyn, int n, const float* pos3d, const float* normal,
                  const float* mn_dist, const float* mx_dist, const uint8_t* desc, const int32_t* octave, float min_desc_dist, float max_repj_dist, PmPending* pend,
                  float4* pos_out = nullptr, const uint32_t* ids = nullptr, const int32_t* rows = nullptr, const float* weights = nullptr, const float* weights_by_row = nullptr,
                  uint4* aux_out = nullptr) {
    const bool prev = octave != nullptr;
    uh_projmatch::Slot& S = h->slot[slot];
    UH_HIP_CHECK(hipSetDevice(h->ctx->device));
    hipStream_t st = h->ctx->stream;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_pos = 256 /* [0, 256): the walk-overflow status word, at a fixed place */;
    const size_t o_bk = 256, o_bd = o_bk + 4 * (size_t)n, o_vis = o_bd + 4 * (size_t)n;
    const size_t o_ovf = 0, o_end = al(o_vis + (size_t)n), total = o_end + 256;
    int rc = S.d_points.reserve(total);
    if (rc) return rc;
    const size_t out_bytes = o_end - o_bk;
    if ((rc = S.h_out.reserve(out_bytes + 64))) return rc;
    char* base = S.d_points.as<char>();
    if (!S.ovf_zeroed || S.ovf_gen != S.d_points.gen) {   // the walk-overflow word: zero once per allocation, the kernel's last workgroup clears it after every call
        UH_HIP_CHECK(hipMemsetAsync(base, 0, 256, st));
        S.ovf_zeroed = true; S.ovf_gen = S.d_points.gen;
    }
    static const bool pm_clk = getenv("UH_PM_CLK") != nullptr;
    if (pm_clk) { const unsigned magic[2] = {0u, 0x434c4bu}; UH_HIP_CHECK(hipMemcpyAsync(base, magic, 8, hipMemcpyHostToDevice, st)); }
    static const bool pm_timing = getenv("UH_PM_TIMING") != nullptr;
    const auto t_pack0 = std::chrono::steady_clock::now();
    {   // one pinned staging block (the previous call's launches are complete: its results were awaited): the candidates as 64-byte records
        if ((rc = S.h_in.reserve(o_pos + 80 * (size_t)n + 64))) return rc;   // (64-byte records, then uh_track_pose's 16-byte ones)
        char* hi = S.h_in.host<char>();
        if (pos_out) {   // uh_track_pose: {id, row of the same point in the local map (a map candidate: its own), the solver weight of that row}
            uint32_t* ax = reinterpret_cast<uint32_t*>(hi + o_pos + 64 * (size_t)n);
            for (int i = 0; i < n; i++, ax += 4) {
                const int row = prev ? (rows ? rows[i] : -1) : i;
                const float w = prev ? (row >= 0 && weights_by_row ? weights_by_row[row] : 1.f) : (weights ? weights[i] : 1.f);
                ax[0] = ids[i]; std::memcpy(ax + 1, &row, 4); std::memcpy(ax + 2, &w, 4); ax[3] = 0;
            }
        }
        float* rec = reinterpret_cast<float*>(hi + o_pos);
        for (int i = 0; i < n; i++, rec += 16) {
            rec[0] = pos3d[3 * i]; rec[1] = pos3d[3 * i + 1]; rec[2] = pos3d[3 * i + 2];
            if (prev) { std::memcpy(rec + 3, octave + i, 4); rec[4] = rec[5] = rec[6] = rec[7] = 0.f; }
            else { rec[3] = normal[3 * i]; rec[4] = normal[3 * i + 1]; rec[5] = normal[3 * i + 2]; rec[6] = mn_dist[i]; rec[7] = mx_dist[i]; }
            std::memcpy(rec + 8, desc + 32 * (size_t)i, 32);
        }
        std::atomic_thread_fence(std::memory_order_release);
    }
    if (pm_timing) fprintf(stderr, "projmatch%s: packing %d records %.1f us\n", prev ? "_prev" : "", n, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_pack0).count());
    PmPoints P;
    P.n = n;
    P.rec = reinterpret_cast<const uint4*>(S.h_in.dev<char>() + o_pos);   // (read by the kernel where they lie: every wave fetches its own candidate once)
    P.best_kp = (int*)(base + o_bk); P.best_dist = (float*)(base + o_bd); P.visible = (unsigned char*)(base + o_vis);
    P.pos_out = pos_out;
    P.aux_in = reinterpret_cast<const uint4*>(S.h_in.dev<char>() + o_pos + 64 * (size_t)n); P.aux_out = aux_out;
    PmPose ps{};
    if (pose_f2g) {
        const float* T = pose_f2g;
        for (int i = 0; i < 12; i++) ps.T[i] = T[i];
        // camCenter = pose_f2g.inv() * (0,0,0), se3transform.h:89-113
        const float m0 = T[0], m1 = T[4], m2 = T[8], m4 = T[1], m5 = T[5], m6 = T[9], m8 = T[2], m9 = T[6], m10 = T[10];
        const float m3 = -(T[3] * m0 + T[7] * m1 + T[11] * m2), m7 = -(T[3] * m4 + T[7] * m5 + T[11] * m6), m11 = -(T[3] * m8 + T[7] * m9 + T[11] * m10);
        ps.cc[0] = m0 * 0.f + m1 * 0.f + m2 * 0.f + m3;
        ps.cc[1] = m4 * 0.f + m5 * 0.f + m6 * 0.f + m7;
        ps.cc[2] = m8 * 0.f + m9 * 0.f + m10 * 0.f + m11;
    }
    {
        const int n_nodes = h->n_nodes, levels = h->max_depth + 2;
        // Points per workgroup (= waves: one point per wave).  The walk is issue-bound — ~70 instructions per tree step and point, whatever shares
        // the SIMD — so the points are spread over as many compute units as there are: 800 previous-frame items ran on 50 CUs at 16 per
        // workgroup (match_prev 58 us); every workgroup stages the frame itself (~80 KB from L2: 2-3 us), so not below 4.
        const int ncu = std::max(h->ctx->num_cus, 64);
        const int gpw = std::min(kGroupsMax, std::max(4, uh_div_up(n, ncu)));
        const size_t stack_bytes = (size_t)levels * gpw * 24 + (size_t)kCandCap * gpw * 8 + 64;
        const size_t tree_bytes = (((size_t)n_nodes * sizeof(KdNodeDev) + 15) & ~(size_t)15) + 16 * (size_t)h->n_kpts;
        static const bool no_lds = getenv("UH_PROJMATCH_NO_LDS") != nullptr;   // env: test knob for the big-frame path
        const bool in_lds = tree_bytes + stack_bytes <= kLdsBudget && !no_lds;
        const size_t lds = stack_bytes + (in_lds ? tree_bytes : 0);
        if (!h->attr_set) {
#define UH_PM_ATTR(A, B, C) UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(projmatch_kernel<A, B, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget))
            UH_PM_ATTR(true, false, true); UH_PM_ATTR(false, false, true); UH_PM_ATTR(true, true, true); UH_PM_ATTR(false, true, true);
            UH_PM_ATTR(true, false, false); UH_PM_ATTR(false, false, false); UH_PM_ATTR(true, true, false); UH_PM_ATTR(false, true, false);
#undef UH_PM_ATTR
            h->attr_set = true;
        }
        const dim3 grid(std::min(uh_div_up(n, gpw), kPmMaxBlocks));
        int* d_ovf = (int*)(base + o_ovf);
        const unsigned long long word = ++h->seq;
        const PmPublish pub{reinterpret_cast<unsigned*>(base + 128), reinterpret_cast<const unsigned long long*>(base + o_bk), reinterpret_cast<unsigned long long*>(S.h_out.dev<char>() + 64),
                            (unsigned)(out_bytes / 8), S.h_out.dev<unsigned long long>(), word};
        const bool lane_stack = levels <= kGroup;
#define UH_PM_LAUNCH(A, B, C) UH_LAUNCH(h->ctx, (projmatch_kernel<A, B, C>), grid, dim3(gpw * kGroup), lds, h->fr, P, ps, min_desc_dist, max_repj_dist, n_nodes, levels, d_ovf, pub, dyn)
        if (lane_stack) {
            if (in_lds && !prev) UH_PM_LAUNCH(true, false, true); else if (!prev) UH_PM_LAUNCH(false, false, true);
            else if (in_lds) UH_PM_LAUNCH(true, true, true); else UH_PM_LAUNCH(false, true, true);
        } else {
            if (in_lds && !prev) UH_PM_LAUNCH(true, false, false); else if (!prev) UH_PM_LAUNCH(false, false, false);
            else if (in_lds) UH_PM_LAUNCH(true, true, false); else UH_PM_LAUNCH(false, true, false);
        }
#undef UH_PM_LAUNCH
        pend->word = word;
    }
    UH_HIP_CHECK(hipGetLastError());
    pend->n = n; pend->slot = slot; pend->prev = prev;
    pend->d_best_kp = P.best_kp; pend->d_best_dist = P.best_dist; pend->d_rec = reinterpret_cast<const float*>(P.rec);
    pend->o_bk = o_bk; pend->o_bd = o_bd; pend->o_vis = o_vis;
    return UH_OK;
}

// Wait for a search and collect its per-candidate results (pinned block -> the caller's arrays, any may be NULL).
int match_collect(uh_projmatch* h, const PmPending& pd, const char* what, const int** bk_out, const float** bd_out, int32_t* best_kp_out, float* best_dist_out, uint8_t* visible_out) {
    uh_projmatch::Slot& S = h->slot[pd.slot];
    int rc;
    // results: the kernel's last workgroup copies [best_kp | best_dist | visible] and the overflow flag into pinned memory and posts the completion word
    if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(S.h_out.host<char>()), pd.word, h->ctx->stream, what))) {
        S.ovf_zeroed = false;   //  a launch that died may have left its ticket / overflow word behind: the next call clears the block again
        return rc;
    }
    h->upload_pending = false;   // (this call ran behind the frame upload on the same stream)
    static const bool pm_clk = getenv("UH_PM_CLK") != nullptr;
    if (pm_clk) {
        long long c[8];
        UH_HIP_CHECK(hipMemcpy(c, S.d_points.as<char>() + 16, sizeof(c), hipMemcpyDeviceToHost));
        fprintf(stderr, "projmatch%s workgroup 0 cycles: stage %lld  visibility+walk %lld (group 0: %lld loop iterations, %lld leaves)  final drain (%lld hits) %lld  tail %lld  total %lld\n",
                pd.prev ? "_prev" : "", c[1] - c[0], c[2] - c[1], c[6], c[7], c[5], c[3] - c[2], c[4] - c[3], c[4] - c[0]);
    }
    const char* ho = S.h_out.host<char>() + 64;
    const int n = pd.n;
    const int* bk = (const int*)ho;
    const float* bd = (const float*)(ho + (pd.o_bd - pd.o_bk));
    const unsigned char* vis = (const unsigned char*)(ho + (pd.o_vis - pd.o_bk));
    const int ovf = *reinterpret_cast<const int*>(S.h_out.host<char>() + 8);
    UH_REQUIRE(!ovf, "%s: kd-tree walk stack overflow", what);
    if (best_kp_out) std::memcpy(best_kp_out, bk, 4 * (size_t)n);
    if (best_dist_out) std::memcpy(best_dist_out, bd, 4 * (size_t)n);
    if (visible_out) std::memcpy(visible_out, vis, (size_t)n);
    if (bk_out) *bk_out = bk;
    if (bd_out) *bd_out = bd;
    return UH_OK;
}

// one implementation behind uh_projmatch_match (octave == nullptr) and uh_projmatch_match_prev (normal/min/max == nullptr)
int match_common(uh_projmatch* h, const float* pose_f2g, int n, const uint32_t* ids, const float* pos3d, const float* normal,
                 const float* mn_dist, const float* mx_dist, const uint8_t* desc, const int32_t* octave, float min_desc_dist,
                 float max_repj_dist, uh_dmatch* matches_out, int32_t cap, int32_t* best_kp_out, float* best_dist_out, uint8_t* visible_out) {
    PmPending pd;
    int rc = match_enqueue(h, 0, pose_f2g, nullptr, n, pos3d, normal, mn_dist, mx_dist, desc, octave, min_desc_dist, max_repj_dist, &pd);
    if (rc) return rc;
    const int* bk = nullptr; const float* bd = nullptr;
    if ((rc = match_collect(h, pd, "uh_projmatch_match", &bk, &bd, best_kp_out, best_dist_out, visible_out))) return rc;
    std::vector<uh_dmatch>& mm = h->mm;
    mm.clear();
    mm.reserve(n);
    for (int i = 0; i < n; i++)
        if (bk[i] >= 0) mm.push_back(uh_dmatch{bk[i], (int32_t)ids[i], -1, bd[i]});
    int k = mm.empty() ? 0 : uh_filter_ambiguous(mm.data(), (int)mm.size(), 0);
    if (k < 0) return k;
    UH_REQUIRE(k <= cap, "uh_projmatch_match: %d matches do not fit the output buffer (cap %d)", k, (int)cap);
    if (k) std::memcpy(matches_out, mm.data(), sizeof(uh_dmatch) * (size_t)k);
    return k;
}

}  // namespace

extern "C" {

int uh_projmatch_match(uh_projmatch* h, const float* pose_f2g, const uh_map_points* mp, float min_desc_dist, float max_repj_dist,
                       uh_dmatch* matches_out, int32_t cap, int32_t* best_kp_out, float* best_dist_out, uint8_t* visible_out) {
    UH_REQUIRE(h && h->have_frame, "uh_projmatch_match: no frame set (call uh_projmatch_set_frame first)");
    UH_REQUIRE(pose_f2g && mp && mp->n >= 0, "uh_projmatch_match: NULL / negative argument");
    UH_REQUIRE(max_repj_dist > 0, "uh_projmatch_match: maxRepjDist must be > 0 (a non-positive radius turns the reference's search into an unbounded one)");
    if (mp->n == 0) return 0;
    UH_REQUIRE(mp->ids && mp->pos3d && mp->normal && mp->min_dist && mp->max_dist && mp->desc, "uh_projmatch_match: map point arrays missing");
    UH_REQUIRE(matches_out && cap >= 0, "uh_projmatch_match: output buffer missing");
    return match_common(h, pose_f2g, mp->n, mp->ids, mp->pos3d, mp->normal, mp->min_dist, mp->max_dist, mp->desc, nullptr, min_desc_dist,
                        max_repj_dist, matches_out, cap, best_kp_out, best_dist_out, visible_out);
}

int uh_projmatch_match_prev(uh_projmatch* h, const float* pose_f2g, const uh_prev_points* pp, float min_desc_dist, float max_repj_dist,
                            uh_dmatch* matches_out, int32_t cap, int32_t* best_kp_out, float* best_dist_out) {
    UH_REQUIRE(h && h->have_frame, "uh_projmatch_match_prev: no frame set (call uh_projmatch_set_frame first)");
    UH_REQUIRE(pose_f2g && pp && pp->n >= 0, "uh_projmatch_match_prev: NULL / negative argument");
    UH_REQUIRE(max_repj_dist > 0, "uh_projmatch_match_prev: maxRepjDist must be > 0 (a non-positive radius turns the reference's search into an unbounded one)");
    if (pp->n == 0) return 0;
    UH_REQUIRE(pp->ids && pp->pos3d && pp->octave && pp->desc, "uh_projmatch_match_prev: point arrays missing");
    UH_REQUIRE(matches_out && cap >= 0, "uh_projmatch_match_prev: output buffer missing");
    for (int i = 0; i < pp->n; i++)   // Frame::scaleFactors[octave] (system.cpp:6130): out of range is undefined in the reference
        UH_REQUIRE(pp->octave[i] >= 0 && pp->octave[i] < h->n_levels, "uh_projmatch_match_prev: octave %d of item %d outside [0,%d)", pp->octave[i], i, h->n_levels);
    return match_common(h, pose_f2g, pp->n, pp->ids, pp->pos3d, nullptr, nullptr, nullptr, pp->desc, pp->octave, min_desc_dist,
                        max_repj_dist, matches_out, cap, best_kp_out, best_dist_out, nullptr);
}

}  // extern "C"

#include "track.hpp"
