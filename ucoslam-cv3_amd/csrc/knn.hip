// Hamming brute-force kNN over 256-bit ORB descriptors, bit-exact with xflann::Index(Linear).
//
// Reference contract (lambdaloop/ucoslam-cv3):
//   3rdparty/xflann/xflann/impl/linear.h:68-86   for each query, scan train rows in index order and
//                                                 push (dist,i) into a ResultSet; unfilled -> (-1, 0)
//   3rdparty/xflann/xflann/impl/resultset.h:64-135  ResultSet = binary max-heap stored in the output row
//   3rdparty/xflann/xflann/index.h:119-134        optional in-place exchange sort of each row
//   3rdparty/xflann/xflann/impl/distances.h:279   distance = 4 x popcount64(xor)
//
// MI355X design: one 64-lane wave owns one query.  Lanes hold 64 CONSECUTIVE train rows per
// step (two coalesced 16-byte loads per lane = 2 KiB per wave-instruction pair), the query lives in
// SGPR-uniform registers, the distance is 8 x (v_xor + v_bcnt_u32 with fused accumulate).  A push
// only happens when d < heap root, so one compare + one 64-bit ballot rejects a whole step; the
// few accepted pushes (~k(1+ln(N/k)) per query) replay the reference's sequential heap exactly,
// in lane (= train index) order, on a heap that is DISTRIBUTED OVER LANES (lane j = heap slot j)
// and manipulated with v_readlane / masked moves -- no LDS, no scratch, wave-uniform control flow.
//
// Large query sets (>= 6000 queries, k <= 16) take the two-phase form further down: an accept-list scan with two queries per wave and
// a replay with one lane per query on register heaps ("two-phase exact search").
//
// Sharding (multi-GPU, row (e) of the scope table): a shard scan starts from an empty heap, so
// its accept test is looser than the global one; it emits everything it accepted (a superset of
// the global accept set, in index order).  Replaying the concatenated shard lists through the
// same push routine reproduces the global heap bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <cstring>
#include <random>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxShards = 64;

struct ShardBounds { int n; int b[kMaxShards + 1]; };

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// compile-time loop: every index inside f is a constant (register arrays stay in registers)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Lane-distributed ResultSet: lane j holds heap slot j.  A push is BRANCH-FREE VECTOR CODE with a short dependent chain
// of cross-lane operations (the latency of those, not ALU work, is what a push costs):
//  * removing the root ("up", resultset.h:104-135): every slot picks its bigger child with the reference's tie rule
//    (dr < dl ? left : right) [1st cross-lane round]; a slot lies on the descent path iff each of its ancestors chose the
//    next node of the chain — all ancestors' choices are fetched at once [2nd round, together with the child's index];
//    the moving element (old last) passes every path slot whose value is > it, each such slot takes its bigger child's
//    entry (or the moving element where the descent stops).  Heap order makes "value > moving" monotone along the path.
//  * appending ("down", :93-100): the ancestors of the new slot whose value is < the new distance take their parent's
//    entry (or the new element where the climb stops) [3rd round].
// The result equals the sequential swaps of the reference (tests compare the unsorted heap rows bit for bit).
struct WaveHeap {
    int hd;    // per lane: distance of slot `lane`
    int hi;    // per lane: index of slot `lane`
    int size;  // wave-uniform
    int lane;

    __device__ __forceinline__ void swap(int a, int b) {   // only used by the final exchange sort
        const int da = rl(hd, a), db = rl(hd, b), ia = rl(hi, a), ib = rl(hi, b);
        const bool ma = lane == a, mb = lane == b;
        hd = ma ? db : (mb ? da : hd);
        hi = ma ? ib : (mb ? ia : hi);
    }
    __device__ __forceinline__ bool accepts(int d, int k, int maxd) const {   // resultset.h:66-69
        if (maxd >= 0 && maxd < d) return false;
        if (size >= k) return d < rl(hd, 0);
        return true;
    }
    // resultset.h:64-82, caller has already established accepts(d).  LV = depth of the deepest heap slot (floor(log2(k)), 6 covers
    // k <= 64): the ancestor walk of the root removal is that many cross-lane rounds.
    template <int LV = 6>
    __device__ __forceinline__ void push_accepted(int d, int idx, int k) {
        const int parent = lane > 0 ? (lane - 1) >> 1 : 0;
        if (size >= k) {
            const int last = size - 1;
            const int md = rl(hd, last), mi = rl(hi, last);   // the element that re-enters at the root
            size = last;
            if (size >= 1) {
                const int l = 2 * lane + 1, r = l + 1;
                const int vl = __shfl(hd, l), vr = __shfl(hd, r);
                const bool hasL = l < size, hasR = r < size;
                const bool pickL = !hasR || (vr < vl);
                const int c = pickL ? l : r;
                const int cv = pickL ? vl : vr;
                const int ci = __shfl(hi, c);
                bool onpath = lane < size;
                int node = lane;
#pragma unroll
                for (int t = 0; t < LV; t++) {
                    const int par = node > 0 ? (node - 1) >> 1 : 0;
                    const int cpar = __shfl(c, par);          // independent of the other rounds: all in flight together
                    onpath = onpath && (node == 0 || cpar == node);
                    node = par;
                }
                const bool reached = onpath && (lane == 0 || hd > md);
                const bool takeChild = hasL && cv > md;
                hd = reached ? (takeChild ? cv : md) : hd;
                hi = reached ? (takeChild ? ci : mi) : hi;
            }
        }
        const int s = size;
        bool onanc = false;
        for (int x = s + 1; x > 0; x >>= 1) onanc = onanc || (x - 1 == lane);   // s and its ancestors
        const int pv = __shfl(hd, parent), pi = __shfl(hi, parent);
        const bool mine = onanc && (lane == s || hd < d);
        const bool takeParent = lane > 0 && pv < d;
        hd = mine ? (takeParent ? pv : d) : hd;
        hi = mine ? (takeParent ? pi : idx) : hi;
        size = s + 1;
    }
    // Alternative formulation: the same sifts walked by wave-uniform SCALAR code (v_readlane + s_cselect), predicated
    // per level instead of branching, slot updates as v_cndmask.  No LDS-crossbar (ds_bpermute) latency on the path.
    __device__ __forceinline__ void put_if(bool on, int slot, int d, int i) {
        const bool m = on && lane == slot;
        hd = m ? d : hd;
        hi = m ? i : hi;
    }
    __device__ __forceinline__ void push_accepted_scalar(int d, int idx, int k) {
        if (size >= k) {
            const int last = size - 1;
            const int md = rl(hd, last), mi = rl(hi, last);
            size = last;
            if (size >= 1) {
                int pos = 0;
                bool go = size > 1;
#pragma unroll
                for (int t = 0; t < 6; t++) {
                    const int l = 2 * pos + 1, r = l + 1;
                    const bool hasL = go && l < size, hasR = go && r < size;
                    const int dl = rl(hd, hasL ? l : 0), dr = rl(hd, hasR ? r : 0);
                    const bool pickL = !hasR || dr < dl;
                    const int c = pickL ? l : r, dc = pickL ? dl : dr;
                    const bool mv = hasL && md < dc;
                    const int ic = rl(hi, mv ? c : 0);
                    put_if(mv, pos, dc, ic);
                    pos = mv ? c : pos;
                    go = mv;
                }
                put_if(true, pos, md, mi);
            }
        }
        int pos = size;
#pragma unroll
        for (int t = 0; t < 6; t++) {
            const int parent = pos > 0 ? (pos - 1) >> 1 : 0;
            const int dp = rl(hd, parent), ip = rl(hi, parent);
            const bool mv = pos > 0 && dp < d;
            put_if(mv, pos, dp, ip);
            pos = mv ? parent : pos;
        }
        put_if(true, pos, d, idx);
        size++;
    }
    __device__ __forceinline__ int threshold(int k) const { return size >= k ? rl(hd, 0) : 0x7fffffff; }
};

// v_bcnt_u32_b32 adds its second operand: a popcount that accumulates costs one instruction.  Written as asm because the
// compiler otherwise splits the chain into zero-based counts plus v_add3 (19 instead of 17 VALU ops per 256-bit distance) — the scan
// is bound by VALU issue, not by the latency of the chain.
__device__ __forceinline__ int bcnt_acc(uint32_t x, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint32_t (&q)[8]) {
    int d0 = __popc(a0.x ^ q[0]), d1 = __popc(a1.x ^ q[4]);
    d0 = bcnt_acc(a0.y ^ q[1], d0); d1 = bcnt_acc(a1.y ^ q[5], d1);
    d0 = bcnt_acc(a0.z ^ q[2], d0); d1 = bcnt_acc(a1.z ^ q[6], d1);
    d0 = bcnt_acc(a0.w ^ q[3], d0); d1 = bcnt_acc(a1.w ^ q[7], d1);
    return d0 + d1;
}

// Feed one step (64 candidates, one per lane, ascending index with lane) into the heap.
template <bool EMIT, int LV = 6>
__device__ __forceinline__ void feed_step(WaveHeap& h, int d, int idx, bool valid, int k, int maxd,
                                          uint64_t* cand_row, int& ncand, int cap) {
    int thr = h.threshold(k);
    bool pass = valid && (maxd < 0 || d <= maxd) && d < thr;
    uint64_t m = __ballot(pass);
    while (m) {
        int l = __builtin_ctzll(m);
        m &= m - 1;
        int dl = rl(d, l);
        if (!h.accepts(dl, k, maxd)) continue;
        int il = rl(idx, l);
        if (EMIT) {
            if (ncand < cap && h.lane == 0) cand_row[ncand] = ((uint64_t)(uint32_t)dl << 32) | (uint32_t)il;
            ncand++;
        }
        h.template push_accepted<LV>(dl, il, k);
    }
}

template <bool EMIT, int LV = 6>
__device__ __forceinline__ void scan_range(WaveHeap& h, const uint8_t* __restrict__ train, int t0, int t1,
                                           const uint32_t (&q)[8], int k, int maxd,
                                           uint64_t* cand_row, int& ncand, int cap) {
    const int lane = h.lane;
    constexpr int UNROLL = 4;
    int base = t0;
    // buffer loads: the per-lane part of the address (lane * 32 + a constant) never changes, the part that moves is wave-uniform
    // and lives in an SGPR — the group loop spends no VALU instruction on addresses (the flat form needs four 64-bit ones per
    // 64 rows).  0x00020000: raw 32-bit-format buffer descriptor word of gfx9-family parts.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(train), 0, t1 * 32, 0x00020000);
    const int voff = lane * 32;
    for (; base + UNROLL * kWave <= t1; base += UNROLL * kWave) {
        uint4 a0[UNROLL], a1[UNROLL];
        const int soff = __builtin_amdgcn_readfirstlane(base * 32);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32, soff, 0);
            const u32x4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32 + 16, soff, 0);
            a0[u] = make_uint4(x0.x, x0.y, x0.z, x0.w);
            a1[u] = make_uint4(x1.x, x1.y, x1.z, x1.w);
        }
        // one scalar decision per UNROLL steps: almost every group has no row below the current threshold, and each
        // VALU -> ballot -> branch round trip costs more than the 16 VALU ops of a distance
        int d[UNROLL];
        bool any = false;
        const int thr = h.threshold(k);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { d[u] = hamming256(a0[u], a1[u], q); any = any || d[u] < thr; }
        if (__ballot(any) == 0) continue;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) feed_step<EMIT, LV>(h, d[u], base + u * kWave + lane, true, k, maxd, cand_row, ncand, cap);
    }
    for (; base < t1; base += kWave) {
        int t = base + lane;
        bool valid = t < t1;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (valid) {
            const uint4* p = reinterpret_cast<const uint4*>(train + (size_t)t * 32);
            a0 = p[0];
            a1 = p[1];
        }
        int d = hamming256(a0, a1, q);
        feed_step<EMIT, LV>(h, d, t, valid, k, maxd, cand_row, ncand, cap);
    }
}

__device__ __forceinline__ void load_query(const uint8_t* __restrict__ queries, int qi, uint32_t (&q)[8]) {
    const uint32_t* qp = reinterpret_cast<const uint32_t*>(queries + (size_t)qi * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_readfirstlane(qp[j]);
}

// linear.h:82-85 (fill) + index.h:119-134 (exchange sort, including its treatment of unfilled slots)
__device__ __forceinline__ void finish_row(WaveHeap& h, int k, int sorted, int32_t* __restrict__ indices,
                                           int32_t* __restrict__ distances, int qi) {
    if (h.lane >= h.size) { h.hd = 0; h.hi = -1; }
    if (sorted) {
        for (int i = 0; i < k - 1; ++i) {
            if (rl(h.hi, i) == -1) continue;
            for (int j = i + 1; j < k; ++j) {
                if (rl(h.hd, i) > rl(h.hd, j)) h.swap(i, j);
            }
        }
    }
    if (h.lane < k) {
        indices[(size_t)qi * k + h.lane] = h.hi;
        distances[(size_t)qi * k + h.lane] = h.hd;
    }
}

// One wave per query; whole train range; writes final rows.  LV: see WaveHeap::push_accepted (1: k <= 3, 3: k <= 15, 6: k <= 64).
template <int LV>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_search_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k,
    int sorted, int maxd, int32_t* __restrict__ indices, int32_t* __restrict__ distances) {
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (qi >= nq) return;
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    int ncand = 0;
    scan_range<false, LV>(h, train, t0, t1, q, k, maxd, nullptr, ncand, 0);
    finish_row(h, k, sorted, indices, distances, qi);
}

// ------------------------------------------------------------------------------------------------ two-phase exact search (nn <= 16)
// The fused kernel above replays every accepted push on a lane-distributed heap inside the scanning wave: ~1000 cycles of dependent
// ds_bpermute rounds per push, ~70 pushes per query (nn = 10, 10 000 rows) — more than the scan itself.  The two-phase form splits it:
//
//  Phase 1 (knn_accept_kernel) only needs the THRESHOLD the reference's heap would have — its root when full = the k-th smallest
//  distance seen so far.  The smallest distances are kept SORTED along the lanes of a 16-lane DPP row: an accepted distance d is inserted
//  with one row_shr:1 + compare + max + select, the threshold is lane k-1 (one v_readlane).  Rows below the threshold at the start of
//  a 64-row step are appended to the query's accept list by all passing lanes at once (a superset of what the reference's heap accepts,
//  in index order).  Two queries per wave share every loaded row (one query per wave is L1-bandwidth bound: 76 us of scan against 50 us,
//  MI355X, 8000 x 10 000).  History of the k-set: unordered in a DPP row, replace-the-maximum + 4-step DPP max + readfirstlane + ballot =
//  25 vector instructions per accepted row: 88 us (38 us of it maintenance); a sorted list in SGPRs updated by 2 k scalar min / max: 100 us
//  (the CU's one scalar unit issues for all four SIMDs, so a scalar instruction costs as much CU time as a vector one).
//
//  Phase 2 (knn_replay_lane_kernel) replays the lists through the reference's ResultSet (resultset.h:64-135) with ONE LANE PER QUERY and the
//  heap in REGISTERS, K a template parameter: every heap index is a compile-time constant (the sift-down selects among the nodes of one
//  level, the sift-up walks the fixed ancestor chain of slot K-1), no LDS, no cross-lane traffic; 64 queries advance per instruction.
//  History: heaps in LDS, one lane per query: 174 us (a ~95-element chain of LDS round trips per wave); one WAVE per query on the
//  lane-distributed heap: 107 us; four queries per wave in 16-lane rows with row-local ds_bpermute: 63 us.
//
//  Lists that overflow their capacity (distances descending with the row index) are collected and recomputed by knn_redo_kernel.
// STREAM (knn_stream_kernel): the lists are consumed WHILE they grow, by replay waves of the same launch on other CUs.  An entry is then
// launch tag << 32 | distance << 23 | row (a reader validates the tag: no flags, no ordering between words, nothing to clear between
// launches) and entries travel in PAIRS: record r of query q = the 16 bytes at (r * nq + q) * 16, two tagged 8-byte halves written by ONE
// 16-byte write-through store (the odd entry waits in a register for its partner; the last record of a list is flushed with an empty
// second half).  No progress word accompanies the entries — a reader simply tries the next records and takes those whose two tags
// match; prog[q] = tag << 32 | 1 << 31 | entries is written once, when the scan of the query ends.
// (Round 3 stored every entry alone — an 8-byte fabric write each — and a progress word after every step that appended: 24.4 MB of
// WRITE_SIZE for 5 MB of list payload per 8000 x 10 000 launch, and the replay's progress -> entries loads were a dependent pair.)
// MODE 0: two-launch form (entry-major superset lists); 1: STREAM (tagged pair records, see above); 2: SHARD — the list of a tile scan
// (uh_knn_scan_shard_dev): row-major [query][cap] words dist << 32 | global row, EXACTLY the rows the reference's heap would accept
// on this tile (the serial walk over a step's passing lanes applies the tightened threshold before listing a row), in row order.
enum : int { kScanTwoLaunch = 0, kScanStream = 1, kScanShard = 2 };
constexpr int kVecStepMin = 4;   // passing rows of a 64-row step from which the step is decided for all rows at once (accept_scan::feed_vec)
template <int QPW, int MODE, int DEPTH = 2>
__device__ __forceinline__ void accept_scan(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k, int maxd,
    uint64_t* __restrict__ cand, int32_t* __restrict__ counts, int cap, int wave, uint64_t* __restrict__ prog, unsigned tag) {
    constexpr bool STREAM = MODE == kScanStream, SHARD = MODE == kScanShard;
    const int lane = threadIdx.x & (kWave - 1);
    const int q0 = __builtin_amdgcn_readfirstlane(wave * QPW);
    if (q0 >= nq) return;
    auto put = [&](size_t at, int d, int idx) { cand[at] = ((uint64_t)(uint32_t)d << 32) | (uint32_t)idx; };   // (two-launch form)
    uint32_t q[QPW][8];
    unsigned pend[QPW];   // STREAM: the open record of each query (the first entry of a pair waits here), wave-uniform
    int sv[QPW];    // every 16-lane row: the smallest distances so far, ascending with the lane (INT_MAX where nothing has been seen yet)
    int thr[QPW];   // wave-uniform: the k-th smallest = lane k-1 of a row
    int nc[QPW];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        load_query(queries, q0 + j < nq ? q0 + j : nq - 1, q[j]);
        sv[j] = 0x7fffffff; thr[j] = 0x7fffffff;
        nc[j] = 0; pend[j] = 0;
    }
    // STREAM: the open record of each query (the first entry of a pair waits here), wave-uniform
    typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc(cand, 0, 0x7FFFFFFF, 0x00020000);
    auto store_record = [&](int rec, int qj, unsigned w0, unsigned w1) {   // aux 16 = sc1: write-through, like ba_persist.hpp's tst()
        if (lane == 0) {
            const st_u32x4 w = {w0, tag, w1, tag};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_rec, (int)(((size_t)rec * nq + qj) * 16), 0, 16);
        }
    };
    // one accepted entry (wave-uniform d, idx), in list order; nc[j] counts it whether it is stored or not (overflow is detected by the count)
    auto append = [&](int j, int qj, int d, int idx) {
        const int p = nc[j]++;
        if (p >= cap) return;
        if constexpr (SHARD) { if (lane == 0) cand[(size_t)qj * cap + p] = ((uint64_t)(uint32_t)d << 32) | (uint32_t)idx; return; }
        const unsigned w = ((unsigned)d << 23) | (unsigned)idx;
        if (!(p & 1)) pend[j] = w;
        else store_record(p >> 1, qj, pend[j], w);
    };
    // set bits of a wave mask below this lane: v_mbcnt_lo / _hi (a precomputed (1 << lane) - 1 was two registers live through the whole scan —
    // spilled to scratch at 96 registers: 2 MB of scratch writes per 8000-query launch)
    auto below = [](unsigned long long mask) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u)); };
    // sorted insert of an accepted distance: every lane whose value exceeds d takes max(left neighbour, d) — its neighbour's value if that
    // also exceeds d, else d itself; the largest value falls off the end of the row.  Four vector instructions + one v_readlane.
    auto tighten = [&](int j, int dl) {
        const int left = __builtin_amdgcn_update_dpp(0, sv[j], 0x111, 0xF, 0xF, false);   // row_shr:1 (lane 0 of a row: 0)
        sv[j] = sv[j] > dl ? max(left, dl) : sv[j];
        thr[j] = rl(sv[j], k - 1);
    };
    // one 64-row step of query j.  Lists are ENTRY-major (entry e of query q at cand[e * nq + q]): the replay reads them one lane per query.
    // The first step of a scan is taken row by row, exactly as the reference accepts (its 64 rows all lie below the initial threshold:
    // emitting them wholesale made every list 64 - k (1 + ln(64 / k)) entries longer); afterwards all rows below the threshold at the
    // start of the step are appended at once, then the threshold is tightened with them.
    // A step with MANY rows below the threshold — the first steps of a scan: 64, ~17, ~12, ~9 ... of the step's 64 rows pass, half of all
    // the pushes of a query fall into its first 256 rows — is decided for all its rows at once instead of one serial walk iteration (~40
    // dependent instructions) per passing row.  Row i is accepted (resultset.h:66-69, dist < worst) iff fewer than k of ALL earlier rows are
    // <= d_i; of those only the k smallest so far (sv) and the step's own earlier passing rows can be that small, so with
    // e_i = #{passing l < i : d_l <= d_i} the test is sv[k - 1 - e_i] > d_i (sv ascending): one short loop over the passing lanes that
    // counts e (a readlane, two compares, an add), one ds_bpermute.  The list gets EXACTLY the accepted rows, in row order (the serial walk
    // of the stream form lists every passing row: entries the replay only rejects); the threshold is tightened with the accepted
    // distances.  Stream form: the accepted words are compacted along the lanes (one ds_permute: accepted lanes to their list slots, the
    // others behind them — a permutation, no two lanes push to one slot) and leave as whole records, one 16-byte store per LANE.
    // FIRST mode — the first step of a scan (empty list, no threshold yet; 64 serial walk iterations of ~200 clocks each were 10 us of a
    // 48 us scan wave, two queries per wave): e and the full rank of every row come from 63 + 63 wave shifts (v_mov_dpp wave_shr:1 /
    // wave_shl:1, a compare and an add-with-carry each — no scalar instruction in the chain), the 64 distances are SORTED by one
    // ds_permute (rank -> lane) and the k smallest replicated into the four rows of sv.
    auto feed_vec = [&](int j, int d, int idx, bool pass, uint64_t m, int qj, bool first) -> bool {
        int e = 0;
        if (first) {
            const int key = pass ? d : 0x7ffffffe;   // (one below the shifts' fill value: a vacated lane never counts, rows that do not pass tie among themselves)
            int sh = key, rk = 0;
#pragma nounroll   // (unrolled — even partly: the compiler finishes the job — the scheduler hoists the 63 shifts and the kernel needs 179 registers)
            for (int t = 1; t < kWave; ++t) {
                sh = __builtin_amdgcn_update_dpp(0x7fffffff, sh, 0x138, 0xF, 0xF, false);   // wave_shr:1: lane i <- lane i - 1 (lane 0 keeps "nothing")
                e += sh <= key ? 1 : 0;
            }
            sh = key;
#pragma nounroll
            for (int t = 1; t < kWave; ++t) {
                sh = __builtin_amdgcn_update_dpp(0x7fffffff, sh, 0x130, 0xF, 0xF, false);   // wave_shl:1: lane i <- lane i + 1
                rk += sh < key ? 1 : 0;
            }
            rk += e;   // rows before i that are <= d_i + rows behind i that are < d_i: a permutation of 0 .. 63 (rows that do not pass sort last, in row order)
            const int sorted = __builtin_amdgcn_ds_permute(rk << 2, key);
            int lsel = lane;   // (through an opaque move: hoisted out of the scan loop, (lane & 15) << 2 is one more register live for the whole kernel — spilled at 96)
            asm volatile("" : "+v"(lsel));
            const int s16 = __builtin_amdgcn_ds_bpermute((lsel & 15) << 2, sorted);
            sv[j] = s16 >= 0x7ffffffe ? 0x7fffffff : s16;
        } else {
            for (uint64_t mm = m; mm;) {
                const int l = __builtin_ctzll(mm);
                mm &= mm - 1;
                const int dl = rl(d, l);
                e += (dl <= d && l < lane) ? 1 : 0;
            }
        }
        const int kth = first ? 0x7fffffff : __builtin_amdgcn_ds_bpermute(max(k - 1 - e, 0) << 2, sv[j]);   // (lane r of every 16-lane row of sv = the r-th smallest)
        const bool acc = pass && e < k && d < kth;
        const uint64_t ma = __ballot(acc);
        const int na = __popcll(ma), base = nc[j];
        const int posl = below(ma);
        if constexpr (STREAM) {
            const int odd = base & 1, tot = na + odd;
            if (tot > kWave || base + na > cap) return false;   // (left to the serial walk: a list about to overflow; 64 accepted rows behind a pending entry)
            const unsigned w = ((unsigned)d << 23) | (unsigned)idx;
            const int slot = acc ? posl + odd : tot + below(~ma);   // (with a pending entry the last of the others wraps to slot 0, which is the pending entry's)
            unsigned c = (unsigned)__builtin_amdgcn_ds_permute((slot & (kWave - 1)) << 2, (int)w);
            if (odd && lane == 0) c = pend[j];
            const unsigned w0 = (unsigned)__builtin_amdgcn_ds_bpermute((2 * lane) << 2, (int)c);
            const unsigned w1 = (unsigned)__builtin_amdgcn_ds_bpermute((2 * lane + 1) << 2, (int)c);
            if (lane < (tot >> 1)) {
                const st_u32x4 rec = {w0, tag, w1, tag};
                __builtin_amdgcn_raw_buffer_store_b128(rec, rs_rec, (int)(((size_t)((base >> 1) + lane) * nq + qj) * 16), 0, 16);
            }
            if (tot & 1) pend[j] = (unsigned)rl((int)c, tot - 1);
        } else if constexpr (SHARD) {
            if (acc && base + posl < cap) cand[(size_t)qj * cap + base + posl] = ((uint64_t)(uint32_t)d << 32) | (uint32_t)idx;
        } else {
            if (acc && base + posl < cap) put((size_t)(base + posl) * nq + qj, d, idx);
        }
        nc[j] = base + na;
        if (!first)
            for (uint64_t mm = ma; mm;) {
                const int l = __builtin_ctzll(mm);
                mm &= mm - 1;
                const int dl = rl(d, l);
                const int left = __builtin_amdgcn_update_dpp(0, sv[j], 0x111, 0xF, 0xF, false);
                sv[j] = sv[j] > dl ? max(left, dl) : sv[j];
            }
        thr[j] = rl(sv[j], k - 1);
        return true;
    };
    auto feed = [&](int j, int d, int idx, bool valid, int qj, bool exact) {
        const bool pass = valid && (maxd < 0 || d <= maxd) && d < thr[j];
        uint64_t m = __ballot(pass);
        if (!m) return;
        if (__popcll(m) >= kVecStepMin && feed_vec(j, d, idx, pass, m, qj, exact && nc[j] == 0)) return;
        if (exact) {
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int dl = rl(d, l);
                if (dl >= thr[j]) continue;
                const int il = rl(idx, l);
                if constexpr (STREAM || SHARD) append(j, qj, dl, il);
                else { if (lane == 0 && nc[j] < cap) put((size_t)nc[j] * nq + qj, dl, il); nc[j]++; }
                tighten(j, dl);
            }
            return;
        }
        if constexpr (!STREAM && !SHARD) {
            const int pos = nc[j] + below(m);
            if (pass && pos < cap) put((size_t)pos * nq + qj, d, idx);
            nc[j] += __popcll(m);
        }
        if constexpr (STREAM) {
            // round 5: the stream form lists exactly the accepted rows too (it listed every row below the step's initial threshold and left the
            // rejection to the replay), and the walk reads ONE packed word per row — the distance is its top nine bits
            const unsigned wv = ((unsigned)d << 23) | (unsigned)idx;
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const unsigned wl = (unsigned)rl((int)wv, l);
                const int dl = (int)(wl >> 23);
                if (dl >= thr[j]) continue;
                const int p = nc[j]++;
                if (p < cap) { if (!(p & 1)) pend[j] = wl; else store_record(p >> 1, qj, pend[j], wl); }
                tighten(j, dl);
            }
            return;
        }
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int dl = rl(d, l);
            if (dl >= thr[j]) continue;
            if constexpr (SHARD) append(j, qj, dl, rl(idx, l));    // exactly the accepted rows
            tighten(j, dl);
        }
    };
    constexpr int UNROLL = 4;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(train), 0, t1 * 32, 0x00020000);
    const int voff = lane * 32;
    int base = t0;
    constexpr int G = UNROLL * kWave;
    // Two groups of 256 rows alternate: while one is scored the other's eight 16-byte loads are in flight (the wave used to wait a
    // whole L1 / L2 latency at the top of every group, hidden only by the three other waves of its SIMD).
    auto fetch = [&](int b, uint4 (&x0)[UNROLL], uint4 (&x1)[UNROLL]) {
        const int soff = __builtin_amdgcn_readfirstlane(b * 32);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 y0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32, soff, 0);
            const u32x4 y1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32 + 16, soff, 0);
            x0[u] = make_uint4(y0.x, y0.y, y0.z, y0.w);
            x1[u] = make_uint4(y1.x, y1.y, y1.z, y1.w);
        }
    };
    auto score = [&](int b, const uint4 (&a0)[UNROLL], const uint4 (&a1)[UNROLL]) {
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            int d[UNROLL];
            bool any = false;
            const int tj = thr[j];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { d[u] = hamming256(a0[u], a1[u], q[j]); any = any || d[u] < tj; }
            if (__ballot(any) == 0) continue;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) feed(j, d[u], b + u * kWave + lane, q0 + j < nq, q0 + j < nq ? q0 + j : 0, u == 0 && b == t0);
        }
    };
    // DEPTH groups of 256 rows rotate: while one is scored the loads of the next DEPTH - 1 are in flight.  Two buffers (rounds 2-4) hide an
    // L2 round trip only behind ANOTHER wave's work: with one scan wave per SIMD (one frame's 2000 queries) a group took ~2900 clocks for
    // ~800 of arithmetic — the wide-register form of the stream kernel rotates four (128 registers of loads).
    if (const int ng = (t1 - base) / G; ng > 0) {
        uint4 x0[DEPTH][UNROLL], x1[DEPTH][UNROLL];
        auto gaddr = [&](int g) { return t0 + min(g, ng - 1) * G; };   // (clamped: the wave-uniform part of the address is not bounds-checked)
        static_for<0, DEPTH - 1>([&](auto sc) { constexpr int i = decltype(sc)::value; fetch(gaddr(i), x0[i], x1[i]); });
        int g = 0;
        for (; g + DEPTH <= ng; g += DEPTH) {
            static_for<0, DEPTH>([&](auto sc) {
                constexpr int i = decltype(sc)::value, nb = (i + DEPTH - 1) % DEPTH;
                fetch(gaddr(g + i + DEPTH - 1), x0[nb], x1[nb]);
                score(t0 + (g + i) * G, x0[i], x1[i]);
            });
        }
        static_for<0, DEPTH - 1>([&](auto sc) { constexpr int i = decltype(sc)::value; if (g + i < ng) score(t0 + (g + i) * G, x0[i], x1[i]); });
        base = t0 + ng * G;
    }
    for (; base < t1; base += kWave) {
        const int t = base + lane;
        const bool valid = t < t1;
        uint4 x0 = make_uint4(0, 0, 0, 0), x1 = x0;
        if (valid) {
            const uint4* p = reinterpret_cast<const uint4*>(train + (size_t)t * 32);
            x0 = p[0];
            x1 = p[1];
        }
#pragma unroll
        for (int j = 0; j < QPW; ++j) feed(j, hamming256(x0, x1, q[j]), t, valid && q0 + j < nq, q0 + j < nq ? q0 + j : 0, base == t0);
    }
#pragma unroll
    for (int j = 0; j < QPW; ++j)
        if (q0 + j < nq) {
            if constexpr (STREAM) {
                const int stored = nc[j] < cap ? nc[j] : cap;
                if (stored & 1) store_record(stored >> 1, q0 + j, pend[j], 0xFFFFFFFFu);   // the odd tail: second half = "no entry"
                if (lane == 0) __hip_atomic_store(prog + q0 + j, ((uint64_t)tag << 32) | 0x80000000u | (uint32_t)nc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane == 0) counts[q0 + j] = nc[j];   // (two-launch and shard forms)
        }
}
template <int QPW>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_accept_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k, int maxd,
    uint64_t* __restrict__ cand, int32_t* __restrict__ counts, int cap, int* __restrict__ redo_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *redo_count = 0;   // (the replay launch behind this one counts the overflowed lists)
    accept_scan<QPW, kScanTwoLaunch>(train, t0, t1, queries, nq, k, maxd, cand, counts, cap, blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), nullptr, 0u);
}
// the tile scan of the sharded search for k <= 16: two queries per wave, threshold kept sorted along a DPP row (no heap pushes in the scan)
// gridDim.y > 1: the range [t0, t1) is cut into gridDim.y consecutive slices, slice s = blockIdx.y with its lists at cand + s * nq * cap and
// its counts at counts + s * nq — the latency form of the exact search for ONE frame's queries (uh_knn_search_dev below): a query's scan
// is spread over several waves, the replay walks the slices' lists in row order.
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_scan_shard2_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k, int maxd,
    uint64_t* __restrict__ cand, int32_t* __restrict__ counts, int cap, int* __restrict__ redo_count, const int32_t* __restrict__ valid_rows) {
    if (redo_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *redo_count = 0;   // (the replay launch behind this one counts the overflowed lists)
    const int S = gridDim.y, s = blockIdx.y;
    const int b0 = t0 + (int)((long long)(t1 - t0) * s / S), b1 = t0 + (int)((long long)(t1 - t0) * (s + 1) / S);
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    int nqv = nq;
    if (valid_rows) {   // uh_knn_set_valid_rows_dev: query rows from *valid_rows on are not scanned and get an EMPTY list (no overflow can come from them)
        nqv = min(nq, max(*valid_rows, 0));
        if ((threadIdx.x & (kWave - 1)) < 2) { const int q = 2 * wave + (threadIdx.x & 1); if (q >= nqv && q < nq) counts[(size_t)s * nq + q] = 0; }
    }
    accept_scan<2, kScanShard>(train, b0, b1, queries, nqv, k, maxd, cand + (size_t)s * nq * cap, counts + (size_t)s * nq, cap, wave, nullptr, 0u);
}

constexpr int kRpK = 16;   // the two-phase form serves k <= 16

// ---- ResultSet in the registers of one lane (K = the search's k): resultset.h:64-135, the sequential swaps of the reference.
// Every index below is a compile-time constant (static_for / template recursion, not unrolled loops with early exits): one index the
// optimiser cannot fold sends the whole heap to scratch memory.
// An entry is ONE 32-bit word, distance << 23 | row index (distance <= 256, fewer than 2^23 rows — the host checks): every move of the
// sifts is one select instead of two.  The heap orders by distance alone (ties never move an entry): d(a) < d(b) <=> a < (b & kDistMask).
constexpr unsigned kDistMask = 0xFF800000u;
__device__ __forceinline__ bool dist_less(unsigned a, unsigned b) { return a < (b & kDistMask); }

// sift the new element up from STATIC slot P (:93-100).  cmz = the new word's distance bits for the lanes that push, 0 for the others:
// "parent < new" is then false by itself in a lane that does not push, and in a pushing lane the heap order (a parent is never closer
// than its child) makes "the climb reached this slot" redundant — d(parent) < d(new) implies d(child) < d(new) — so no step needs the
// conjunction of two lane masks (a scalar instruction between two vector ones, on a wave whose every instruction is latency).
template <int K, int P>
__device__ __forceinline__ void heap_append_step(unsigned (&hw)[K], bool up, unsigned cw, unsigned cmz) {
    if constexpr (P == 0) {
        hw[0] = up ? cw : hw[0];
    } else {
        constexpr int par = (P - 1) >> 1;
        const bool mv = hw[par] < cmz;
        hw[P] = up ? (mv ? hw[par] : cw) : hw[P];
        heap_append_step<K, par>(hw, mv, cw, cmz);
    }
}
// remove the root of a FULL heap (:104-135): the last element re-enters at the root and sinks; nodes 0 .. K-2 take part.  Level by
// level: the lane's position `pos` is one of the level's nodes (or -1: the lane does not pop, or its element has come to rest), its
// children are selected from the next level.  A child that does not exist reads as the word 0 (distance 0): nothing is closer than
// it, so the element never moves there, and "right child closer than left" is false for it — exactly the reference's bounds tests,
// without a mask to carry: the only predicates are pos == n (one compare, three selects per node) and the two distance compares.
template <int K, int LVL>
__device__ __forceinline__ void heap_pop_level(unsigned (&hw)[K], unsigned mw, int pos) {
    constexpr int NS = K - 1, first = (1 << LVL) - 1;
    if constexpr (first < NS) {
        constexpr int last = 2 * first < NS - 1 ? 2 * first : NS - 1;
        constexpr bool kids = 2 * first + 1 < NS;
        unsigned wl = 0, wr = 0;
        static_for<first, last + 1>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (2 * n + 1 < NS) wl = pos == n ? hw[2 * n + 1] : wl;
            if constexpr (2 * n + 2 < NS) wr = pos == n ? hw[2 * n + 2] : wr;
        });
        const bool pickL = dist_less(wr, wl);   // :113 (dr < dl ? left : right); only a left child: wr = 0 picks it unless d(left) = 0, and then nothing moves
        const unsigned wc = pickL ? wl : wr;
        const bool mv = dist_less(mw, wc);
        const unsigned put = mv ? wc : mw;
        static_for<first, last + 1>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            hw[n] = pos == n ? put : hw[n];
        });
        if constexpr (kids) heap_pop_level<K, LVL + 1>(hw, mw, mv ? 2 * pos + (pickL ? 1 : 2) : -1);
    }
}
template <int K>
__device__ __forceinline__ void heap_pop_full(unsigned (&hw)[K], bool acc) {
    if constexpr (K >= 2) heap_pop_level<K, 0>(hw, hw[K - 1], acc ? 0 : -1);
}
// any mixture of sizes inside the wave (lists of different lengths, max_dist): dynamic positions through select chains
template <int K>
__device__ __forceinline__ unsigned heap_get(const unsigned (&a)[K], int i) {
    unsigned r = a[0];
    static_for<1, K>([&](auto nc) { constexpr int n = decltype(nc)::value; r = i == n ? a[n] : r; });
    return r;
}
template <int K>
__device__ __forceinline__ void heap_put(unsigned (&a)[K], int i, bool on, unsigned v) {
    static_for<0, K>([&](auto nc) { constexpr int n = decltype(nc)::value; a[n] = (on && i == n) ? v : a[n]; });
}
// append only (no accepting lane is full yet — the first K entries of the lists): sift up from the lane's own size
template <int K>
__device__ __forceinline__ void heap_append_generic(unsigned (&hw)[K], int& size, bool acc, unsigned cw) {
    int pos = size;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int parent = pos > 0 ? (pos - 1) >> 1 : 0;
        const unsigned wp = heap_get<K>(hw, parent);
        const bool mv = acc && pos > 0 && dist_less(wp, cw);
        heap_put<K>(hw, pos, mv, wp);
        pos = mv ? parent : pos;
    }
    heap_put<K>(hw, pos, acc, cw);
    size = acc ? size + 1 : size;
}
template <int K>
__device__ __forceinline__ void heap_push_generic(unsigned (&hw)[K], int& size, bool acc, unsigned cw) {
    const bool pop = acc && size >= K;
    const int ns = pop ? size - 1 : size;
    const unsigned mw = heap_get<K>(hw, ns < K ? ns : K - 1);
    int pos = 0;
    bool go = pop && ns > 1;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int l = 2 * pos + 1, r = l + 1;
        const bool hasL = go && l < ns, hasR = go && r < ns;
        const unsigned wl = heap_get<K>(hw, l < K ? l : K - 1), wr = heap_get<K>(hw, r < K ? r : K - 1);
        const bool pickL = !hasR || dist_less(wr, wl);
        const int c = pickL ? l : r;
        const unsigned wc = pickL ? wl : wr;
        const bool mv = hasL && dist_less(mw, wc);
        heap_put<K>(hw, pos, mv, wc);
        pos = mv ? c : pos;
        go = mv;
    }
    heap_put<K>(hw, pos, pop && ns >= 1, mw);
    size = ns;
    heap_append_generic<K>(hw, size, acc, cw);
}

// one entry for every lane of the wave (acc: the lanes whose entry passes resultset.h:66-69)
template <int K>
__device__ __forceinline__ void heap_push_wave(unsigned (&hw)[K], int& size, bool acc, unsigned cw) {
    if (__ballot(acc && size != K) == 0) {                      // steady state: every accepting lane's heap is full
        heap_pop_full<K>(hw, acc);
        heap_append_step<K, K - 1>(hw, acc, cw, acc ? cw & kDistMask : 0u);
    } else if (__ballot(acc && size >= K) == 0) {               // filling: nobody has to remove a root (select chains, no dynamic index:
        heap_append_generic<K>(hw, size, acc, cw);              // a dispatch on the common size ends up as an indexed store to scratch)
    } else {
        heap_push_generic<K>(hw, size, acc, cw);
    }
}
// linear.h:82-85 (fill) + index.h:119-134 (exchange sort; idx[i] is tested once, before the inner loop)
template <int K>
__device__ __forceinline__ void heap_write_row(const unsigned (&hw)[K], int size, int sorted, bool on_q, int qi, int32_t* __restrict__ indices,
                                               int32_t* __restrict__ distances) {
    int hd[K], hi[K];
#pragma unroll
    for (int i = 0; i < K; i++) { const bool on = i < size; hd[i] = on ? (int)(hw[i] >> 23) : 0; hi[i] = on ? (int)(hw[i] & 0x7FFFFFu) : -1; }
    if (sorted) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            const bool on = hi[i] != -1;
#pragma unroll
            for (int j = i + 1; j < K; ++j) {
                const bool sw = on && hd[i] > hd[j];
                const int td = hd[i], ti = hi[i];
                hd[i] = sw ? hd[j] : td; hi[i] = sw ? hi[j] : ti;
                hd[j] = sw ? td : hd[j]; hi[j] = sw ? ti : hi[j];
            }
        }
    }
    if (on_q) {
#pragma unroll
        for (int i = 0; i < K; i++) { indices[(size_t)qi * K + i] = hi[i]; distances[(size_t)qi * K + i] = hd[i]; }
    }
}

// A wave's staged entries (LDS, entry u of the lane's query at stage[u * 64]; kNoEntry where the lane has none), pushed in order
// (resultset.h:66-69; the radius bound of :66 has been applied by the scan).  Two loops: while some lane that still expects entries is
// not full, the general push; from then on — every list after its first K entries — the steady-state form whose every step is one
// compare + branch when no lane accepts and root removal + append otherwise, with nothing else in the loop: on a wave that owns its
// SIMD's issue slot every instruction of the step costs ~10 cycles, loop bookkeeping included.
constexpr unsigned kNoEntry = 0xFFFFFFFFu;   // (distance 511: closer than nothing)
template <int K>
__device__ __forceinline__ void replay_staged(unsigned (&hw)[K], int& size, const unsigned* stage, int steps, bool live) {
    if (steps <= 0) return;
    int u = 0;
    unsigned nxt = stage[0];
    if (__ballot(live && size != K) != 0) {
        for (; u < steps;) {
            const unsigned cw = nxt;
            ++u;
            nxt = stage[min(u, steps - 1) * kWave];
            // the first K entries of the lists, all live lanes in step (the first round of a replay wave: every list opens with the ~28 rows
            // its scan accepts among the first 64): the append slot is the same constant in every lane — a 3-step climb instead of the
            // select chains of the general form (~165 instructions per entry, 4 us per wave for K = 10)
            const unsigned long long lv = __ballot(live);
            const int s0 = lv ? rl(size, __builtin_ctzll(lv)) : K;   // (the size of the first live lane)
            if (s0 < K && __ballot(live && (size != s0 || cw == kNoEntry)) == 0) {
                static_for<0, K>([&](auto pc) {
                    constexpr int P = decltype(pc)::value;
                    if (s0 == P) heap_append_step<K, P>(hw, live, cw, live ? cw & kDistMask : 0u);
                });
                size = live ? size + 1 : size;
            } else {
                const bool acc = cw != kNoEntry && (size < K || dist_less(cw, hw[0]));
                if (__ballot(acc)) heap_push_wave<K>(hw, size, acc, cw);
            }
            if (__ballot(live && size != K) == 0) break;
        }
    }
    for (; u < steps;) {
        const unsigned cw = nxt;
        ++u;
        nxt = stage[min(u, steps - 1) * kWave];
        const bool acc = dist_less(cw, hw[0]);
        if (!__ballot(acc)) continue;
        heap_pop_full<K>(hw, acc);
        heap_append_step<K, K - 1>(hw, acc, cw, acc ? cw & kDistMask : 0u);
    }
}

// Accept scan and replay in ONE launch of one-wave workgroups: the first `nrep` replay (64 queries each, one lane per query, the
// register heap of knn_replay_lane_kernel), the others scan two queries each (accept_scan<2, true>).  The replay of a query is a chain
// of ~k (1 + ln(N/k)) pushes of ~0.4 us each whatever the number of queries (47 us behind an 87 us scan in the two-launch form, on 125
// of the chip's 1024 SIMDs); here a replay lane consumes its query's list while the scan is still appending to it, and the launch ends
// a few microseconds after the last scan wave (8000 x 10 000, nn 10: scan waves end at 43..98 us — the SIMDs serve their oldest wave
// first —, the replay waves 2..28 us behind their last one, the launch at 105 us; nn 2: 1..4 us behind).  What it costs: the replay
// waves run at raised priority beside scan waves and take issue slots from them, the lists travel as write-through stores, and the
// scan's registers are capped by the replay's: the streaming scan alone is ~8 % slower than knn_accept_kernel, which is why short
// replays (nn <= 5) keep the two launches.  One-wave workgroups: with 4125 workgroups the dispatcher evens out the SIMDs (four-wave
// workgroups left some CUs a fifth one and the launch 20 us longer).
// The scanning waves wait for nobody, the replay workgroups are a small fixed part of the grid and come FIRST (resident before any
// scan workgroup could crowd them out), so the launch always makes progress; a replay lane that sees no progress for two seconds hands
// its query to knn_redo_kernel (as it does with an overflowed list), so even then the rows are right.
// Protocol: accept_scan<., true>.  A replay wave fetches, per lane and round, the query's closing word and its next records (two, or
// kStreamStage / 2 when the previous round filled the short fetch; all loads in flight together, none depends on another), keeps the prefix
// of records whose two tags match, stages their entries in LDS and pushes them.
constexpr int kStreamStage = 16;
constexpr int kMinRec = 2;   // (0 / 1 / 2 / 3 at 2000 queries: 56.0 / 47.0 / 46.7 / 48.1 us per search)
// a replay lane that sees no new record for this long hands its query to knn_redo_kernel (the rows stay right; only time is lost):
// 100 ms by default (UH_KNN_STREAM_TIMEOUT_MS; round 3: 2 s) — the scan of a whole launch takes ~0.1 ms
constexpr long long kStreamTimeoutDefault = 10000000ll;   // 100 ms of the 100 MHz wall clock
// W = waves per SIMD the launch is compiled for: 5 (102 registers) when the scan workgroups outnumber the chip's SIMDs several times
// (8000 queries: 4125 workgroups, two queries per scan wave, two row groups in rotation), 2 for one frame's queries (2000 + 32
// workgroups on 1024 SIMDs): ONE query per scan wave — twice the waves for the same arithmetic, every SIMD holds two, the serial
// head of a scan (the first 256 rows) overlaps with the other wave's — and four row groups in rotation (128 registers of loads).
// 2000 x 10 000, nn 10 (MI355X, HIP events, round 5): 72.3 us -> 46.7 with the first-step / lockstep / static-fill changes above and
// this form; scan waves end at 24-38 us (were 41-57), the replay waves 7 us behind the last one (were 15).
// The overflowed (or timed-out) queries are redone by the replay workgroups THEMSELVES once all of them have finished (a ticket per
// workgroup; they are co-resident from the first cycle of the launch): no dependent knn_redo_kernel launch behind every search
// (4.5 us + the launch gap, always paid, almost never needed).
// Host form (uh_knn_search with pinned result arrays): the replay workgroups copy the finished rows to the host arrays themselves — 16 bytes
// per lane and store: a row written where it is produced would be ten 4-byte PCIe writes — and the last of them posts the completion word
// (projmatch.hip's PmPublish): one launch instead of search + two copies + word.
struct KnnHostOut {
    int32_t* host_idx; int32_t* host_dist; unsigned n16;          // pinned twins of `indices` / `distances`, 16-byte units per array
    unsigned long long* host_done; unsigned long long word;      // pinned completion word (NULL: device form)
};
template <int K, int W>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(W, W))) void knn_stream_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int sorted, int maxd,
    uint64_t* __restrict__ cand, uint64_t* __restrict__ prog, int cap, unsigned tag, int nrep,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances, int* __restrict__ redo_list, int* __restrict__ redo_count, int* __restrict__ redo_next,
    long long timeout_ticks, KnnHostOut ho) {
    // redo_count[0] = overflowed queries of this launch, [2] = replay workgroups that have finished their lists, [4] = ... their redo share,
    // [6] = ... their part of the copy to the host; redo_next likewise for the next launch
    if ((int)blockIdx.x >= nrep) {
        accept_scan<(W < 5 ? 1 : 2), kScanStream, (W < 4 ? 4 : 2)>(train, t0, t1, queries, nq, K, maxd, cand, nullptr, cap, (int)blockIdx.x - nrep, prog, tag);
        return;
    }
    __shared__ unsigned s_stage[kStreamStage * kWave];
    if (blockIdx.x == 0 && threadIdx.x == 0) { redo_next[0] = 0; redo_next[2] = 0; redo_next[4] = 0; redo_next[6] = 0; }   // the NEXT launch's counters (this launch's were cleared by the previous one)
    __builtin_amdgcn_s_setprio(3);                              // a dependent chain beside throughput-bound scan waves: issue first
    const int lane = threadIdx.x & (kWave - 1);
    unsigned* stage = s_stage + lane;
    const int qi = blockIdx.x * kWave + threadIdx.x;
    const bool haveq = qi < nq;
    const uint64_t* pq = prog + (haveq ? qi : 0);
    typedef unsigned int ld_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc(cand, 0, 0x7FFFFFFF, 0x00020000);
    const int nrec_cap = (cap + 1) >> 1;                  // records per list
    constexpr int kRec = kStreamStage / 2;                 // records fetched per lane and round (all loads in flight together)
    unsigned hw[K];
#pragma unroll
    for (int i = 0; i < K; i++) hw[i] = 0;
    int size = 0, r = 0;                                   // r: records consumed
    bool fin = !haveq, over = false;
    int deep = 0;                                          // wave-uniform: the previous round filled its short fetch -> fetch the long one
    long long tlast = wall_clock64();
    // One round: the closing word (written once, when the query's scan ends) and the next records, all loads independent of each other.
    // (Round 5 tried issuing the NEXT round's loads before pushing this round's entries: the records it sees are one round old, the rounds
    // get smaller and more numerous — 2000 queries: 75 -> 87 us.)
    uint64_t pw;
    ld_u32x4 v[kRec];
    const int qv = haveq ? qi : 0;
    auto fetch = [&](int rr, int dp) {
        pw = __hip_atomic_load(pq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 2; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, (int)(((size_t)min(rr + u, nrec_cap - 1) * nq + qv) * 16), 0, 16);   // aux 16 = sc1
        if (dp) {
#pragma unroll
            for (int u = 2; u < kRec; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, (int)(((size_t)min(rr + u, nrec_cap - 1) * nq + qv) * 16), 0, 16);
        } else {
#pragma unroll
            for (int u = 2; u < kRec; u++) v[u] = ld_u32x4{0u, 0u, 0u, 0u};
        }
    };
    while (__ballot(!fin) != 0) {
        fetch(r, deep);
        const bool closed = !fin && (unsigned)(pw >> 32) == tag && ((unsigned)pw & 0x80000000u) != 0;
        const int nc = closed ? (int)((unsigned)pw & 0x7fffffffu) : 0;
        if (closed && nc > cap) { over = true; fin = true; }
        const int total = closed ? (min(nc, cap) + 1) >> 1 : nrec_cap;   // records this list will hold (once known)
        int np = 0;
        bool run = !fin;
#pragma unroll
        for (int u = 0; u < kRec; u++) {
            run = run && r + u < total && v[u].y == tag && v[u].w == tag;   // both halves of the record carry this launch's tag
            np += run ? 1 : 0;
            stage[(2 * u) * kWave] = run ? v[u].x : kNoEntry;
            stage[(2 * u + 1) * kWave] = run ? v[u].z : kNoEntry;
        }
        int most = 0;   // max over the lanes of np (0 .. kRec): eight independent ballots instead of a six-deep shuffle chain
#pragma unroll
        for (int c = 1; c <= kRec; c++) most = __ballot(np >= c) != 0 ? c : most;
        deep = most >= 2 ? 1 : 0;
        // The 64 queries of the wave push in LOCKSTEP: a round costs max-over-lanes steps, so rounds in which a few lanes have one or two
        // entries each (the trickle behind the first 256 rows of a scan: ~0.4 entries per query and microsecond) made the wave take 2-3
        // steps per entry it consumed.  A round now starts only when EVERY unfinished lane has kMinRec records waiting or all it
        // will ever get (its scan has closed the list): the lanes then hold about the same number of entries.
        const bool ready = fin || np >= kMinRec || (closed && r + np >= total);
        const bool wait_more = __ballot(!ready) != 0 && most != 0;
        if (most == 0 || wait_more) {
            if (closed && r >= total) fin = true;
            if (wall_clock64() - tlast > timeout_ticks) { over = over || !fin; fin = true; }
            // ~0.6 us between polls (every poll is a fabric read of each lane's next records).  Round 6 tried a back-off for waves that find nothing
            // new (1.3, then 1.9 us): kernel 136.5 us and FETCH_SIZE x 2 18.5 MB inside the headline loop, both unchanged — the polls are not where
            // the 3.6 MB of round 5 went; removed again.
            if (__ballot(!fin) != 0) __builtin_amdgcn_s_sleep(24);
            continue;
        }
        tlast = wall_clock64();
        const int rn = r + np;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): a lane reads back only what it staged itself
        replay_staged<K>(hw, size, stage, 2 * most, !fin);
        r = rn;
        if (closed && r >= total) fin = true;
    }
    if (over) redo_list[atomicAdd(redo_count, 1)] = qi;
    heap_write_row<K>(hw, size, sorted, haveq && !over, qi, indices, distances);
    // ---- the queries to redo (overflowed lists, time-outs), shared among the replay workgroups once ALL of them are through
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_fetch_add(redo_count + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(redo_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nrep) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int nredo = __builtin_amdgcn_readfirstlane(__hip_atomic_load(redo_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (nredo != 0) {
        __builtin_amdgcn_s_setprio(0);
        constexpr int LV = K <= 3 ? 1 : K <= 15 ? 3 : 6;   // (knn_search_kernel's heap levels for this k)
        for (int j = blockIdx.x; j < nredo; j += nrep) {
            const int rq = __builtin_amdgcn_readfirstlane(__hip_atomic_load(redo_list + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            uint32_t q[8];
            load_query(queries, rq, q);
            WaveHeap h{0, -1, 0, lane};
            int ncand = 0;
            scan_range<false, LV>(h, train, t0, t1, q, K, maxd, nullptr, ncand, 0);
            finish_row(h, K, sorted, indices, distances, rq);
        }
        if (ho.host_done) {   // the redone rows of every workgroup before anybody copies
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) __hip_atomic_fetch_add(redo_count + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(redo_count + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nrep) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (!ho.host_done) return;
    {
        const uint4* si = reinterpret_cast<const uint4*>(indices);
        const uint4* sd = reinterpret_cast<const uint4*>(distances);
        uint4* di = reinterpret_cast<uint4*>(ho.host_idx);
        uint4* dd = reinterpret_cast<uint4*>(ho.host_dist);
        for (unsigned i = blockIdx.x * kWave + lane; i < ho.n16; i += (unsigned)nrep * kWave) { di[i] = si[i]; dd[i] = sd[i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every lane: its stores into pinned memory before the ticket, and so before the word
        int last = 0;
        if (lane == 0) last = __hip_atomic_fetch_add(redo_count + 6, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nrep - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(ho.host_done, ho.word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int K>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void knn_replay_lane_kernel(
    const uint64_t* __restrict__ cand, const int32_t* __restrict__ counts, int nq, int sorted, int maxd, int cap,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances, int* __restrict__ redo_list, int* __restrict__ redo_count) {
    extern __shared__ uint64_t s_list[];   // [cap][64]: this wave's 64 lists, staged with every load in flight at once
    const int qi = blockIdx.x * kWave + threadIdx.x;
    const bool haveq = qi < nq;
    const int cnt_raw = haveq ? counts[qi] : 0;
    const bool over = cnt_raw > cap;
    if (over) redo_list[atomicAdd(redo_count, 1)] = qi;
    const int cnt = over ? 0 : cnt_raw;
    int maxcnt = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o));
    maxcnt = __builtin_amdgcn_readfirstlane(maxcnt);   // (tell the compiler: the loop below is wave-uniform)
    const uint64_t* col = cand + (haveq ? qi : 0);   // entry e of this lane's query: col[e * nq] (coalesced across the wave)
    unsigned hw[K];
#pragma unroll
    for (int i = 0; i < K; i++) hw[i] = 0;
    int size = 0;
    // Stage the lists in LDS first: maxcnt coalesced loads (512 bytes each), all in flight together = one memory round trip per wave.
    // (64 entries per batch: a batch is one memory round trip however many loads it holds, and the wave has the register file to itself)
    for (int e0 = 0; e0 < maxcnt; e0 += 64) {
        uint64_t v[64];
#pragma unroll
        for (int u = 0; u < 64; u++) v[u] = col[(size_t)min(e0 + u, cap - 1) * nq];
#pragma unroll
        for (int u = 0; u < 64; u++) if (e0 + u < maxcnt) s_list[(e0 + u) * kWave + threadIdx.x] = v[u];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the lane reads back only what it wrote itself
    uint64_t nxt = s_list[threadIdx.x];
    for (int e = 0; e < maxcnt; e++) {
        const uint64_t cur = nxt;
        nxt = s_list[min(e + 1, maxcnt - 1) * kWave + threadIdx.x];
        const int d = (int)(cur >> 32);
        const unsigned cw = ((unsigned)d << 23) | (unsigned)(uint32_t)cur;
        const bool valid = e < cnt && !(maxd >= 0 && maxd < d);     // resultset.h:66
        const bool acc = valid && (size < K || dist_less(cw, hw[0]));   // :67-69
        if (!__ballot(acc)) continue;
        heap_push_wave<K>(hw, size, acc, cw);
    }
    heap_write_row<K>(hw, size, sorted, haveq && !over, qi, indices, distances);
}

// Replay of tile lists for k <= 16 with ONE LANE PER QUERY, the heap in registers like knn_replay_lane_kernel, shard after shard (= global
// row order); chosen for a single list only (see launch_replay_lanes).  A shard's lists are row-major [query][cap] words dist << 32 | row;
// a lane's list is staged in LDS with all its loads in flight (one memory round trip per shard and wave).  A list that overflowed its
// capacity sets *overflow (tile-only ranks cannot rescan another rank's rows).  The wave-per-query replay (knn_replay_kernel: ~1000
// cycles of cross-lane traffic per accepted push) stays for k > 16 and for the form that rescans.
template <int K>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 2))) void knn_replay_shards_lane_kernel(
    const uint64_t* __restrict__ cand_all, size_t cand_shard_stride, const int32_t* __restrict__ counts_all, size_t count_shard_stride, int nshards,
    int nq, int sorted, int maxd, int cap, int32_t* __restrict__ indices, int32_t* __restrict__ distances, int* __restrict__ overflow,
    int* __restrict__ redo_list, int* __restrict__ redo_count) {
    extern __shared__ uint64_t s_list[];   // [cap][64]
    const int qi = blockIdx.x * kWave + threadIdx.x;
    const bool haveq = qi < nq;
    const int qv = haveq ? qi : 0;
    unsigned hw[K];
#pragma unroll
    for (int i = 0; i < K; i++) hw[i] = 0;
    int size = 0;
    bool over = false;
    for (int s = 0; s < nshards; ++s) {
        const int cnt_raw = haveq ? counts_all[(size_t)s * count_shard_stride + qv] : 0;
        if (cnt_raw > cap) over = true;
        const int cnt = cnt_raw > cap ? 0 : cnt_raw;
        int maxcnt = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, o));
        maxcnt = __builtin_amdgcn_readfirstlane(maxcnt);
        if (maxcnt == 0) continue;
        const uint64_t* row = cand_all + (size_t)s * cand_shard_stride + (size_t)qv * cap;
        for (int e0 = 0; e0 < maxcnt; e0 += 32) {
            uint64_t v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) v[u] = row[min(e0 + u, cap - 1)];
#pragma unroll
            for (int u = 0; u < 32; u++) if (e0 + u < maxcnt) s_list[(e0 + u) * kWave + threadIdx.x] = v[u];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the lane reads back only what it wrote itself
        uint64_t nxt = s_list[threadIdx.x];
        for (int e = 0; e < maxcnt; e++) {
            const uint64_t cur = nxt;
            nxt = s_list[min(e + 1, maxcnt - 1) * kWave + threadIdx.x];
            const int d = (int)(cur >> 32);
            const unsigned cw = ((unsigned)d << 23) | (unsigned)(uint32_t)cur;
            const bool valid = e < cnt && !(maxd >= 0 && maxd < d);       // resultset.h:66
            const bool acc = valid && (size < K || dist_less(cw, hw[0]));   // :67-69
            if (!__ballot(acc)) continue;
            heap_push_wave<K>(hw, size, acc, cw);
        }
    }
    if (over && redo_list) redo_list[atomicAdd(redo_count, 1)] = qi;   // (the search entry point: the fused kernel recomputes this query)
    else if (over && overflow) atomicOr(overflow, 1);
    heap_write_row<K>(hw, size, sorted, haveq && !(over && redo_list), qi, indices, distances);
}

// the (rare) queries whose accept list overflowed: the fused one-wave search, over a compacted list
template <int LV>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_redo_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int k, int sorted, int maxd,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances, const int* __restrict__ redo_list, const int* __restrict__ redo_count) {
    const int lane = threadIdx.x & (kWave - 1);
    const int n = __builtin_amdgcn_readfirstlane(*redo_count);
    const int nw = gridDim.x * kWavesPerBlock;
    for (int j = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)); j < n; j += nw) {
        const int qi = __builtin_amdgcn_readfirstlane(redo_list[j]);
        uint32_t q[8];
        load_query(queries, qi, q);
        WaveHeap h{0, -1, 0, lane};
        int ncand = 0;
        scan_range<false, LV>(h, train, t0, t1, q, k, maxd, nullptr, ncand, 0);
        finish_row(h, k, sorted, indices, distances, qi);
    }
}

// QPW queries per wave: a group of train rows is loaded once and scored against QPW queries (their words live in SGPRs): 1/QPW of the
// L2 -> CU traffic and 1/QPW of the resident waves of the one-query form; every query keeps its own heap (two VGPRs) and its pushes
// happen in ascending row order exactly as in the one-query form.
template <int LV, int QPW>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_search_mq_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k,
    int sorted, int maxd, int32_t* __restrict__ indices, int32_t* __restrict__ distances) {
    const int lane = threadIdx.x & (kWave - 1);
    const int q0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * QPW);
    if (q0 >= nq) return;
    uint32_t q[QPW][8];
    WaveHeap h[QPW];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        load_query(queries, q0 + j < nq ? q0 + j : nq - 1, q[j]);
        h[j] = WaveHeap{0, -1, 0, lane};
    }
    constexpr int UNROLL = 4;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(train), 0, t1 * 32, 0x00020000);
    const int voff = lane * 32;
    int dummy = 0;
    int base = t0;
    for (; base + UNROLL * kWave <= t1; base += UNROLL * kWave) {
        uint4 a0[UNROLL], a1[UNROLL];
        const int soff = __builtin_amdgcn_readfirstlane(base * 32);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32, soff, 0);
            const u32x4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + u * kWave * 32 + 16, soff, 0);
            a0[u] = make_uint4(x0.x, x0.y, x0.z, x0.w);
            a1[u] = make_uint4(x1.x, x1.y, x1.z, x1.w);
        }
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
            int d[UNROLL];
            bool any = false;
            const int thr = h[j].threshold(k);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { d[u] = hamming256(a0[u], a1[u], q[j]); any = any || d[u] < thr; }
            if (__ballot(any) == 0) continue;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) feed_step<false, LV>(h[j], d[u], base + u * kWave + lane, true, k, maxd, nullptr, dummy, 0);
        }
    }
    for (; base < t1; base += kWave) {
        const int t = base + lane;
        const bool valid = t < t1;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (valid) {
            const uint4* p = reinterpret_cast<const uint4*>(train + (size_t)t * 32);
            a0 = p[0];
            a1 = p[1];
        }
#pragma unroll
        for (int j = 0; j < QPW; ++j) feed_step<false, LV>(h[j], hamming256(a0, a1, q[j]), t, valid, k, maxd, nullptr, dummy, 0);
    }
#pragma unroll
    for (int j = 0; j < QPW; ++j)
        if (q0 + j < nq) finish_row(h[j], k, sorted, indices, distances, q0 + j);
}

// Shard scan: emits the locally accepted candidates in index order.
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_scan_shard_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k,
    int maxd, uint64_t* __restrict__ cand, int32_t* __restrict__ counts, int cap, const int32_t* __restrict__ valid_rows) {
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (qi >= nq) return;
    if (valid_rows && qi >= *valid_rows) { if (lane == 0) counts[qi] = 0; return; }   // (uh_knn_set_valid_rows_dev)
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    int ncand = 0;
    scan_range<true>(h, train, t0, t1, q, k, maxd, cand + (size_t)qi * cap, ncand, cap);
    if (lane == 0) counts[qi] = ncand;
}

// Replay the concatenated shard candidate lists (shard order = index order) through the exact heap.
// A shard whose list overflowed (count > cap) is rescanned from the descriptors.
template <int LV>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_replay_kernel(
    const uint8_t* __restrict__ train, ShardBounds sb, const uint8_t* __restrict__ queries, int nq, int k,
    int sorted, int maxd, const uint64_t* __restrict__ cand_all, const int32_t* __restrict__ counts_all, int cap,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances, int* __restrict__ overflow,
    size_t cand_shard_stride, size_t count_shard_stride) {   // elements between two shards' blocks (nq * cap / nq when they are packed; a gathered message's length otherwise)
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (qi >= nq) return;
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    int dummy = 0;
    for (int s = 0; s < sb.n; ++s) {
        int cnt = __builtin_amdgcn_readfirstlane(counts_all[(size_t)s * count_shard_stride + qi]);
        if (cnt > cap && overflow) {   // tile-only ranks cannot rescan another rank's rows: report, the caller retries with a larger cap
            if (lane == 0) atomicOr(overflow, 1);
            continue;
        }
        if (cnt > cap) {
            scan_range<false, LV>(h, train, sb.b[s], sb.b[s + 1], q, k, maxd, nullptr, dummy, 0);
            continue;
        }
        const uint64_t* row = cand_all + (size_t)s * cand_shard_stride + (size_t)qi * cap;
        for (int base = 0; base < cnt; base += kWave) {
            int j = base + lane;
            bool valid = j < cnt;
            uint64_t c = valid ? row[j] : 0;
            feed_step<false, LV>(h, (int)(c >> 32), (int)(uint32_t)c, valid, k, maxd, nullptr, dummy, 0);
        }
    }
    finish_row(h, k, sorted, indices, distances, qi);
}

// ------------------------------------------------------------------------------------------------ hierarchical k-means search
// KMeansIndex::_knnsearch_nn (impl/kmeansindex.h:356-410) on the reference's own block data (see uh_knn_build_kmeans): best-bin-first
// with a branch min-heap (impl/heap.h) and the same ResultSet as the linear index.  One wave per query:
//   * a block's <= 64 children: one lane each computes its Hamming distance (32-byte features straight from the block);
//   * internal block: the reference's sequential "running best, push everything else" loop is replayed with wave-uniform
//     v_readlane reads; the branch heap (dist, offset) lives in LDS and its sifts are executed by the whole wave on broadcast
//     addresses (the order of equal-distance branches is observable, so the heap is the reference's, swap for swap);
//   * leaf block: feed_step() = the ResultSet pushes in child order.
// Bound by the dependent LDS round trips of ~31 heap pushes per internal block, not by bandwidth.
constexpr int kBranchMax = 5000;   // impl/heap.h: Heap<T, maxSize = 5000>; a push beyond it is dropped ("Heap max size reached")
constexpr int kKmWaves = 2;        // waves (queries) per workgroup: 2 x 16 KB of branch heap

struct BranchHeapLds {
    int* d; unsigned int* o;   // [cap] each, per wave; cap = min(kBranchMax, worst case of this search), sized by the host
    int size;
    int cap;                   // the LDS region really holds `cap` entries: a push beyond it is dropped (== the reference whenever cap == kBranchMax,
                               // and never an overrun into the neighbouring wave's heap should the host's worst-case bound ever be too small)
    // heap.h push + "down" (the new entry climbs while it is smaller than its parent).  Round 6: the whole climb in ONE LDS round trip — lane j
    // reads the j-th ancestor of the insertion slot (slot_j = ((size + 1) >> j) - 1), a ballot finds how far the entry climbs, the lanes of
    // the passed ancestors write them one level down and the entry takes the slot where it stopped: the same swaps as the sequential loop
    // (the order of equal-distance branches is observable), 2 reads + 2 writes per lane instead of log2(size) dependent round trips per push —
    // ~31 pushes per internal block were most of a query's time.
    __device__ __forceinline__ void push(int dist, unsigned int off, int lane) {
        if (size >= cap) return;
        const int s1 = size + 1;
        const int levels = 31 - __builtin_clz(s1);                 // ancestors of the insertion slot: j = 1 .. levels (slot_levels = 0, the root)
        const int slot = (s1 >> min(lane, 31)) - 1;                // lane 0: the insertion slot itself
        const bool anc = lane >= 1 && lane <= levels;
        const int dj = anc ? d[slot] : 0;
        const unsigned int oj = anc ? o[slot] : 0u;
        // climbs past ancestor j iff dist < d_j for every ancestor up to j: the first ancestor that holds stops it
        const unsigned long long stops = __ballot(anc && !(dist < dj)) | (1ull << (levels + 1));
        const int m = __builtin_ctzll(stops) - 1;                  // ancestors passed: 1 .. m
        if (anc && lane <= m) { const int below = (s1 >> (lane - 1)) - 1; d[below] = dj; o[below] = oj; }
        if (lane == 0) { const int at = (s1 >> m) - 1; d[at] = dist; o[at] = off; }
        size++;
    }
    __device__ __forceinline__ unsigned int pop() {                      // heap.h pop + "up"
        const unsigned int res = o[0];
        size--;
        const int md = d[size];
        const unsigned int mo = o[size];
        if (size > 1) {   // the old last element enters at the root and sinks
            int i = 0;
            for (;;) {
                const int l = 2 * i + 1, r = l + 1;
                if (l >= size) break;
                int c = l;
                int dc = d[l];
                if (r < size) { const int dr = d[r]; if (!(dc < dr)) { c = r; dc = dr; } }   // left only if strictly smaller
                if (dc < md) { d[i] = dc; o[i] = o[c]; i = c; } else break;
            }
            d[i] = md; o[i] = mo;
        } else if (size == 1) { d[0] = md; o[0] = mo; }
        return res;
    }
};

__global__ __launch_bounds__(kWave* kKmWaves) void knn_kmeans_search_kernel(
    const uint8_t* __restrict__ blob, const uint8_t* __restrict__ queries, int nq, int k, int max_checks, int sorted,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances, int cap) {
    extern __shared__ int s_heap[];   // per wave: cap distances, cap offsets
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const int qi = blockIdx.x * kKmWaves + wv;
    if (qi >= nq) return;
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    BranchHeapLds bh{s_heap + (size_t)wv * 2 * cap, reinterpret_cast<unsigned int*>(s_heap + (size_t)wv * 2 * cap + cap), 0, cap};
    bh.push(0, 0u, lane);
    int nchecks = 0, ncand = 0;
    while (nchecks < max_checks && bh.size != 0) {
        unsigned int off = bh.pop();
        int n;
        for (;;) {
            const uint8_t* blk = blob + off;
            const unsigned int hdr = *reinterpret_cast<const unsigned int*>(blk);          // u16 n | u8 isLeaf | pad
            const unsigned int hs = *reinterpret_cast<const unsigned int*>(blk + 4);
            n = (int)(hdr & 0xffffu);
            const bool leaf = ((hdr >> 16) & 0xffu) != 0;
            const bool on = lane < n;
            uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
            unsigned int info = 0;
            if (on) {   // blocks are 8-byte aligned: read the feature 8 bytes at a time
                const uint2* f2 = reinterpret_cast<const uint2*>(blk + hs + 32 * (size_t)lane);
                const uint2 w0 = f2[0], w1 = f2[1], w2 = f2[2], w3 = f2[3];
                a0 = make_uint4(w0.x, w0.y, w1.x, w1.y); a1 = make_uint4(w2.x, w2.y, w3.x, w3.y);
                info = reinterpret_cast<const unsigned int*>(blk + 8 + 8 * (size_t)lane)[0];   // low word: child offset / row index
            }
            const int dl = hamming256(a0, a1, q);
            if (leaf) {
                uint64_t* none = nullptr;
                feed_step<false>(h, dl, (int)info, on, k, -1, none, ncand, 0);
                break;
            }
            int bestd = 0x7fffffff, besto = -1;
            for (int c = 0; c < n; c++) {
                const int dc = rl(dl, c);
                const int oc = rl((int)info, c);
                if (dc < bestd) {
                    if (besto != -1) bh.push(bestd, (unsigned int)besto, lane);
                    bestd = dc; besto = oc;
                } else bh.push(dc, (unsigned int)oc, lane);
            }
            off = (unsigned int)besto;
        }
        nchecks += n;
    }
    finish_row(h, k, sorted, indices, distances, qi);
}

// Measurement aid (scripts/knn_push_bench.py): cycles of N always-accepted pushes on one wave, with and without the
// surrounding feed_step loop, to separate the heap update from the scan/branch overhead around it.
__global__ void knn_push_bench_kernel(int k, int n, long long* out) {
    const int lane = threadIdx.x & 63;
    WaveHeap h{0, -1, 0, lane};
    for (int i = 0; i < k; i++) h.push_accepted(100000 - i, i, k);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) h.push_accepted(90000 - i, k + i, k);      // strictly decreasing: always accepted
    long long t1 = __builtin_readcyclecounter();
    int dummy = 0;
    for (int i = 0; i < n; i++) {                                         // same pushes through feed_step (lane 0 carries the candidate)
        const int d = lane == (i & 63) ? 80000 - i : 0x7ffffff0;
        feed_step<false>(h, d, i, true, k, -1, nullptr, dummy, 0);
    }
    long long t2 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) h.push_accepted_scalar(70000 - i, k + i, k);
    long long t3 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = h.hd + h.hi; }
}

}  // namespace


// ------------------------------------------------------------------------------------------------ hierarchical k-means build (host)
// xflann::Index::build(features, HKMeansParams(k, maxIters)) = KMeansIndexCreator::build + convert (impl/kmeansindexcreator.{h,cpp}),
// producing the reference's own serialised block data (8-byte block header {u16 n, u8 isLeaf, u32 header_size}, n 8-byte node
// infos {child block offset | 1<<63 + row index}, n 32-byte features; blocks laid out breadth first).  It is built on the
// host like the reference does — per train frame, ~0.1 ms for 2000 rows: std::shuffle of a default std::mt19937 decides the
// centres, so the libstdc++ call itself is the specification — and searched on the GPU.
namespace {

struct KmNode {   // breadth-first order == the block order of KMeansIndexCreator::convert
    uint32_t begin = 0, count = 0;   // its rows: lvl_rows[level][begin, begin + count)
    int level = 0;
    int first_child = -1, n_children = 0;
    uint32_t centre = 0;             // the feature shown in the parent's block: a train row, or (>= n) extra centre number centre - n (a bitwise majority)
};

inline int host_hamming32(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

// ---- std::shuffle(first, last, std::mt19937()) as a position permutation.  The reference default-constructs the generator for every
// node it splits (kmeansindexcreator.cpp:44-), so the permutation depends on the node's SIZE only, and libstdc++'s loop walks the
// positions front to back (one draw per pair of positions below 65536 elements, an extra first draw for even sizes): the swaps of a size
// are a prefix of the swaps of every larger size of the same parity.  They are RECORDED from the library call itself (a value type whose
// ADL swap logs its operands) once per parity and replayed per node — no generator construction (624-word seeding + twist) per node.
struct KmSwapRec { uint32_t id; };
thread_local std::vector<uint32_t>* g_km_swaplog = nullptr;
thread_local const KmSwapRec* g_km_swapbase = nullptr;
inline void swap(KmSwapRec& a, KmSwapRec& b) noexcept {
    g_km_swaplog->push_back((uint32_t)(&a - g_km_swapbase));
    g_km_swaplog->push_back((uint32_t)(&b - g_km_swapbase));
    const uint32_t t = a.id; a.id = b.id; b.id = t;
}
constexpr int kKmPairLimit = 65535;   // libstdc++: two positions per draw while n * n <= the generator's range (2^32 - 1)
struct KmShuffleTables {
    std::mutex mu;
    std::vector<uint32_t> tgt[2];   // by parity of the size: tgt[i] = the position element i is swapped with (step i, i >= 1)
    std::vector<uint32_t> root_perm; int root_n = -1;   // the last root's permutation
    void ensure(int m) {            // caller holds mu
        std::vector<uint32_t>& t = tgt[m & 1];
        if ((int)t.size() >= m) return;
        int want = std::max(m, 4096);
        want = std::min(want + want / 2, kKmPairLimit);
        if ((want & 1) != (m & 1)) want -= 1;
        if (want < m) want = m;
        std::vector<KmSwapRec> v(want);
        for (int i = 0; i < want; i++) v[i].id = (uint32_t)i;
        std::vector<uint32_t> log;
        log.reserve(2 * (size_t)want);
        g_km_swaplog = &log; g_km_swapbase = v.data();
        std::mt19937 gen;
        std::shuffle(v.begin(), v.end(), gen);
        g_km_swaplog = nullptr; g_km_swapbase = nullptr;
        t.assign(want, 0);
        for (size_t q = 0; q + 1 < log.size(); q += 2) t[log[q]] = log[q + 1];   // (iter_swap(i, first + pos): the first operand is step i)
    }
};
KmShuffleTables& km_tables() { static KmShuffleTables t; return t; }

// The assignment step's operands, packed by the builder where the step reads them (host memory; pinned and device-visible behind
// uh_knn_build_kmeans): for every row of every node being split on this level its row number and the node's slot, per slot the number of
// centres and their 32-byte features.  The step fills `cluster`.
struct KmAssignIO {
    size_t npos = 0, nslots = 0;
    uint32_t* pos_row = nullptr; uint16_t* pos_slot = nullptr; uint8_t* slot_nc = nullptr; uint8_t* centres = nullptr;   // centres: slot * k * 32
    const uint8_t* cluster = nullptr;
};
struct KmAssigner {
    virtual ~KmAssigner() {}
    virtual int reserve(size_t npos, size_t nslots, int k, KmAssignIO& io) = 0;   // buffers for a level (contents need not survive the next reserve)
    virtual int run(const uint8_t* rows, int k, KmAssignIO& io) = 0;             // nearest centre, first minimum ("an exact hit ends the scan" picks the same one)
};
struct KmAssignerHost : KmAssigner {
    std::vector<uint8_t> buf, out;
    int reserve(size_t npos, size_t nslots, int k, KmAssignIO& io) override {
        auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
        const size_t o_row = 0, o_slot = al(4 * npos), o_nc = al(o_slot + 2 * npos), o_cen = al(o_nc + nslots), total = o_cen + nslots * (size_t)k * 32;
        buf.resize(total); out.resize(npos);
        io.npos = npos; io.nslots = nslots;
        io.pos_row = reinterpret_cast<uint32_t*>(buf.data() + o_row); io.pos_slot = reinterpret_cast<uint16_t*>(buf.data() + o_slot);
        io.slot_nc = buf.data() + o_nc; io.centres = buf.data() + o_cen; io.cluster = out.data();
        return UH_OK;
    }
    int run(const uint8_t* rows, int k, KmAssignIO& io) override {
        for (size_t p = 0; p < io.npos; p++) {
            const unsigned slot = io.pos_slot[p];
            const int nc = io.slot_nc[slot];
            const uint8_t* cen = io.centres + (size_t)slot * k * 32;
            int best = 0, bestd = 0x7fffffff;
            for (int c = 0; c < nc; c++) {
                const int d = host_hamming32(cen + 32 * (size_t)c, rows + 32 * (size_t)io.pos_row[p]);
                if (d < bestd) { bestd = d; best = c; }
                if (bestd == 0) break;
            }
            out[p] = (uint8_t)best;
        }
        return UH_OK;
    }
};

// The assignment step of the build on the device: one thread per row of a node being split (positions of all the level's nodes
// concatenated) — Hamming distance to each of its node's <= k centres, first minimum.  Inputs are read where the host packed them
// (pinned, device-visible: 6 bytes per row + the centres), the cluster numbers go back the same way; the train rows are in HBM.
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const uint8_t* __restrict__ rows, const uint32_t* __restrict__ pos_row, const uint16_t* __restrict__ pos_slot,
                                                            const uint8_t* __restrict__ centres, const uint8_t* __restrict__ slot_nc, int k, int npos,
                                                            uint8_t* __restrict__ out) {
    // The centres live in pinned HOST memory (un-cached on the device side): read once per workgroup into LDS — a workgroup's 256 consecutive
    // positions belong to a few consecutive slots — instead of k x 32 bytes over the host link per THREAD (round 6, first form: 24 us per launch,
    // ~10 MB of link reads for 10 000 rows).  More than kSlotsLds slots in one workgroup (a level of tiny nodes): those threads read in place.
    constexpr int kSlotsLds = 16;
    __shared__ __attribute__((aligned(16))) uint4 s_cen[kSlotsLds * 64 * 2];   // k <= 64 centres of 32 bytes per slot
    const int p0 = blockIdx.x * blockDim.x, p = p0 + threadIdx.x;
    const unsigned sl0 = pos_slot[p0], sl1 = pos_slot[min(p0 + (int)blockDim.x, npos) - 1];
    const int nsl = min((int)(sl1 - sl0) + 1, kSlotsLds);
    {
        const uint4* src = reinterpret_cast<const uint4*>(centres + (size_t)sl0 * k * 32);
        for (int i = threadIdx.x; i < nsl * k * 2; i += blockDim.x) s_cen[i] = src[i];
    }
    __syncthreads();
    if (p >= npos) return;
    const uint32_t r = pos_row[p];
    const unsigned slot = pos_slot[p];
    const int nc = slot_nc[slot];
    const uint4* rp = reinterpret_cast<const uint4*>(rows + 32 * (size_t)r);
    const uint4 a = rp[0], b = rp[1];
    const bool in_lds = slot - sl0 < (unsigned)nsl;
    const uint4* cl = s_cen + (size_t)(in_lds ? slot - sl0 : 0u) * k * 2;
    const uint4* cg = reinterpret_cast<const uint4*>(centres + (size_t)slot * k * 32);
    int best = 0, bestd = 0x7fffffff;
    for (int c = 0; c < nc; c++) {
        const uint4 x = in_lds ? cl[2 * c] : cg[2 * c], y = in_lds ? cl[2 * c + 1] : cg[2 * c + 1];
        const int d = __popc(a.x ^ x.x) + __popc(a.y ^ x.y) + __popc(a.z ^ x.z) + __popc(a.w ^ x.w) + __popc(b.x ^ y.x) + __popc(b.y ^ y.y) + __popc(b.z ^ y.z) +
                      __popc(b.w ^ y.w);
        if (d < bestd) { bestd = d; best = c; }
    }
    out[p] = (uint8_t)best;
}

// returns UH_OK, or UH_EINVAL with the error text set.  depth_out: levels of internal blocks above the deepest leaf block.
// Level by level (the reference recurses; its convert() lays the blocks out breadth first, which is the order nodes are appended in
// here): every node of a level is shuffled and given its centres on the host, then ONE assignment step serves all of them (the device
// kernel behind uh_knn_build_kmeans, KmAssignerHost behind the host-only hook) — the distances are where the build's time goes.  The
// rows of a level live in one flat array grouped by node; a node's clusters are a stable counting sort of its shuffled rows.
// blob_alloc(total) returns where the block data goes (total bytes, written completely).
int kmeans_build_blob(const uint8_t* rows, int n, int k, int max_iters, const std::function<uint8_t*(size_t)>& blob_alloc, int& depth_out, KmAssigner& asg) {
    std::vector<KmNode> nodes(1);
    std::vector<std::vector<uint32_t>> lvl_rows(1);
    std::vector<uint8_t> extra;   // majority centres (k-means rounds), 32 bytes each
    lvl_rows[0].resize(n);
    for (int i = 0; i < n; i++) lvl_rows[0][i] = (uint32_t)i;
    nodes[0].count = (uint32_t)n;
    depth_out = 0;
    std::vector<uint32_t> perm, centres;
    struct Split { int node; size_t pos0; uint32_t count; int nc; size_t cen0; bool active; size_t prev_hash, cur_hash, niters; };
    size_t lvl_b = 0, lvl_e = 1;
    for (int level = 0; lvl_b < lvl_e; level++) {
        std::vector<Split> splits;
        size_t npos = 0;
        for (size_t cur = lvl_b; cur < lvl_e; cur++) {
            if (cur != 0 && (int)nodes[cur].count <= k) continue;   // leaf (the root is always split)
            if (level > 64) {
                uh::set_error("uh_knn_build_kmeans: the tree does not stop splitting (a cluster keeps collapsing into one child); the reference does not terminate on this input");
                return UH_EINVAL;
            }
            splits.push_back(Split{(int)cur, npos, nodes[cur].count, 0, 0, true, 0, 1, 0});
            npos += nodes[cur].count;
        }
        if (splits.empty()) break;
        KmAssignIO io;
        int rc = asg.reserve(npos, splits.size(), k, io);
        if (rc) return rc;
        UH_REQUIRE(splits.size() <= 65535, "uh_knn_build_kmeans: %zu nodes split on one level", splits.size());
        const std::vector<uint32_t>& cur_rows = lvl_rows[level];
        centres.clear();
        for (size_t si = 0; si < splits.size(); si++) {
            Split& sp = splits[si];
            const KmNode& nd = nodes[sp.node];
            uint32_t* dst = io.pos_row + sp.pos0;
            const uint32_t* src = cur_rows.data() + nd.begin;
            const int m = (int)nd.count;
            // std::shuffle(rows, std::mt19937()) through the recorded swaps
            if (m > kKmPairLimit) { std::memcpy(dst, src, 4 * (size_t)m); std::mt19937 gen; std::shuffle(dst, dst + m, gen); }
            else if (m < 2) { if (m) dst[0] = src[0]; }
            else {
                KmShuffleTables& T = km_tables();
                std::lock_guard<std::mutex> lk(T.mu);
                if (sp.node == 0 && T.root_n == m) std::memcpy(dst, T.root_perm.data(), 4 * (size_t)m);   // (the root's rows are 0 .. n-1: its shuffle IS the permutation; train sets repeat their size)
                else {
                    T.ensure(m);
                    perm.resize(m);
                    const uint32_t* t = T.tgt[m & 1].data();
                    uint32_t* p = perm.data();
                    for (int i = 0; i < m; i++) p[i] = (uint32_t)i;
                    for (int i = 1; i < m; i++) { const uint32_t j = t[i]; const uint32_t a = p[i]; p[i] = p[j]; p[j] = a; }
                    for (int i = 0; i < m; i++) dst[i] = src[p[i]];
                    if (sp.node == 0) { T.root_perm.assign(dst, dst + m); T.root_n = m; }
                }
            }
            for (int i = 0; i < m; i++) io.pos_slot[sp.pos0 + i] = (uint16_t)si;
            sp.cen0 = centres.size();
            for (int next = 0; next < m && sp.nc < k; next++) {   // first k mutually distinct rows
                bool dup = false;
                for (int c = 0; c < sp.nc; c++) if (host_hamming32(rows + 32 * (size_t)dst[next], rows + 32 * (size_t)centres[sp.cen0 + c]) == 0) { dup = true; break; }
                if (!dup) { centres.push_back(dst[next]); sp.nc++; }
            }
            io.slot_nc[si] = (uint8_t)sp.nc;
            for (int c = 0; c < sp.nc; c++) std::memcpy(io.centres + (si * (size_t)k + c) * 32, rows + 32 * (size_t)centres[sp.cen0 + c], 32);
        }
        if ((rc = asg.run(rows, k, io))) return rc;
        // clusters -> the next level's flat array: per node a stable counting sort of its shuffled rows (the order push_back gives the reference)
        lvl_rows.emplace_back(npos);
        std::vector<uint32_t>& next_rows = lvl_rows.back();
        std::vector<uint32_t> cnt((size_t)k * splits.size());
        auto sort_clusters = [&](const Split& sp, size_t si) {
            uint32_t* c = cnt.data() + si * (size_t)k;
            std::fill(c, c + k, 0u);
            const uint8_t* cl = io.cluster + sp.pos0;
            for (uint32_t i = 0; i < sp.count; i++) c[cl[i]]++;
            uint32_t off[64];
            uint32_t o = (uint32_t)sp.pos0;
            for (int q = 0; q < sp.nc; q++) { off[q] = o; o += c[q]; }
            const uint32_t* src = io.pos_row + sp.pos0;
            for (uint32_t i = 0; i < sp.count; i++) next_rows[off[cl[i]]++] = src[i];
        };
        for (size_t si = 0; si < splits.size(); si++) sort_clusters(splits[si], si);
        // k-means rounds (HKMeansParams maxIters; -1 = until the assignment hash repeats): centres move to the bitwise majority of
        // their clusters (kmeansindexcreator.h:245-262, 390-422); every node runs its own number of rounds
        std::vector<uint8_t> majority;   // per split: k x 32 bytes, valid once a round has run for it
        if (max_iters != 0) majority.assign(splits.size() * (size_t)k * 32, 0);
        std::vector<char> has_majority(splits.size(), 0);
        for (;;) {
            bool any = false;
            for (size_t si = 0; si < splits.size(); si++) {
                Split& sp = splits[si];
                if (!sp.active) continue;
                if (!(sp.cur_hash != sp.prev_hash && (max_iters == -1 || sp.niters++ < (size_t)max_iters))) { sp.active = false; continue; }
                std::swap(sp.prev_hash, sp.cur_hash);
                const uint32_t* c = cnt.data() + si * (size_t)k;
                uint32_t o = (uint32_t)sp.pos0;
                for (int q = 0; q < sp.nc; q++) {
                    int sum[256] = {0};
                    const uint32_t cn = c[q] ? c[q] : 1u;   // an empty cluster is given its centre row back (:390-)
                    for (uint32_t i = 0; i < cn; i++) {
                        const uint8_t* pr = rows + 32 * (size_t)(c[q] ? next_rows[o + i] : centres[sp.cen0 + q]);
                        for (int j = 0; j < 32; j++)
                            for (int bit = 0; bit < 8; bit++) if (pr[j] & (128 >> bit)) ++sum[j * 8 + bit];
                    }
                    o += c[q];
                    const int half = (int)cn / 2 + (int)(cn % 2);
                    uint8_t* ce = majority.data() + (si * (size_t)k + q) * 32;
                    std::memset(ce, 0, 32);
                    for (int i = 0; i < 256; i++) if (sum[i] >= half) ce[i / 8] |= (uint8_t)(1 << (7 - (i % 8)));
                    std::memcpy(io.centres + (si * (size_t)k + q) * 32, ce, 32);
                }
                has_majority[si] = 1;
                any = true;
            }
            if (!any) break;
            // (one step over the whole level: a node that is no longer active gets the assignment it already has — its centres did not move)
            if ((rc = asg.run(rows, k, io))) return rc;
            for (size_t si = 0; si < splits.size(); si++) {
                Split& sp = splits[si];
                if (!sp.active) continue;
                sort_clusters(sp, si);
                size_t seed = 0;
                for (uint32_t i = 0; i < sp.count; i++) { const uint32_t id = next_rows[sp.pos0 + i]; seed ^= id + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
                sp.cur_hash = seed;
            }
        }
        const size_t next_b = nodes.size();
        for (size_t si = 0; si < splits.size(); si++) {
            const Split& sp = splits[si];
            const uint32_t* c = cnt.data() + si * (size_t)k;
            int nkept = 0;
            for (int q = 0; q < sp.nc; q++) nkept += c[q] != 0;
            if (nkept == 1 && (int)sp.count > k) {
                bool same = true;
                const uint32_t* pr = io.pos_row + sp.pos0;
                for (uint32_t i = 0; i < sp.count; i++) if (host_hamming32(rows + 32 * (size_t)pr[i], rows + 32 * (size_t)pr[0]) != 0) { same = false; break; }
                if (same) {
                    uh::set_error("uh_knn_build_kmeans: more than k=%d identical descriptors: the reference's tree construction does not terminate on this input", k);
                    return UH_EINVAL;
                }
            }
            const int first = (int)nodes.size();
            uint32_t o = (uint32_t)sp.pos0;
            for (int q = 0; q < sp.nc; q++) {   // empty clusters are dropped (:268-271)
                if (c[q] == 0) continue;
                KmNode kd;
                kd.begin = o; kd.count = c[q]; kd.level = level + 1;
                if (has_majority[si]) {
                    kd.centre = (uint32_t)n + (uint32_t)(extra.size() / 32);
                    extra.insert(extra.end(), majority.data() + (si * (size_t)k + q) * 32, majority.data() + (si * (size_t)k + q + 1) * 32);
                } else kd.centre = centres[sp.cen0 + q];
                nodes.push_back(kd);
                o += c[q];
            }
            nodes[sp.node].first_child = first;
            nodes[sp.node].n_children = nkept;
            depth_out = std::max(depth_out, level + 1);
        }
        lvl_b = next_b;
        lvl_e = nodes.size();
    }
    auto pad8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    std::vector<uint64_t> off(nodes.size());
    uint64_t total = 0;
    for (size_t i = 0; i < nodes.size(); i++) {
        const size_t cnt = nodes[i].n_children ? (size_t)nodes[i].n_children : nodes[i].count;
        off[i] = total;
        total += pad8(8 + 8 * cnt) + 32 * cnt;
    }
    if (total >= (1ull << 31)) { uh::set_error("uh_knn_build_kmeans: index of %llu bytes exceeds the 31-bit block offsets of the reference's search", (unsigned long long)total); return UH_EINVAL; }
    uint8_t* const blob = blob_alloc((size_t)total);
    if (!blob) return UH_ENOMEM;
    for (size_t i = 0; i < nodes.size(); i++) {
        const KmNode& nd = nodes[i];
        const bool leaf = nd.n_children == 0;
        const uint32_t cnt = leaf ? nd.count : (uint32_t)nd.n_children;
        uint8_t* blk = blob + off[i];
        const uint32_t hs = (uint32_t)pad8(8 + 8 * (size_t)cnt);
        std::memset(blk, 0, hs);
        const uint16_t n16 = (uint16_t)cnt;
        std::memcpy(blk, &n16, 2);
        blk[2] = leaf ? 1 : 0;
        std::memcpy(blk + 4, &hs, 4);
        const uint32_t* lr = leaf ? lvl_rows[nd.level].data() + nd.begin : nullptr;
        for (uint32_t j = 0; j < cnt; j++) {
            const uint64_t info = leaf ? ((uint64_t)lr[j] | 0x8000000000000000ull) : off[nd.first_child + j];
            const uint32_t src = leaf ? lr[j] : nodes[nd.first_child + j].centre;
            const uint8_t* feat = src < (uint32_t)n ? rows + 32 * (size_t)src : extra.data() + 32 * (size_t)(src - (uint32_t)n);   // row, or majority centre
            std::memcpy(blk + 8 + 8 * (size_t)j, &info, 8);
            std::memcpy(blk + hs + 32 * (size_t)j, feat, 32);
        }
    }
    return UH_OK;
}

}  // namespace

struct uh_knn {
    uh_ctx* ctx = nullptr;
    uh::DevBuf train_store;       // owned copy (build from host)
    const uint8_t* d_train = nullptr;
    int nt = 0;
    int shard_begin = 0, shard_end = 0;
    const int32_t* d_valid_rows = nullptr;   // uh_knn_set_valid_rows_dev: the shard scan's query rows from *d_valid_rows on get empty lists
    int row_offset = 0;           // global index of row 0 (uh_knn_set_row_offset): an index that holds only one tile of a sharded train set
    int qpw = 1;                  // queries per wave of the exact search (uh_knn_set_queries_per_wave)
    // Exact search, nn <= 16, which kernels run (MI355X, 10 000 train rows; scripts/knn_forms.py): below 3000 queries the fused one-wave-per-query
    // search (2000 x nn 10: 73 us, the list forms ~75-90); from 3000 queries on the accept-list forms — nn >= 6: scan and replay in ONE launch
    // (knn_stream_kernel; 8000 x nn 10: 105 us against 126 as two launches and 154 fused), nn <= 5: two launches (the replay is short and the
    // plain scan faster than the streaming one: 8000 x nn 2: 73 against 81 us).  UH_KNN_FORM=fused | twophase | stream forces one form.
    int two_phase_min_nq = 3000;
    int stream_min_nn = 6;
    int stream_min_nq = 1;        // round 5: nn >= stream_min_nn takes the stream form at every size (2000 queries: 43 us against 74 fused, 64 queries: 33 against 61)
    bool stream_when_few = true;  // ... and EVERY nn <= 16 while the one-query-per-wave form applies (queries + replay workgroups <= 2 per SIMD, ~2000 queries):
                                  // nn 2: 20-34 us against 29-36 fused, nn 5: 24-37 against 42-52 (scripts/knn_nq_sweep.py, profiles/r05_knn_nq_sweep.txt)
    unsigned replay_attr = 0;     // bit k: knn_replay_lane_kernel<k>'s dynamic-LDS attribute has been set on this index's device
    int accept_qpw = 2;           // queries per wave of the accept scan (UH_KNN_ACCEPT_QPW=1 for the A/B)
    unsigned stream_tag = 0;      // launch tag of the streamed lists (0 = the value freshly cleared memory holds, never used)
    unsigned stream_tag0 = 0;     // UH_KNN_TAG0: first tag of a fresh buffer (test hook for the wrap)
    uh::DevBuf redo_buf;
    const void* stream_buf = nullptr;   // the list buffer the tags refer to (a new allocation is cleared and starts over)
    unsigned stream_gen = 0;            // ... and its allocation generation (hipFree + hipMalloc may hand back the same base address)
    uh::DevBuf list_buf;          // accept lists, counts and redo flags of the two-phase search
    uh::DevBuf q_buf, idx_buf, dist_buf;  // staging for the host-pointer API
    bool split_form = false;              // UH_KNN_FORM=split: the range-split form of small query sets (see uh_knn_search_dev)
    uh::MappedBuf h_word;                 // completion word of a host-pointer search with pinned buffers
    unsigned long long host_seq = 0;
    // hierarchical k-means form of the same index (uh_knn_build_kmeans)
    std::vector<uint8_t> km_blob;         // host copy of the block data (streams, uh_knn_kmeans_blob): filled from km_pin on demand after a device build
    uh::MappedBuf km_pin;                 // the build writes the block data here (pinned): [completion word | blocks]; a 16-byte-wide launch moves it to km_dev
    size_t km_size = 0;                   // bytes of block data
    bool km_blob_stale = false;           // km_blob has not been copied out of km_pin yet
    unsigned long long km_pin_word = 0;   // completion word of the launch that last read km_pin
    uh::DevBuf km_dev;
    uh::DevBuf km_rows;                   // the train rows of the build in HBM (the assignment kernel's operand)
    uh::MappedBuf km_rows_pin;            // ... on their way there when the caller's array is pageable
    uh::MappedBuf km_stage;               // [completion word | clusters out | row per position | slot per position | centres per slot | centres of a slot]
    unsigned long long km_seq = 0;
    double km_assign_us = 0; int km_assign_calls = 0;   // UH_KM_TIMING
    int km_k = 0, km_n = 0, km_depth = 0;
    bool km_attr = false;
};

// ancestor-walk depth of the lane-distributed heap by nn (see WaveHeap::push_accepted)
static void launch_replay(uh_knn* idx, dim3 grid, dim3 block, const ShardBounds& sb, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                          const uint64_t* d_cand, const int32_t* d_counts, int cap, int32_t* d_indices, int32_t* d_distances, int* d_overflow,
                          size_t cand_stride = 0, size_t count_stride = 0) {
    if (!cand_stride) cand_stride = (size_t)nq * cap;
    if (!count_stride) count_stride = (size_t)nq;
    if (nn <= 3) UH_LAUNCH(idx->ctx, knn_replay_kernel<1>, grid, block, 0, idx->d_train, sb, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_cand, d_counts, cap, d_indices, d_distances, d_overflow, cand_stride, count_stride);
    else if (nn <= 15) UH_LAUNCH(idx->ctx, knn_replay_kernel<3>, grid, block, 0, idx->d_train, sb, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_cand, d_counts, cap, d_indices, d_distances, d_overflow, cand_stride, count_stride);
    else UH_LAUNCH(idx->ctx, knn_replay_kernel<6>, grid, block, 0, idx->d_train, sb, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_cand, d_counts, cap, d_indices, d_distances, d_overflow, cand_stride, count_stride);
}

// tile lists -> rows with one lane per query (k <= 16, rows below 2^23, no rescan); false: the caller uses the wave-per-query replay
static bool launch_replay_lanes(uh_knn* idx, int nq, int nn, int sorted, int max_dist, const uint64_t* d_cand, size_t cand_stride, const int32_t* d_counts,
                                size_t count_stride, int nshards, int cap, int32_t* d_indices, int32_t* d_distances, int* d_overflow,
                                int* d_redo_list = nullptr, int* d_redo_count = nullptr) {
    // One lane per query steps through EVERY listed row in lockstep over 64 queries, the wave-per-query replay pays per ACCEPTED push: with
    // one list (world 1: 70 rows listed, 79 accepted overall) the lane form wins (42 vs 51 us for 2000 queries), with the 8 lists of 8 tiles
    // (425 rows listed per query) it loses (158 vs 52 us; scripts/time_shard_replay.py).  UH_KNN_SHARD_FORM=lanes / old force one for the A/B.
    const char* form = getenv("UH_KNN_SHARD_FORM");
    const bool force_lanes = form && std::string(form) == "lanes";
    if (nn > kRpK || cap > 256 || (form && !force_lanes) || (nshards > 1 && !force_lanes && !d_redo_list)) return false;
    if (!cand_stride) cand_stride = (size_t)nq * cap;
    if (!count_stride) count_stride = (size_t)nq;
    const dim3 grid(uh_div_up(nq, kWave)), block(kWave);
    const size_t lds = (size_t)cap * kWave * 8;
#define UH_KNN_RPS(K) case K: { static bool attr##K = false; if (!attr##K) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_replay_shards_lane_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * kWave * 8); attr##K = true; } \
        UH_LAUNCH(idx->ctx, knn_replay_shards_lane_kernel<K>, grid, block, lds, d_cand, cand_stride, d_counts, count_stride, nshards, nq, sorted ? 1 : 0, max_dist, cap, d_indices, d_distances, d_overflow, d_redo_list, d_redo_count); } break
    switch (nn) {
        UH_KNN_RPS(1); UH_KNN_RPS(2); UH_KNN_RPS(3); UH_KNN_RPS(4); UH_KNN_RPS(5); UH_KNN_RPS(6); UH_KNN_RPS(7); UH_KNN_RPS(8);
        UH_KNN_RPS(9); UH_KNN_RPS(10); UH_KNN_RPS(11); UH_KNN_RPS(12); UH_KNN_RPS(13); UH_KNN_RPS(14); UH_KNN_RPS(15); UH_KNN_RPS(16);
        default: return false;
    }
#undef UH_KNN_RPS
    return true;
}

// compute units the context's stream can really use (a CU-masked context reports its share, ADVICE r5: the replay workgroups of the stream
// form barrier on each other inside the launch and must all be resident — the capacity tests must not assume more CUs than the mask gives)
static inline int km_cus(const uh_ctx* c) { return c->num_cus > 0 ? c->num_cus : 64; }

extern "C" {

int uh_knn_create(uh_ctx* ctx, uh_knn** out) {
    UH_REQUIRE(ctx && out, "uh_knn_create: NULL argument");
    uh_knn* k = new uh_knn();
    k->ctx = ctx;
    if (const char* e = getenv("UH_KNN_FORM")) {
        const std::string f(e);
        k->split_form = f == "split";
        if (f == "fused") { k->two_phase_min_nq = k->stream_min_nq = 0x7fffffff; k->stream_when_few = false; }
        else if (f == "twophase") { k->two_phase_min_nq = 0; k->stream_min_nn = 0x7fffffff; k->stream_when_few = false; }
        else if (f == "stream") { k->two_phase_min_nq = 0; k->stream_min_nn = 0; }
    }
    if (const char* f = getenv("UH_KNN_ACCEPT_QPW")) k->accept_qpw = atoi(f) == 1 ? 1 : 2;
    if (const char* f = getenv("UH_KNN_TAG0")) k->stream_tag0 = (unsigned)strtoul(f, nullptr, 0);
    *out = k;
    return UH_OK;
}

void uh_knn_destroy(uh_knn* idx) { delete idx; }

int uh_knn_build(uh_knn* idx, const uint8_t* train, int nt, size_t stride, int desc_bytes) {
    UH_REQUIRE(idx, "uh_knn_build: NULL index");
    UH_REQUIRE(desc_bytes == 32, "uh_knn_build: only 32-byte (ORB, 256-bit) descriptors are supported, got %d", desc_bytes);
    idx->d_train = nullptr;
    idx->nt = 0;
    idx->shard_begin = idx->shard_end = 0;
    if (nt <= 0) return UH_OK;  // index.cpp:49 — empty features leave the index unbuilt
    UH_REQUIRE(train != nullptr, "uh_knn_build: NULL train pointer");
    UH_REQUIRE(stride >= 32, "uh_knn_build: row stride %zu < 32", stride);
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    int rc = idx->train_store.reserve((size_t)nt * 32);
    if (rc) return rc;
    UH_HIP_CHECK(hipMemcpy2DAsync(idx->train_store.p, 32, train, stride, 32, (size_t)nt, hipMemcpyHostToDevice,
                                  idx->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(idx->ctx->stream));
    idx->d_train = idx->train_store.as<uint8_t>();
    idx->nt = nt;
    idx->shard_begin = 0;
    idx->shard_end = nt;
    return UH_OK;
}

int uh_knn_build_dev(uh_knn* idx, const uint8_t* d_train, int nt) {
    UH_REQUIRE(idx, "uh_knn_build_dev: NULL index");
    idx->d_train = nullptr;
    idx->nt = 0;
    idx->shard_begin = idx->shard_end = 0;
    if (nt <= 0) return UH_OK;
    UH_REQUIRE(d_train != nullptr, "uh_knn_build_dev: NULL train pointer");
    UH_REQUIRE((reinterpret_cast<uintptr_t>(d_train) & 15) == 0, "uh_knn_build_dev: train rows must be 16-byte aligned");
    idx->d_train = d_train;
    idx->nt = nt;
    idx->shard_end = nt;
    return UH_OK;
}

int uh_knn_set_shard(uh_knn* idx, int begin, int end) {
    UH_REQUIRE(idx, "uh_knn_set_shard: NULL index");
    UH_REQUIRE(0 <= begin && begin <= end && end <= idx->nt, "uh_knn_set_shard: [%d,%d) outside [0,%d)", begin, end, idx->nt);
    idx->shard_begin = begin;
    idx->shard_end = end;
    return UH_OK;
}

int uh_knn_size(const uh_knn* idx) { return idx ? idx->nt : 0; }

int uh_knn_set_queries_per_wave(uh_knn* idx, int qpw) {
    UH_REQUIRE(idx, "uh_knn_set_queries_per_wave: NULL index");
    UH_REQUIRE(qpw == 1 || qpw == 2 || qpw == 4, "uh_knn_set_queries_per_wave: %d (supported: 1, 2, 4)", qpw);
    idx->qpw = qpw;
    return UH_OK;
}

int uh_knn_debug_push_cycles(uh_knn* idx, int k, int n, long long* out3) {
    long long* d = nullptr;
    UH_HIP_CHECK(hipMalloc(&d, 64));
    UH_LAUNCH(idx->ctx, knn_push_bench_kernel, dim3(1), dim3(64), 0, k, n, d);
    UH_HIP_CHECK(hipMemcpy(out3, d, 32, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return UH_OK;
}

static int check_search_args(const uh_knn* idx, const void* q, int nq, int nn, const void* i, const void* d) {
    UH_REQUIRE(idx, "uh_knn_search: NULL index");
    if (idx->d_train == nullptr) {
        uh::set_error("uh_knn_search: could not run search because index not created");  // index.cpp:82-85
        return UH_ENOTBUILT;
    }
    UH_REQUIRE(nq >= 0, "uh_knn_search: negative query count");
    UH_REQUIRE(nn >= 1 && nn <= kWave, "uh_knn_search: nn=%d outside [1,%d]", nn, kWave);
    if (nq > 0) UH_REQUIRE(q && i && d, "uh_knn_search: NULL buffer");
    return UH_OK;
}

// ho (host form): where the stream form hands the rows to the host itself; *handed is set when it did (the other forms leave that to the caller)
static int knn_search_dev_impl(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int32_t* d_indices,
                               int32_t* d_distances, int sorted, int max_dist, KnnHostOut ho, bool* handed) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_indices, d_distances);
    if (rc) return rc;
    if (nq == 0) return UH_OK;
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    // queries per wave (uh_knn_set_queries_per_wave): 2 or 4 queries share every train row a wave loads — the search itself gets
    // slower (the pushes of a wave's queries serialise), its L1/L2 traffic and its resident waves drop to 1/2 or 1/4, which is what a
    // latency-bound neighbour on another stream (the local BA) needs
    const int qpw = nn <= 15 ? idx->qpw : 1;
    const bool few_form = uh_div_up(nq, kWave) + nq <= 2 * 4 * km_cus(idx->ctx);
    // one query per scan wave at up to four waves per SIMD (~4000 queries); nn <= 5: up to three (at 4000 queries the two launches win: nn 2 51 us against 58)
    const bool few_any = uh_div_up(nq, kWave) + nq <= (nn >= idx->stream_min_nn ? 4 : 3) * 4 * km_cus(idx->ctx);   // at most two one-wave workgroups per SIMD with ONE query per scan wave (knn_stream_kernel<K, 2>)
    if (nn <= kRpK && (nq >= idx->two_phase_min_nq || (nn >= idx->stream_min_nn && nq >= idx->stream_min_nq) || (few_any && idx->stream_when_few)) && idx->shard_end <= (1 << 23)) {   // (the replay packs distance and row index into 32 bits)
        // accept-list capacity: the expected number of accepted pushes is k (1 + ln(N / k)) (a record process), its spread ~ sqrt of
        // that; lists that still overflow (distances descending with the row index) are redone by knn_redo_kernel
        const int nrows = std::max(idx->shard_end - idx->shard_begin, 1);
        const double expect = nn * (1.0 + std::log(std::max((double)nrows / nn, 1.0)));
        const int cap = std::min(std::min(std::max(((int)(1.6 * expect) + 32 + 31) & ~31, 32), std::max((nrows + 31) & ~31, 32)), 256);
        int rc;
        if ((rc = idx->list_buf.reserve((size_t)nq * cap * 8 + (size_t)nq * 16 + 256))) return rc;
        // the replay workgroups of the stream form wait for each other at the end (the shared redo, the hand-over to the host): ALL of them must be
        // resident at once — they are the first workgroups of the grid, and they are kept to at most two per compute unit (32 768 queries on
        // MI355X; a 20th of the one-wave slots); larger batches take the two launches
        const bool stream_fits = uh_div_up(nq, kWave) <= 2 * km_cus(idx->ctx);
        if (stream_fits && cap >= kWave /* (the vector first step writes a whole step before the list-full test: ADVICE r5) */ && (nn >= idx->stream_min_nn || (few_any && idx->stream_when_few)) && idx->accept_qpw == 2 && (size_t)nq * cap * 8 < ((size_t)1 << 31)) {   // (record offsets are 32-bit buffer offsets)
            uint64_t* d_cand = idx->list_buf.as<uint64_t>();
            uint64_t* d_prog = d_cand + (size_t)nq * cap;          // (list_buf holds tagged words only: any layout of an earlier launch is harmless)
            const unsigned had = idx->redo_buf.gen;
            if ((rc = idx->redo_buf.reserve((size_t)(nq + 8) * 4))) return rc;
            int* d_nredo = idx->redo_buf.as<int>();                // [parity]: queries to redo, [2 / 4 / 6 + parity]: replay workgroups through their lists / redo share / host copy; then the compacted list of the queries to redo
            int* d_redo = d_nredo + 8;
            if (had != idx->redo_buf.gen) UH_HIP_CHECK(hipMemsetAsync(d_nredo, 0, 32, idx->ctx->stream));
            const bool same_mem = idx->stream_buf == idx->list_buf.p && idx->stream_gen == idx->list_buf.gen;
            if (!same_mem || idx->stream_tag == 0xFFFFFFFFu) {   // new memory, or the tag wraps: no stale word may match
                UH_HIP_CHECK(hipMemsetAsync(idx->list_buf.p, 0, idx->list_buf.cap, idx->ctx->stream));
                idx->stream_tag = same_mem ? 0 : idx->stream_tag0;
                idx->stream_buf = idx->list_buf.p;
                idx->stream_gen = idx->list_buf.gen;
                // the overflow counters are picked by tag parity and a launch clears only the NEXT launch's slot: after a tag reset the slot
                // of the coming launch may still hold the count an earlier launch of the same parity left behind (stale query ids would
                // be replayed by knn_redo_kernel) — clear both
                UH_HIP_CHECK(hipMemsetAsync(d_nredo, 0, 32, idx->ctx->stream));
            }
            const unsigned tag = ++idx->stream_tag;
            static const long long stream_timeout = [] { const char* e = getenv("UH_KNN_STREAM_TIMEOUT_MS"); const long long ms = e ? atoll(e) : 0; return ms > 0 ? ms * 100000ll : kStreamTimeoutDefault; }();
            const int nrep = uh_div_up(nq, kWave);
            const bool few = few_form;
            const bool few3 = !few && nrep + nq <= 3 * 4 * km_cus(idx->ctx);   // one query per scan wave at three waves per SIMD (2000 < queries <= ~3000; 170 registers)
            const bool few4 = !few && !few3 && nrep + nq <= 4 * 4 * km_cus(idx->ctx);   // ... at four (two row groups in rotation: 128 registers)
            const dim3 gs(nrep + (few || few3 || few4 ? nq : uh_div_up(nq, 2)));
#define UH_KNN_STREAM_W(K, W) UH_LAUNCH(idx->ctx, (knn_stream_kernel<K, W>), gs, dim3(kWave), 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, sorted ? 1 : 0, max_dist, \
            d_cand, d_prog, cap, tag, nrep, d_indices, d_distances, d_redo, d_nredo + (tag & 1u), d_nredo + ((tag + 1u) & 1u), stream_timeout, ho)
#define UH_KNN_STREAM(K) case K: if (few) UH_KNN_STREAM_W(K, 2); else if (few3) UH_KNN_STREAM_W(K, 3); else if (few4) UH_KNN_STREAM_W(K, 4); else UH_KNN_STREAM_W(K, 5); break
            switch (nn) {
                UH_KNN_STREAM(1); UH_KNN_STREAM(2); UH_KNN_STREAM(3); UH_KNN_STREAM(4); UH_KNN_STREAM(5); UH_KNN_STREAM(6); UH_KNN_STREAM(7); UH_KNN_STREAM(8);
                UH_KNN_STREAM(9); UH_KNN_STREAM(10); UH_KNN_STREAM(11); UH_KNN_STREAM(12); UH_KNN_STREAM(13); UH_KNN_STREAM(14); UH_KNN_STREAM(15); UH_KNN_STREAM(16);
                default: break;
            }
#undef UH_KNN_STREAM
#undef UH_KNN_STREAM_W
            UH_HIP_CHECK(hipGetLastError());
            if (handed) *handed = ho.host_done != nullptr;
            return UH_OK;
        }
        idx->stream_buf = nullptr;   // (the two-launch form writes untagged words into the same buffer)
        uint64_t* d_cand = idx->list_buf.as<uint64_t>();
        int32_t* d_counts = reinterpret_cast<int32_t*>(d_cand + (size_t)nq * cap);
        int* d_redo = d_counts + nq;       // [nq] compacted list of overflowed queries, [nq] their number
        int* d_nredo = d_redo + nq;
        const int aq = idx->accept_qpw;   // queries per wave of the accept scan (default 2: one is L1-bandwidth bound)
        const dim3 ga(uh_div_up(nq, kWavesPerBlock * aq));
#define UH_KNN_ACCEPT(Q) UH_LAUNCH(idx->ctx, knn_accept_kernel<Q>, ga, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, max_dist, d_cand, d_counts, cap, d_nredo)
        if (aq == 1) UH_KNN_ACCEPT(1); else UH_KNN_ACCEPT(2);
#undef UH_KNN_ACCEPT
        const dim3 gr(uh_div_up(nq, kWave));
        const size_t lds_lists = (size_t)cap * kWave * 8;   // cap <= 256: at most 128 KB
        // (the attribute is raised once per k to the largest list area the replay may ask for: 256 entries x 64 lanes x 8 bytes)
#define UH_KNN_REPLAY(K) case K: if (!(idx->replay_attr & (1u << K))) { UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_replay_lane_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * kWave * 8)); idx->replay_attr |= 1u << K; } \
        UH_LAUNCH(idx->ctx, knn_replay_lane_kernel<K>, gr, dim3(kWave), lds_lists, d_cand, d_counts, nq, sorted ? 1 : 0, max_dist, cap, d_indices, d_distances, d_redo, d_nredo); break
        switch (nn) {
            UH_KNN_REPLAY(1); UH_KNN_REPLAY(2); UH_KNN_REPLAY(3); UH_KNN_REPLAY(4); UH_KNN_REPLAY(5); UH_KNN_REPLAY(6); UH_KNN_REPLAY(7); UH_KNN_REPLAY(8);
            UH_KNN_REPLAY(9); UH_KNN_REPLAY(10); UH_KNN_REPLAY(11); UH_KNN_REPLAY(12); UH_KNN_REPLAY(13); UH_KNN_REPLAY(14); UH_KNN_REPLAY(15); UH_KNN_REPLAY(16);
            default: break;
        }
#undef UH_KNN_REPLAY
        const dim3 gd(64);
        if (nn <= 3) UH_LAUNCH(idx->ctx, knn_redo_kernel<1>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
        else if (nn <= 15) UH_LAUNCH(idx->ctx, knn_redo_kernel<3>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
        else UH_LAUNCH(idx->ctx, knn_redo_kernel<6>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
        UH_HIP_CHECK(hipGetLastError());
        return UH_OK;
    }
    // ---- UH_KNN_FORM=split (measured and NOT chosen): for ONE frame's queries the range is cut into S slices scanned by S times as many
    // waves (the threshold-only scan of the sharded search, exact lists per slice) and a lane-per-query replay walks the slices' lists in
    // row order.  The scan gets S times shorter, but every slice starts from an empty heap and lists 65 rows where the global heap accepts
    // 79 in all: the replay wades through 260 entries per query in lockstep over 64 queries — 2000 x 10 000, nn 10: 144 us against 77
    // fused / 73 as one stream-form launch.  Kept as the single-GPU rehearsal of the sharded search (same kernels).
    {
        const int nrows = idx->shard_end - idx->shard_begin;
        const bool no_split = !idx->split_form;
        int S = std::min(16, std::max(1, 8192 / std::max(nq, 1)));
        S = std::min(S, nrows / 512);
        if (!no_split && nn <= kRpK && S >= 2 && idx->shard_end <= (1 << 23) && qpw == 1) {
            const double rows_s = (double)nrows / S, expect = nn * (1.0 + std::log(std::max(rows_s / nn, 1.0)));
            const int cap = std::min(256, std::max(32, (((int)(expect + 4.0 * std::sqrt(expect)) + 15) & ~15)));
            int rc;
            if ((rc = idx->list_buf.reserve((size_t)S * nq * cap * 8 + (size_t)S * nq * 4 + (size_t)(nq + 2) * 4 + 256))) return rc;
            idx->stream_buf = nullptr;   // (untagged words go into the buffer the stream form keeps its tagged ones in)
            uint64_t* d_cand = idx->list_buf.as<uint64_t>();
            int32_t* d_counts = reinterpret_cast<int32_t*>(d_cand + (size_t)S * nq * cap);
            int* d_redo = d_counts + (size_t)S * nq;
            int* d_nredo = d_redo + nq;
            const dim3 gs(uh_div_up(uh_div_up(nq, 2), kWavesPerBlock), S);
            UH_LAUNCH(idx->ctx, knn_scan_shard2_kernel, gs, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, max_dist, d_cand, d_counts, cap, d_nredo, static_cast<const int32_t*>(nullptr));
            if (launch_replay_lanes(idx, nq, nn, sorted, max_dist, d_cand, 0, d_counts, 0, S, cap, d_indices, d_distances, nullptr, d_redo, d_nredo)) {
                const dim3 gd(64);
                if (nn <= 3) UH_LAUNCH(idx->ctx, knn_redo_kernel<1>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
                else if (nn <= 15) UH_LAUNCH(idx->ctx, knn_redo_kernel<3>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
                else UH_LAUNCH(idx->ctx, knn_redo_kernel<6>, gd, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances, (const int*)d_redo, (const int*)d_nredo);
                UH_HIP_CHECK(hipGetLastError());
                return UH_OK;
            }
        }
    }
    if (qpw > 1) {
        const dim3 gq(uh_div_up(nq, kWavesPerBlock * qpw));
#define UH_KNN_MQ(LV, Q) UH_LAUNCH(idx->ctx, (knn_search_mq_kernel<LV, Q>), gq, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances)
        if (nn <= 3) { if (qpw == 2) UH_KNN_MQ(1, 2); else UH_KNN_MQ(1, 4); }
        else { if (qpw == 2) UH_KNN_MQ(3, 2); else UH_KNN_MQ(3, 4); }
#undef UH_KNN_MQ
    } else
    if (nn <= 3) UH_LAUNCH(idx->ctx, knn_search_kernel<1>, grid, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances);
    else if (nn <= 15) UH_LAUNCH(idx->ctx, knn_search_kernel<3>, grid, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances);
    else UH_LAUNCH(idx->ctx, knn_search_kernel<6>, grid, block, 0, idx->d_train, idx->shard_begin, idx->shard_end, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

int uh_knn_search_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int32_t* d_indices,
                      int32_t* d_distances, int sorted, int max_dist) {
    return knn_search_dev_impl(idx, d_queries, nq, nn, d_indices, d_distances, sorted, max_dist, KnnHostOut{nullptr, nullptr, 0u, nullptr, 0ull}, nullptr);
}

int uh_knn_search(uh_knn* idx, const uint8_t* queries, int nq, size_t q_stride, int nn, int32_t* indices,
                  int32_t* distances, int sorted, int max_dist) {
    int rc = check_search_args(idx, queries, nq, nn, indices, distances);
    if (rc) return rc;
    if (nq == 0) return UH_OK;
    UH_REQUIRE(q_stride >= 32, "uh_knn_search: query stride %zu < 32", q_stride);
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    hipStream_t st = idx->ctx->stream;
    if ((rc = idx->q_buf.reserve((size_t)nq * 32))) return rc;
    if ((rc = idx->idx_buf.reserve((size_t)nq * nn * 4))) return rc;
    if ((rc = idx->dist_buf.reserve((size_t)nq * nn * 4))) return rc;
    // Pinned buffers (uh_host_alloc / hipHostMalloc / hipHostRegister) move as 16-byte-wide launches on the context stream and the host
    // polls a completion word: one frame's search is latency, and a copy-engine transfer + stream synchronisation on each side costs
    // more than the search.  Pageable buffers go through the runtime's copies (contiguous rows as ONE copy: a "2-D" copy of 2000 rows
    // of 32 bytes is issued row by row).
    const size_t out_bytes = (size_t)nq * nn * 4;
    const uint8_t* pq = q_stride == 32 ? static_cast<const uint8_t*>(uh::device_alias_of_host(queries)) : nullptr;
    if (pq && (reinterpret_cast<uintptr_t>(pq) & 15) == 0) { if ((rc = uh::copy16(idx->ctx, idx->q_buf.p, pq, (size_t)nq * 32))) return rc; }
    else if (q_stride == 32) UH_HIP_CHECK(hipMemcpyAsync(idx->q_buf.p, queries, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    else UH_HIP_CHECK(hipMemcpy2DAsync(idx->q_buf.p, 32, queries, q_stride, 32, (size_t)nq, hipMemcpyHostToDevice, st));
    int32_t* pi = static_cast<int32_t*>(uh::device_alias_of_host(indices));
    int32_t* pd = pi ? static_cast<int32_t*>(uh::device_alias_of_host(distances)) : nullptr;
    const bool pinned_out = pi && pd && (out_bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(pi) | reinterpret_cast<uintptr_t>(pd)) & 15) == 0;
    KnnHostOut ho{nullptr, nullptr, 0u, nullptr, 0ull};
    if (pinned_out) {
        if ((rc = idx->h_word.reserve(64))) return rc;
        ho = KnnHostOut{pi, pd, (unsigned)(out_bytes / 16), idx->h_word.dev<unsigned long long>(), idx->host_seq + 1};
    }
    bool handed = false;
    rc = knn_search_dev_impl(idx, idx->q_buf.as<uint8_t>(), nq, nn, idx->idx_buf.as<int32_t>(), idx->dist_buf.as<int32_t>(), sorted, max_dist, ho, &handed);
    if (rc) return rc;
    if (handed) {   // the stream form copied the rows and posts the word itself
        const unsigned long long word = ++idx->host_seq;
        return uh::wait_host_word(idx->h_word.host<volatile unsigned long long>(), word, st, "uh_knn_search");
    }
    if (pinned_out) {
        if ((rc = uh::copy16(idx->ctx, pi, idx->idx_buf.p, out_bytes))) return rc;
        if ((rc = uh::copy16(idx->ctx, pd, idx->dist_buf.p, out_bytes))) return rc;
        const unsigned long long word = ++idx->host_seq;
        if ((rc = uh::post_host_word(idx->ctx, idx->h_word.dev<unsigned long long>(), word))) return rc;
        return uh::wait_host_word(idx->h_word.host<volatile unsigned long long>(), word, st, "uh_knn_search");
    }
    UH_HIP_CHECK(hipMemcpyAsync(indices, idx->idx_buf.p, out_bytes, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(distances, idx->dist_buf.p, out_bytes, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    return UH_OK;
}

int uh_knn_scan_shard_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int max_dist, uint64_t* d_cand,
                          int32_t* d_counts, int cap) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_cand, d_counts);
    if (rc) return rc;
    UH_REQUIRE(cap >= 1, "uh_knn_scan_shard_dev: cap must be >= 1");
    if (nq == 0) return UH_OK;
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    // rows [shard_begin, shard_end) of this index carry the global indices row_offset + row: the kernel indexes a virtual base that
    // lies row_offset rows in front of the tile (only rows inside the tile are ever addressed)
    if (nn <= kRpK && (long long)idx->row_offset + idx->shard_end <= (1 << 23) && !getenv("UH_KNN_SHARD_FORM")) {   // the threshold-only scan, two queries per wave
        const dim3 g2(uh_div_up(uh_div_up(nq, 2), kWavesPerBlock));
        UH_LAUNCH(idx->ctx, knn_scan_shard2_kernel, g2, block, 0, idx->d_train - (size_t)idx->row_offset * 32, idx->row_offset + idx->shard_begin,
                  idx->row_offset + idx->shard_end, d_queries, nq, nn, max_dist, d_cand, d_counts, cap, static_cast<int*>(nullptr), idx->d_valid_rows);
        UH_HIP_CHECK(hipGetLastError());
        return UH_OK;
    }
    UH_LAUNCH(idx->ctx,knn_scan_shard_kernel, grid, block, 0, idx->d_train - (size_t)idx->row_offset * 32, idx->row_offset + idx->shard_begin,
                       idx->row_offset + idx->shard_end, d_queries, nq, nn, max_dist, d_cand, d_counts, cap, idx->d_valid_rows);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

// Query rows [*d_valid_rows, nq) of every following uh_knn_scan_shard_dev are not scanned and emit empty lists (NULL: every row counts).
// For callers whose query buffer is a fixed-capacity frame block of which only the first `count` rows — a number that exists on the
// device only — are this frame's descriptors (the stream of fstream.hip): stale rows could otherwise overflow a list and raise the
// replay's overflow flag for a frame whose valid rows all fit.
int uh_knn_set_valid_rows_dev(uh_knn* idx, const int32_t* d_valid_rows) {
    UH_REQUIRE(idx, "uh_knn_set_valid_rows_dev: NULL index");
    idx->d_valid_rows = d_valid_rows;
    return UH_OK;
}

int uh_knn_replay_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                      const uint64_t* d_cand_all, const int32_t* d_counts_all, int nshards, int cap,
                      int32_t* d_indices, int32_t* d_distances) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_indices, d_distances);
    if (rc) return rc;
    UH_REQUIRE(nshards >= 1 && nshards <= kMaxShards, "uh_knn_replay_dev: nshards=%d outside [1,%d]", nshards, kMaxShards);
    UH_REQUIRE(cap >= 1 && d_cand_all && d_counts_all, "uh_knn_replay_dev: bad candidate buffers");
    if (nq == 0) return UH_OK;
    // shard s covers rows [s*nt/n, (s+1)*nt/n): the same split every rank uses (see parallel.py)
    ShardBounds sb;
    sb.n = nshards;
    for (int s = 0; s <= nshards; ++s) sb.b[s] = (int)((long long)idx->nt * s / nshards);
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    launch_replay(idx, grid, block, sb, d_queries, nq, nn, sorted, max_dist, d_cand_all, d_counts_all, cap, d_indices, d_distances, nullptr);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

// replay for ranks that hold only their own tile of the train set: no rescan; *d_overflow is OR-ed with 1 if some list overflowed
int uh_knn_replay_tiles_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                            const uint64_t* d_cand_all, const int32_t* d_counts_all, int nshards, int cap,
                            int32_t* d_indices, int32_t* d_distances, int32_t* d_overflow) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_indices, d_distances);
    if (rc) return rc;
    UH_REQUIRE(nshards >= 1 && nshards <= kMaxShards, "uh_knn_replay_tiles_dev: nshards=%d outside [1,%d]", nshards, kMaxShards);
    UH_REQUIRE(cap >= 1 && d_cand_all && d_counts_all && d_overflow, "uh_knn_replay_tiles_dev: bad candidate buffers");
    if (nq == 0) return UH_OK;
    ShardBounds sb;
    sb.n = nshards;
    for (int s = 0; s <= nshards; ++s) sb.b[s] = 0;   // never used: there is no rescan in this form
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    if (!launch_replay_lanes(idx, nq, nn, sorted, max_dist, d_cand_all, 0, d_counts_all, 0, nshards, cap, d_indices, d_distances, (int*)d_overflow))
        launch_replay(idx, grid, block, sb, d_queries, nq, nn, sorted, max_dist, d_cand_all, d_counts_all, cap, d_indices, d_distances, (int*)d_overflow);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

// the same replay on lists that still sit inside the gathered messages: shard s's block begins cand_stride / count_stride ELEMENTS behind shard
// s-1's (the sharded frame stream, csrc/fstream.hip: nothing is unpacked before the replay)
int uh_knn_replay_tiles_strided_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int sorted, int max_dist,
                                    const uint64_t* d_cand_all, size_t cand_stride, const int32_t* d_counts_all, size_t count_stride, int nshards, int cap,
                                    int32_t* d_indices, int32_t* d_distances, int32_t* d_overflow) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_indices, d_distances);
    if (rc) return rc;
    UH_REQUIRE(nshards >= 1 && nshards <= kMaxShards, "uh_knn_replay_tiles_strided_dev: nshards=%d outside [1,%d]", nshards, kMaxShards);
    UH_REQUIRE(cap >= 1 && d_cand_all && d_counts_all && d_overflow, "uh_knn_replay_tiles_strided_dev: bad candidate buffers");
    UH_REQUIRE(nshards == 1 || (cand_stride >= (size_t)nq * cap && count_stride >= (size_t)nq), "uh_knn_replay_tiles_strided_dev: strides shorter than a shard's block");
    if (nq == 0) return UH_OK;
    ShardBounds sb;
    sb.n = nshards;
    for (int s = 0; s <= nshards; ++s) sb.b[s] = 0;
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    if (!launch_replay_lanes(idx, nq, nn, sorted, max_dist, d_cand_all, cand_stride, d_counts_all, count_stride, nshards, cap, d_indices, d_distances, (int*)d_overflow))
        launch_replay(idx, grid, block, sb, d_queries, nq, nn, sorted, max_dist, d_cand_all, d_counts_all, cap, d_indices, d_distances, (int*)d_overflow, cand_stride, count_stride);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

int uh_knn_set_row_offset(uh_knn* idx, int offset) {
    UH_REQUIRE(idx && offset >= 0, "uh_knn_set_row_offset: bad argument");
    idx->row_offset = offset;
    return UH_OK;
}

// ---- hierarchical k-means index
int uh_knn_build_kmeans(uh_knn* idx, const uint8_t* features, int n, int k, int max_iters) {
    UH_REQUIRE(idx, "uh_knn_build_kmeans: NULL index");
    idx->km_blob.clear();
    idx->km_blob_stale = false;
    idx->km_n = 0;
    if (n <= 0) return UH_OK;   // index.cpp:49 — empty features leave the index unbuilt
    UH_REQUIRE(features != nullptr, "uh_knn_build_kmeans: NULL features");
    UH_REQUIRE(k >= 2 && k <= kWave, "uh_knn_build_kmeans: k=%d outside [2,%d] (one lane per child)", k, kWave);
    UH_REQUIRE(max_iters >= -1, "uh_knn_build_kmeans: maxIters=%d (use -1 for 'until convergence')", max_iters);
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    hipStream_t st = idx->ctx->stream;
    int rc;
    if ((rc = idx->km_rows.reserve(32 * (size_t)n))) return rc;
    {   // the rows to HBM by a 16-byte-wide launch, from where they lie if the caller's array is pinned, else through this object's pinned block
        // (hipMemcpyAsync from pageable memory holds the calling thread for the whole staged copy: ~30 us per 320 KB); crosses while the host shuffles the root
        const void* src = ((reinterpret_cast<uintptr_t>(features) & 15) == 0) ? uh::device_alias_of_host(features) : nullptr;
        if (!src) {
            if ((rc = idx->km_rows_pin.reserve(32 * (size_t)n + 16))) return rc;
            std::memcpy(idx->km_rows_pin.host<uint8_t>(), features, 32 * (size_t)n);
            std::atomic_thread_fence(std::memory_order_release);
            src = idx->km_rows_pin.dev<uint8_t>();
        }
        if ((rc = uh::copy16(idx->ctx, idx->km_rows.p, src, 32 * (size_t)n))) return rc;
    }
    // the distances — where the reference's build time goes (kmeansindexcreator.cpp:44-) — on the device, one launch per level and k-means round
    struct DevAssigner : KmAssigner {
        uh_knn* idx; hipStream_t st; size_t o_out = 0;
        int reserve(size_t npos, size_t nslots, int kk, KmAssignIO& io) override {
            auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
            o_out = 64;
            const size_t o_row = al(o_out + npos), o_slot = al(o_row + 4 * npos), o_nc = al(o_slot + 2 * npos), o_cen = al(o_nc + nslots),
                         total = al(o_cen + nslots * (size_t)kk * 32);
            const int rc2 = idx->km_stage.reserve(total);   // (the previous level's launch has been awaited: the block is free)
            if (rc2) return rc2;
            char* hb = idx->km_stage.host<char>();
            io.npos = npos; io.nslots = nslots;
            io.pos_row = reinterpret_cast<uint32_t*>(hb + o_row); io.pos_slot = reinterpret_cast<uint16_t*>(hb + o_slot);
            io.slot_nc = reinterpret_cast<uint8_t*>(hb + o_nc); io.centres = reinterpret_cast<uint8_t*>(hb + o_cen);
            io.cluster = reinterpret_cast<const uint8_t*>(hb + o_out);
            return UH_OK;
        }
        int run(const uint8_t*, int kk, KmAssignIO& io) override {
            const auto ta = std::chrono::steady_clock::now();
            char* hb = idx->km_stage.host<char>();
            char* db = idx->km_stage.dev<char>();
            auto dev = [&](const void* h) { return db + (reinterpret_cast<const char*>(h) - hb); };
            std::atomic_thread_fence(std::memory_order_release);
            UH_LAUNCH(idx->ctx, kmeans_assign_kernel, dim3(uh_div_up((int)io.npos, 256)), dim3(256), 0, (const uint8_t*)idx->km_rows.as<uint8_t>(),
                      (const uint32_t*)dev(io.pos_row), (const uint16_t*)dev(io.pos_slot), (const uint8_t*)dev(io.centres), (const uint8_t*)dev(io.slot_nc), kk,
                      (int)io.npos, reinterpret_cast<uint8_t*>(db + o_out));
            UH_HIP_CHECK(hipGetLastError());
            const unsigned long long word = ++idx->km_seq;
            int rc2;
            if ((rc2 = uh::post_host_word(idx->ctx, reinterpret_cast<unsigned long long*>(db), word))) return rc2;
            if ((rc2 = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(hb), word, st, "uh_knn_build_kmeans"))) return rc2;
            idx->km_assign_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ta).count();
            idx->km_assign_calls++;
            return UH_OK;
        }
    } assign_dev;
    assign_dev.idx = idx; assign_dev.st = st;
    idx->km_assign_us = 0; idx->km_assign_calls = 0;
    static const bool km_timing = getenv("UH_KM_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (idx->km_pin_word) {   // the previous build's upload has read the pinned block (long since, unless two builds follow each other directly)
        if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(idx->km_pin.host<char>()), idx->km_pin_word, st, "uh_knn_build_kmeans"))) return rc;
        idx->km_pin_word = 0;
    }
    int rc_alloc = UH_OK;
    const std::function<uint8_t*(size_t)> blob_alloc = [idx, &rc_alloc](size_t total) -> uint8_t* {
        rc_alloc = idx->km_pin.reserve(total + 64 + 16);
        if (rc_alloc) return nullptr;
        idx->km_size = total;
        return idx->km_pin.host<uint8_t>() + 64;
    };
    rc = kmeans_build_blob(features, n, k, max_iters, blob_alloc, idx->km_depth, assign_dev);
    if (rc) { idx->km_size = 0; return rc_alloc ? rc_alloc : rc; }
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = idx->km_dev.reserve(idx->km_size + 64))) return rc;
    std::atomic_thread_fence(std::memory_order_release);
    if ((rc = uh::copy16(idx->ctx, idx->km_dev.p, idx->km_pin.dev<char>() + 64, idx->km_size))) return rc;   // no synchronisation: searches follow on the same stream
    idx->km_pin_word = ++idx->km_seq;
    if ((rc = uh::post_host_word(idx->ctx, idx->km_pin.dev<unsigned long long>(), idx->km_pin_word))) return rc;
    idx->km_blob_stale = true;
    if (km_timing) fprintf(stderr, "kmeans build n=%d: tree %.1f us (assign steps %.1f us in %d launches), blob upload enqueued in %.1f us\n", n,
                           std::chrono::duration<double, std::micro>(t1 - t0).count(), idx->km_assign_us, idx->km_assign_calls,
                           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count());
    idx->km_k = k;
    idx->km_n = n;
    return UH_OK;
}

// host-only form of the build (no GPU needed): writes min(cap, size) bytes of the block data, returns the full size in *size
int uh_knn_kmeans_build_host(const uint8_t* features, int n, int k, int max_iters, uint8_t* out, uint64_t cap, uint64_t* size) {
    UH_REQUIRE(features && n > 0 && k >= 2 && k <= kWave && max_iters >= -1 && size, "uh_knn_kmeans_build_host: bad arguments");
    std::vector<uint8_t> blob;
    int depth = 0;
    KmAssignerHost host_asg;
    const int rc = kmeans_build_blob(features, n, k, max_iters, [&blob](size_t total) { blob.resize(total); return blob.data(); }, depth, host_asg);
    if (rc) return rc;
    *size = blob.size();
    if (out && cap) std::memcpy(out, blob.data(), (size_t)std::min<uint64_t>(cap, blob.size()));
    return UH_OK;
}

// ---- xflann::Index::toStream / fromStream (index.cpp:153-188) for the k-means index: u64 12837333433, u64 std::hash<std::string>("kmeans"),
// then KMeansIndex::toStream (kmeansindex.cpp:209-216): u64 55824124, the 40-byte params struct, the block data.
namespace {
constexpr uint64_t kXflannSig = 12837333433ull, kKmeansSig = 55824124ull;
struct KmStreamParams {   // KMeansIndex::params (kmeansindex.h:86-93), natural x86-64 layout
    uint32_t aligment; uint32_t pad0; uint64_t desc_size_bytes_wp; uint64_t total_size; int32_t desc_type; uint32_t desc_size; uint32_t npoints; uint32_t pad1;
};
static_assert(sizeof(KmStreamParams) == 40, "KMeansIndex::params layout");
uint64_t impl_hash(const char* name) { return (uint64_t)std::hash<std::string>()(std::string(name)); }   // libstdc++, like the reference's build

// depth (levels of internal blocks above the deepest leaf block) and largest child count of a block-data blob; false if an offset leaves it
bool km_blob_shape(const std::vector<uint8_t>& blob, int& depth, int& max_n) {
    depth = 0; max_n = 0;
    struct It { uint64_t off; int lvl; };
    std::vector<It> todo{{0, 0}};
    size_t visited = 0;
    while (!todo.empty()) {
        const It it = todo.back(); todo.pop_back();
        if (++visited > blob.size() / 8 + 1) return false;                       // a cycle
        if (it.off + 8 > blob.size()) return false;
        uint32_t hdr, hs;
        std::memcpy(&hdr, &blob[it.off], 4); std::memcpy(&hs, &blob[it.off + 4], 4);
        const int n = (int)(hdr & 0xffffu);
        const bool leaf = ((hdr >> 16) & 0xffu) != 0;
        if (n > kWave || it.off + hs + 32ull * n > blob.size() || it.off + 8 + 8ull * n > blob.size()) return false;
        max_n = std::max(max_n, n);
        if (leaf) { depth = std::max(depth, it.lvl); continue; }
        for (int c = 0; c < n; c++) {
            uint64_t info;
            std::memcpy(&info, &blob[it.off + 8 + 8ull * c], 8);
            if (info >> 63) continue;                                            // a row index, not a child block
            todo.push_back({info, it.lvl + 1});
        }
    }
    return true;
}
}  // namespace

// the host copy of a device-built index's block data, made on first use (the build leaves it in pinned memory only)
static void km_sync_host_blob(uh_knn* idx) {
    if (!idx->km_blob_stale) return;
    idx->km_blob.assign(idx->km_pin.host<uint8_t>() + 64, idx->km_pin.host<uint8_t>() + 64 + idx->km_size);
    idx->km_blob_stale = false;
}

int uh_knn_to_stream(uh_knn* idx, uint8_t* out, uint64_t cap, uint64_t* size) {
    UH_REQUIRE(idx && size, "uh_knn_to_stream: NULL argument");
    km_sync_host_blob(idx);
    if (idx->km_n == 0 || idx->km_blob.empty()) {
        // Index::toStream throws for an index that does not own its features; Linear::toStream is "Not yet" in the reference (linear.cpp:78-80)
        uh::set_error("uh_knn_to_stream: only the hierarchical k-means index has a stream form (xflann: Linear::toStream is not implemented)");
        return UH_ENOTBUILT;
    }
    *size = 64 + idx->km_blob.size();
    if (!out || cap < *size) return out ? UH_ECAPACITY : UH_OK;
    KmStreamParams P{};
    P.aligment = 8; P.desc_size_bytes_wp = 32; P.total_size = idx->km_blob.size(); P.desc_type = 0 /* XFLANN_8U */; P.desc_size = 32; P.npoints = (uint32_t)idx->km_n;
    const uint64_t h = impl_hash("kmeans");
    std::memcpy(out, &kXflannSig, 8); std::memcpy(out + 8, &h, 8); std::memcpy(out + 16, &kKmeansSig, 8);
    std::memcpy(out + 24, &P, 40);
    std::memcpy(out + 64, idx->km_blob.data(), idx->km_blob.size());
    return UH_OK;
}

int uh_knn_from_stream(uh_knn* idx, const uint8_t* data, uint64_t nbytes) {
    UH_REQUIRE(idx && data, "uh_knn_from_stream: NULL argument");
    UH_REQUIRE(nbytes >= 16, "uh_knn_from_stream: stream too short");
    uint64_t sig, h;
    std::memcpy(&sig, data, 8); std::memcpy(&h, data + 8, 8);
    UH_REQUIRE(sig == kXflannSig, "Invalid signature in stream");                                       // index.cpp:173
    if (h == impl_hash("linear") || h == impl_hash("kdtree")) {
        uh::set_error("uh_knn_from_stream: the stream holds a %s index; only the k-means index is supported (the reference cannot stream Linear either)", h == impl_hash("linear") ? "linear" : "kd-tree");
        return UH_EINVAL;
    }
    UH_REQUIRE(h == impl_hash("kmeans"), "Could not determine the type of implementation from the hash readed");   // index.cpp:184
    UH_REQUIRE(nbytes >= 64, "uh_knn_from_stream: truncated k-means header");
    uint64_t sig2;
    std::memcpy(&sig2, data + 16, 8);
    UH_REQUIRE(sig2 == kKmeansSig, "XFlann::fromStream invalid signature");                             // kmeansindex.cpp:223 (the reference only constructs the exception)
    KmStreamParams P;
    std::memcpy(&P, data + 24, 40);
    UH_REQUIRE(P.desc_type == 0 && P.desc_size == 32 && P.desc_size_bytes_wp == 32, "uh_knn_from_stream: not a 32-byte binary-descriptor index (type %d, %u bytes)", P.desc_type, P.desc_size);
    UH_REQUIRE(P.total_size <= nbytes - 64 && P.total_size > 0 && P.npoints > 0, "uh_knn_from_stream: block data truncated (%llu of %llu bytes)", (unsigned long long)(nbytes - 64), (unsigned long long)P.total_size);
    std::vector<uint8_t> blob(data + 64, data + 64 + P.total_size);
    int depth = 0, max_n = 0;
    UH_REQUIRE(km_blob_shape(blob, depth, max_n) && max_n >= 1, "uh_knn_from_stream: inconsistent block data");
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    int rc;
    if ((rc = idx->km_dev.reserve(blob.size() + 64))) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(idx->km_dev.p, blob.data(), blob.size(), hipMemcpyHostToDevice, idx->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(idx->ctx->stream));
    idx->km_blob.swap(blob);
    idx->km_blob_stale = false;
    idx->km_depth = depth;
    idx->km_k = std::max(max_n, 2);
    idx->km_n = (int)P.npoints;
    return UH_OK;
}

int uh_knn_kmeans_blob(uh_knn* idx, const uint8_t** data, uint64_t* size) {
    UH_REQUIRE(idx && data && size, "uh_knn_kmeans_blob: NULL argument");
    km_sync_host_blob(idx);
    *data = idx->km_blob.empty() ? nullptr : idx->km_blob.data();
    *size = idx->km_blob.size();
    return UH_OK;
}

int uh_knn_search_kmeans_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int max_checks, int sorted, int32_t* d_indices,
                             int32_t* d_distances) {
    UH_REQUIRE(idx, "uh_knn_search_kmeans: NULL index");
    if (idx->km_n == 0) {
        uh::set_error("uh_knn_search_kmeans: could not run search because index not created");   // index.cpp:82-85
        return UH_ENOTBUILT;
    }
    UH_REQUIRE(nq >= 0, "uh_knn_search_kmeans: negative query count");
    UH_REQUIRE(nn >= 1 && nn <= kWave, "uh_knn_search_kmeans: nn=%d outside [1,%d]", nn, kWave);
    UH_REQUIRE(!(nn == 1 && max_checks == 1) && !(nn == 2 && max_checks <= 2),
               "uh_knn_search_kmeans: (nn=%d, maxChecks=%d) selects the reference's greedy 1-/2-nn descents (kmeansindex.h:216-224), which return a constant row with distance -1 for binary descriptors; not reproduced", nn, max_checks);
    if (nq == 0) return UH_OK;
    UH_REQUIRE(d_queries && d_indices && d_distances, "uh_knn_search_kmeans: NULL buffer");
    // worst case of the branch heap: every pop adds >= 1 check, every descent pushes <= (k-1) entries per level; beyond the
    // reference's capacity its pushes are dropped, and so are ours
    const long long worst = (long long)std::max(max_checks, 0) * std::max(idx->km_depth, 1) * (idx->km_k - 1) + 2;
    const int cap = (int)std::min<long long>(worst, kBranchMax);
    const size_t lds = (size_t)kKmWaves * 2 * cap * sizeof(int);
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    if (!idx->km_attr) {
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kmeans_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kKmWaves * 2 * kBranchMax * 4));
        idx->km_attr = true;
    }
    UH_LAUNCH(idx->ctx, knn_kmeans_search_kernel, dim3(uh_div_up(nq, kKmWaves)), dim3(kWave * kKmWaves), lds, idx->km_dev.as<uint8_t>(), d_queries, nq,
              nn, max_checks, sorted ? 1 : 0, d_indices, d_distances, cap);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

int uh_knn_search_kmeans(uh_knn* idx, const uint8_t* queries, int nq, int nn, int max_checks, int sorted, int32_t* indices, int32_t* distances) {
    UH_REQUIRE(idx, "uh_knn_search_kmeans: NULL index");
    if (nq > 0) UH_REQUIRE(queries && indices && distances, "uh_knn_search_kmeans: NULL buffer");
    int rc;
    hipStream_t st = idx->ctx->stream;
    if (nq > 0) {
        UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
        if ((rc = idx->q_buf.reserve((size_t)nq * 32))) return rc;
        if ((rc = idx->idx_buf.reserve((size_t)nq * std::max(nn, 1) * 4))) return rc;
        if ((rc = idx->dist_buf.reserve((size_t)nq * std::max(nn, 1) * 4))) return rc;
        UH_HIP_CHECK(hipMemcpyAsync(idx->q_buf.p, queries, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    }
    rc = uh_knn_search_kmeans_dev(idx, idx->q_buf.as<uint8_t>(), nq, nn, max_checks, sorted, idx->idx_buf.as<int32_t>(), idx->dist_buf.as<int32_t>());
    if (rc || nq == 0) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(indices, idx->idx_buf.p, (size_t)nq * nn * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(distances, idx->dist_buf.p, (size_t)nq * nn * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    return UH_OK;
}

}  // extern "C"
