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
// Sharding (multi-GPU, row (e) of the scope table): a shard scan starts from an empty heap, so
// its accept test is looser than the global one; it emits everything it accepted (a superset of
// the global accept set, in index order).  Replaying the concatenated shard lists through the
// same push routine reproduces the global heap bit for bit.
#include "common.hpp"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxShards = 64;

struct ShardBounds { int n; int b[kMaxShards + 1]; };

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

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
    // resultset.h:64-82, caller has already established accepts(d)
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
                for (int t = 0; t < 6; t++) {                 // heap depth <= 6 for k <= 64
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

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint32_t (&q)[8]) {
    int d = __popc(a0.x ^ q[0]);
    d += __popc(a0.y ^ q[1]);
    d += __popc(a0.z ^ q[2]);
    d += __popc(a0.w ^ q[3]);
    d += __popc(a1.x ^ q[4]);
    d += __popc(a1.y ^ q[5]);
    d += __popc(a1.z ^ q[6]);
    d += __popc(a1.w ^ q[7]);
    return d;
}

// Feed one step (64 candidates, one per lane, ascending index with lane) into the heap.
template <bool EMIT>
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
        h.push_accepted(dl, il, k);
    }
}

template <bool EMIT>
__device__ __forceinline__ void scan_range(WaveHeap& h, const uint8_t* __restrict__ train, int t0, int t1,
                                           const uint32_t (&q)[8], int k, int maxd,
                                           uint64_t* cand_row, int& ncand, int cap) {
    const int lane = h.lane;
    constexpr int UNROLL = 4;
    int base = t0;
    for (; base + UNROLL * kWave <= t1; base += UNROLL * kWave) {
        uint4 a0[UNROLL], a1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint4* p = reinterpret_cast<const uint4*>(train + (size_t)(base + u * kWave + lane) * 32);
            a0[u] = p[0];
            a1[u] = p[1];
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
        for (int u = 0; u < UNROLL; ++u) feed_step<EMIT>(h, d[u], base + u * kWave + lane, true, k, maxd, cand_row, ncand, cap);
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
        feed_step<EMIT>(h, d, t, valid, k, maxd, cand_row, ncand, cap);
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

// One wave per query; whole train range; writes final rows.
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
    scan_range<false>(h, train, t0, t1, q, k, maxd, nullptr, ncand, 0);
    finish_row(h, k, sorted, indices, distances, qi);
}

// Shard scan: emits the locally accepted candidates in index order.
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_scan_shard_kernel(
    const uint8_t* __restrict__ train, int t0, int t1, const uint8_t* __restrict__ queries, int nq, int k,
    int maxd, uint64_t* __restrict__ cand, int32_t* __restrict__ counts, int cap) {
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (qi >= nq) return;
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    int ncand = 0;
    scan_range<true>(h, train, t0, t1, q, k, maxd, cand + (size_t)qi * cap, ncand, cap);
    if (lane == 0) counts[qi] = ncand;
}

// Replay the concatenated shard candidate lists (shard order = index order) through the exact heap.
// A shard whose list overflowed (count > cap) is rescanned from the descriptors.
__global__ __launch_bounds__(kWave* kWavesPerBlock) void knn_replay_kernel(
    const uint8_t* __restrict__ train, ShardBounds sb, const uint8_t* __restrict__ queries, int nq, int k,
    int sorted, int maxd, const uint64_t* __restrict__ cand_all, const int32_t* __restrict__ counts_all, int cap,
    int32_t* __restrict__ indices, int32_t* __restrict__ distances) {
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (qi >= nq) return;
    uint32_t q[8];
    load_query(queries, qi, q);
    WaveHeap h{0, -1, 0, lane};
    int dummy = 0;
    for (int s = 0; s < sb.n; ++s) {
        int cnt = __builtin_amdgcn_readfirstlane(counts_all[(size_t)s * nq + qi]);
        if (cnt > cap) {
            scan_range<false>(h, train, sb.b[s], sb.b[s + 1], q, k, maxd, nullptr, dummy, 0);
            continue;
        }
        const uint64_t* row = cand_all + ((size_t)s * nq + qi) * cap;
        for (int base = 0; base < cnt; base += kWave) {
            int j = base + lane;
            bool valid = j < cnt;
            uint64_t c = valid ? row[j] : 0;
            feed_step<false>(h, (int)(c >> 32), (int)(uint32_t)c, valid, k, maxd, nullptr, dummy, 0);
        }
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

struct uh_knn {
    uh_ctx* ctx = nullptr;
    uh::DevBuf train_store;       // owned copy (build from host)
    const uint8_t* d_train = nullptr;
    int nt = 0;
    int shard_begin = 0, shard_end = 0;
    uh::DevBuf q_buf, idx_buf, dist_buf;  // staging for the host-pointer API
};

extern "C" {

int uh_knn_create(uh_ctx* ctx, uh_knn** out) {
    UH_REQUIRE(ctx && out, "uh_knn_create: NULL argument");
    uh_knn* k = new uh_knn();
    k->ctx = ctx;
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

int uh_knn_search_dev(uh_knn* idx, const uint8_t* d_queries, int nq, int nn, int32_t* d_indices,
                      int32_t* d_distances, int sorted, int max_dist) {
    int rc = check_search_args(idx, d_queries, nq, nn, d_indices, d_distances);
    if (rc) return rc;
    if (nq == 0) return UH_OK;
    UH_HIP_CHECK(hipSetDevice(idx->ctx->device));
    dim3 grid(uh_div_up(nq, kWavesPerBlock)), block(kWave * kWavesPerBlock);
    UH_LAUNCH(idx->ctx,knn_search_kernel, grid, block, 0, idx->d_train, idx->shard_begin,
                       idx->shard_end, d_queries, nq, nn, sorted ? 1 : 0, max_dist, d_indices, d_distances);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
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
    UH_HIP_CHECK(hipMemcpy2DAsync(idx->q_buf.p, 32, queries, q_stride, 32, (size_t)nq, hipMemcpyHostToDevice, st));
    rc = uh_knn_search_dev(idx, idx->q_buf.as<uint8_t>(), nq, nn, idx->idx_buf.as<int32_t>(),
                           idx->dist_buf.as<int32_t>(), sorted, max_dist);
    if (rc) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(indices, idx->idx_buf.p, (size_t)nq * nn * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(distances, idx->dist_buf.p, (size_t)nq * nn * 4, hipMemcpyDeviceToHost, st));
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
    UH_LAUNCH(idx->ctx,knn_scan_shard_kernel, grid, block, 0, idx->d_train, idx->shard_begin,
                       idx->shard_end, d_queries, nq, nn, max_dist, d_cand, d_counts, cap);
    UH_HIP_CHECK(hipGetLastError());
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
    UH_LAUNCH(idx->ctx,knn_replay_kernel, grid, block, 0, idx->d_train, sb, d_queries, nq, nn,
                       sorted ? 1 : 0, max_dist, d_cand_all, d_counts_all, cap, d_indices, d_distances);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

}  // extern "C"
