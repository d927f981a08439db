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

// Lane-distributed ResultSet: lane j holds heap slot j.  Every index below is wave-uniform (derived from ballots and
// v_readlane results), so slot reads are single v_readlane_b32 and slot writes are v_cndmask selects — no exec-mask
// branches inside a push.  The sifts use the "hole" formulation, which performs exactly the swaps of resultset.h.
struct WaveHeap {
    int hd;    // per lane: distance of slot `lane`
    int hi;    // per lane: index of slot `lane`
    int size;  // wave-uniform
    int lane;

    __device__ __forceinline__ void put(int slot, int d, int i) {
        const bool m = lane == slot;
        hd = m ? d : hd;
        hi = m ? i : hi;
    }
    __device__ __forceinline__ void swap(int a, int b) {
        const int da = rl(hd, a), db = rl(hd, b), ia = rl(hi, a), ib = rl(hi, b);
        put(a, db, ib);
        put(b, da, ia);
    }
    // accept test of resultset.h:66-69 (radius bound, then "full and not better than the worst")
    __device__ __forceinline__ bool accepts(int d, int k, int maxd) const {
        if (maxd >= 0 && maxd < d) return false;
        if (size >= k) return d < rl(hd, 0);
        return true;
    }
    // resultset.h:64-82, caller has already established accepts(d)
    __device__ __forceinline__ void push_accepted(int d, int idx, int k) {
        if (size >= k) {
            // swap(0,size-1); size--; if (size>1) up(0): the old root moves to slot size-1, which the new element
            // overwrites below, so only the old LAST element has to be re-seated from the root downwards ("up", :104-135)
            const int last = size - 1;
            const int md = rl(hd, last), mi = rl(hi, last);
            size = last;
            if (size >= 1) {
                int pos = 0;
                if (size > 1) {
                    for (;;) {
                        const int l = 2 * pos + 1, r = l + 1;
                        if (l >= size) break;
                        const int dl = rl(hd, l);
                        if (r >= size) {
                            if (md < dl) { put(pos, dl, rl(hi, l)); pos = l; }
                            break;
                        }
                        const int dr = rl(hd, r);
                        const int c = (dr < dl) ? l : r;
                        const int dc = (dr < dl) ? dl : dr;
                        if (!(md < dc)) break;
                        put(pos, dc, rl(hi, c));
                        pos = c;
                    }
                }
                put(pos, md, mi);
            }
        }
        // append at slot `size` and sift towards the root ("down", :93-100)
        int pos = size;
        while (pos != 0) {
            const int parent = (pos - 1) >> 1;
            const int dp = rl(hd, parent);
            if (!(dp < d)) break;
            put(pos, dp, rl(hi, parent));
            pos = parent;
        }
        put(pos, d, idx);
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
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int d = hamming256(a0[u], a1[u], q);
            feed_step<EMIT>(h, d, base + u * kWave + lane, true, k, maxd, cand_row, ncand, cap);
        }
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
