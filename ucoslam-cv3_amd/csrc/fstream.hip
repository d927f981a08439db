// One frame stream sharded over the GPUs of a node (BASELINE config 5), below Python: C ABI uh_fstream_*.
//
// What is sharded (every rank = one process = one GPU):
//   * ORB extraction of frame t: pyramid levels [first, end) per rank (ORBextractor.cpp:501-513: a level's budget, thresholds and
//     cell grid do not depend on the other levels; :1286-1300: outputs concatenated in level order), uh_orb_set_level_range;
//   * matching of frame t-1 against the map: rank r holds ONLY tile r of the train descriptors (uh_knn_set_row_offset), scans it for
//     every query and emits the candidates its local heap accepted with GLOBAL row indices (uh_knn_scan_shard_dev);
//   * the fbow descent of frame t-1's descriptors: a slice of the rows per rank (uh_bow_transform_dev).
// ONE collective per frame: an all-gather of fixed-size messages.  Nothing is packed: the producers write straight into the send
// buffer (the extractor's keypoint / descriptor / count outputs, the scan's lists, the descent's arrays ARE regions of the message),
// and nothing but the level rows is unpacked: the exact replay (uh_knn_replay_tiles_strided_dev) reads every rank's lists where the
// all-gather left them.  Counts stay on the device: a step issues launches and one collective and never synchronises with the host;
// the stream is software-pipelined (results for frame t-1 arrive with the extraction of frame t), so a frame costs one collective.
//
// Message layout (bytes, every field 16-byte aligned; F = max_features, C = cand_cap, S = ceil(F / world)):
//   header  16        int32 n_level_rows (the extractor's count output, clamped by the readers), 0, 0, 0
//   kps     F * 28    cv::KeyPoint rows of this rank's levels
//   desc    F * 32
//   counts  F * 4     accept-list lengths of this rank's tile for the previous frame's F query rows
//   cand    F * C * 8 accept lists (distance << 32 | global train row)
//   bow     S * 13 (+ pad)   word[S] u32, weight[S] f32, node[S] u32, valid[S] u8 of this rank's slice of the previous frame's rows
// The collective: RCCL's all-gather through a communicator this library creates from a unique id (uh_fstream_comm_*; librccl is
// loaded with dlopen so that hosts without it can still use everything else), or the caller's own ncclComm_t, or — world 1 — a copy.
#include <dlfcn.h>

#include "common.hpp"

namespace {

struct MsgLayout { size_t hdr, kps, desc, counts, cand, bow_word, bow_weight, bow_node, bow_valid, total; int S; };

MsgLayout layout(int F, int C, int world, bool bow) {
    MsgLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
    L.hdr = take(16); L.kps = take((size_t)F * 28); L.desc = take((size_t)F * 32); L.counts = take((size_t)F * 4); L.cand = take((size_t)F * C * 8);
    L.S = (F + world - 1) / world;
    if (bow) { L.bow_word = take((size_t)L.S * 4); L.bow_weight = take((size_t)L.S * 4); L.bow_node = take((size_t)L.S * 4); L.bow_valid = take((size_t)L.S); }
    L.total = off;
    return L;
}

// All ranks' messages -> the complete extraction: the level rows compacted in rank (= level) order, 15 dwords per row, into the
// caller's arrays AND the stream's own copy (the next step's query rows); the total count; the BoW slices concatenated.
// grid = (row chunks, world); a rank's offset = the sum of the (clamped) counts of the ranks before it.
__global__ __launch_bounds__(256) void fstream_unpack_kernel(const uint8_t* __restrict__ recv, MsgLayout L, int world, int F,
                                                             uint32_t* __restrict__ kps_a, uint32_t* __restrict__ desc_a, uint32_t* __restrict__ kps_b,
                                                             uint32_t* __restrict__ desc_b, int32_t* __restrict__ count_a, int32_t* __restrict__ count_b,
                                                             uint32_t* __restrict__ bow_word, float* __restrict__ bow_weight, uint32_t* __restrict__ bow_node,
                                                             uint8_t* __restrict__ bow_valid, int have_bow, int32_t* __restrict__ overflow,
                                                             const int32_t* __restrict__ prev_count_src, int32_t* __restrict__ prev_count_dst) {
    const int r = blockIdx.y;
    __shared__ int s_off, s_cnt;
    if (threadIdx.x == 0) {
        int off = 0, total = 0, mine = 0;
        bool clipped = false;
        for (int i = 0; i < world; i++) {
            int c = reinterpret_cast<const int32_t*>(recv + (size_t)i * L.total + L.hdr)[0];
            if (c < 0) c = 0;
            if (c > F) { c = F; clipped = true; }
            if (total + c > F) { c = F - total; clipped = true; }   // the level budgets sum to at most F: anything beyond is a corrupt message
            if (i < r) off += c;
            if (i == r) mine = c;
            total += c;
        }
        s_off = off; s_cnt = mine;
        if (clipped && overflow) atomicOr(overflow, 2);
        if (r == 0 && blockIdx.x == 0) {
            if (prev_count_dst) *prev_count_dst = prev_count_src ? *prev_count_src : 0;   // (read before count_b is written: they may be the same row of the pair)
            if (count_a) *count_a = total;
            if (count_b) *count_b = total;
        }
    }
    __syncthreads();
    const int off = s_off, cnt = s_cnt;
    const uint32_t* ks = reinterpret_cast<const uint32_t*>(recv + (size_t)r * L.total + L.kps);
    const uint32_t* ds = reinterpret_cast<const uint32_t*>(recv + (size_t)r * L.total + L.desc);
    // one thread per dword: a keypoint row is 7 dwords, a descriptor row 8
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt * 8; i += gridDim.x * 256) {
        const uint32_t v = ds[i];
        if (desc_a) desc_a[(size_t)off * 8 + i] = v;
        if (desc_b) desc_b[(size_t)off * 8 + i] = v;
        if (i < cnt * 7) {
            const uint32_t k = ks[i];
            if (kps_a) kps_a[(size_t)off * 7 + i] = k;
            if (kps_b) kps_b[(size_t)off * 7 + i] = k;
        }
    }
    if (have_bow && bow_word) {   // slice r = rows [F r / world, F (r+1) / world) of the previous frame
        const int b0 = (int)((long long)F * r / world), b1 = (int)((long long)F * (r + 1) / world);
        const uint8_t* m = recv + (size_t)r * L.total;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < b1 - b0; i += gridDim.x * 256) {
            bow_word[b0 + i] = reinterpret_cast<const uint32_t*>(m + L.bow_word)[i];
            bow_weight[b0 + i] = reinterpret_cast<const float*>(m + L.bow_weight)[i];
            bow_node[b0 + i] = reinterpret_cast<const uint32_t*>(m + L.bow_node)[i];
            bow_valid[b0 + i] = (m + L.bow_valid)[i];
        }
    }
}

// ---- librccl, resolved at run time
struct Id128 { char b[128]; };   // ncclUniqueId
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return UH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};   // (by soname first: the copy a host such as torch has already loaded)
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { uh::set_error("uh_fstream: librccl.so could not be loaded (%s)", dlerror()); return UH_ENODEVICE; }
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, Id128, int)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(h, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.CommCount = reinterpret_cast<int (*)(void*, int*)>(dlsym(h, "ncclCommCount"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        uh::set_error("uh_fstream: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        return UH_ENODEVICE;
    }
    g_rccl.h = h;
    return UH_OK;
}

}  // namespace

struct uh_fstream {
    uh_ctx* ctx = nullptr;
    uh_orb* ext = nullptr;
    uh_knn* tile = nullptr;
    uh_bow* voc = nullptr;
    uh_fstream_params p{};
    MsgLayout L{};
    uh::DevBuf send, recv, rows[2], cnt, bow_tmp;   // rows[i]: kps F*28 | desc F*32 of the frame completed in step parity i
    int cur = 0;              // rows[cur] = the frame the last finish completed (the next step's query rows)
    bool have_prev = false;   // a complete frame exists (false until the first finish)
    bool sent_prev = false;   // the message in flight carries lists for it
    void* comm = nullptr;
    bool own_comm = false;
    uint8_t* desc_of(int i) { return rows[i].as<uint8_t>() + (size_t)p.max_features * 28; }
    ~uh_fstream() { if (comm && own_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm); }
};

extern "C" {

int uh_fstream_create(uh_ctx* ctx, uh_orb* ext, uh_knn* tile, uh_bow* voc, const uh_fstream_params* p, uh_fstream** out) {
    UH_REQUIRE(ctx && ext && tile && p && out, "uh_fstream_create: NULL argument");
    UH_REQUIRE(p->world >= 1 && p->world <= 64 && p->rank >= 0 && p->rank < p->world, "uh_fstream_create: rank %d / world %d", p->rank, p->world);
    UH_REQUIRE(p->max_features >= 1 && p->cand_cap >= 1 && p->nn >= 1 && p->nn <= 64, "uh_fstream_create: bad sizes (max_features %d, cand_cap %d, nn %d)",
               p->max_features, p->cand_cap, p->nn);
    {   // rows per frame = the extractor's feature budget: a mismatch would make the fixed-size message regions disagree with what the extractor writes
        uh_feat_params efp;
        if (uh_orb_get_params(ext, &efp) == UH_OK) UH_REQUIRE(efp.maxFeatures == p->max_features, "uh_fstream_create: max_features %d differs from the extractor's maxFeatures %d",
                                                              p->max_features, efp.maxFeatures);
    }
    uh_fstream* f = new uh_fstream();
    f->ctx = ctx; f->ext = ext; f->tile = tile; f->voc = voc; f->p = *p;
    f->L = layout(p->max_features, p->cand_cap, p->world, voc != nullptr);
    int rc = UH_OK;
    auto fail = [&](int code) { delete f; return code; };
    if (hipSetDevice(ctx->device) != hipSuccess) { uh::set_error("uh_fstream_create: hipSetDevice failed"); return fail(UH_ENODEVICE); }
    if ((rc = f->send.reserve(f->L.total)) || (rc = f->recv.reserve(f->L.total * (size_t)p->world))) return fail(rc);
    for (int i = 0; i < 2; i++) if ((rc = f->rows[i].reserve((size_t)p->max_features * 60 + 64))) return fail(rc);
    if ((rc = f->cnt.reserve(64))) return fail(rc);
    // persistent buffers: cleared once (rows beyond a frame's count are scanned as queries — any bytes do, but they must be initialised)
    (void)hipMemsetAsync(f->send.p, 0, f->send.cap, ctx->stream);
    (void)hipMemsetAsync(f->recv.p, 0, f->recv.cap, ctx->stream);
    for (int i = 0; i < 2; i++) (void)hipMemsetAsync(f->rows[i].p, 0, f->rows[i].cap, ctx->stream);
    (void)hipMemsetAsync(f->cnt.p, 0, f->cnt.cap, ctx->stream);
    *out = f;
    return UH_OK;
}

void uh_fstream_destroy(uh_fstream* f) { delete f; }

size_t uh_fstream_message_bytes(const uh_fstream* f) { return f ? f->L.total : 0; }
void* uh_fstream_send_buffer(uh_fstream* f) { return f ? f->send.p : nullptr; }
void* uh_fstream_recv_buffer(uh_fstream* f) { return f ? f->recv.p : nullptr; }

int uh_fstream_comm_unique_id(uint8_t id_out[128]) {
    UH_REQUIRE(id_out, "uh_fstream_comm_unique_id: NULL");
    int rc = load_rccl();
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id_out);
    if (e) { uh::set_error("ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return UH_ENODEVICE; }
    return UH_OK;
}

int uh_fstream_comm_init(uh_fstream* f, const uint8_t id[128]) {
    UH_REQUIRE(f && id, "uh_fstream_comm_init: NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    Id128 u;
    std::memcpy(u.b, id, 128);
    void* c = nullptr;
    const int e = g_rccl.CommInitRank(&c, f->p.world, u, f->p.rank);
    if (e) { uh::set_error("ncclCommInitRank(rank %d of %d): %s", f->p.rank, f->p.world, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return UH_ENODEVICE; }
    f->comm = c; f->own_comm = true;
    return UH_OK;
}

// ranks of the communicator the stream gathers over, as RCCL reports it (ncclCommCount); 1 without a communicator (world 1 needs none)
int uh_fstream_comm_ranks(uh_fstream* f) {
    UH_REQUIRE(f, "uh_fstream_comm_ranks: NULL");
    if (!f->comm) return 1;
    UH_REQUIRE(g_rccl.CommCount || load_rccl() == UH_OK, "uh_fstream_comm_ranks: librccl not loaded");
    UH_REQUIRE(g_rccl.CommCount, "uh_fstream_comm_ranks: librccl.so lacks ncclCommCount");
    int n = 0;
    const int e = g_rccl.CommCount(f->comm, &n);
    if (e) { uh::set_error("ncclCommCount: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return UH_ENODEVICE; }
    return n;
}

int uh_fstream_set_comm(uh_fstream* f, void* nccl_comm) {
    UH_REQUIRE(f && nccl_comm, "uh_fstream_set_comm: NULL argument");
    int rc = load_rccl();
    if (rc) return rc;
    f->comm = nccl_comm; f->own_comm = false;
    return UH_OK;
}

// step t, this rank's share, written straight into the send buffer
int uh_fstream_local_dev(uh_fstream* f, const uint8_t* d_frame, int w, int h, size_t stride, int level_first, int level_end) {
    UH_REQUIRE(f && d_frame, "uh_fstream_local_dev: NULL argument");
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    const int F = f->p.max_features;
    uint8_t* m = f->send.as<uint8_t>();
    int rc = uh_orb_set_level_range(f->ext, level_first, level_end);
    if (rc) return rc;
    rc = uh_orb_extract_dev(f->ext, d_frame, w, h, stride, stride * (size_t)h, 1, reinterpret_cast<uh_keypoint*>(m + f->L.kps), m + f->L.desc, F,
                            reinterpret_cast<int32_t*>(m + f->L.hdr));
    (void)uh_orb_set_level_range(f->ext, 0, -1);
    if (rc) return rc;
    f->sent_prev = f->have_prev;
    if (f->have_prev) {
        const uint8_t* q = f->desc_of(f->cur);
        // only the first cnt[cur] rows of the block are frame t-1's descriptors (finish rewrites that many): the rows behind them are
        // older frames' and must not be able to overflow a list
        if ((rc = uh_knn_set_valid_rows_dev(f->tile, f->cnt.as<int32_t>() + f->cur))) return rc;
        rc = uh_knn_scan_shard_dev(f->tile, q, F, f->p.nn, -1, reinterpret_cast<uint64_t*>(m + f->L.cand), reinterpret_cast<int32_t*>(m + f->L.counts), f->p.cand_cap);
        (void)uh_knn_set_valid_rows_dev(f->tile, nullptr);   // (the launch has taken the pointer: the caller's index does not keep one into this stream's memory)
        if (rc) return rc;
        if (f->voc) {
            const int b0 = (int)((long long)F * f->p.rank / f->p.world), b1 = (int)((long long)F * (f->p.rank + 1) / f->p.world);
            if (b1 > b0 && (rc = uh_bow_transform_dev(f->voc, q + (size_t)b0 * 32, b1 - b0, f->p.bow_level, reinterpret_cast<uint32_t*>(m + f->L.bow_word),
                                                      reinterpret_cast<float*>(m + f->L.bow_weight), reinterpret_cast<uint32_t*>(m + f->L.bow_node), m + f->L.bow_valid)))
                return rc;
        }
    }
    return UH_OK;   // (header word 0 = the extractor's own count output; the readers clamp it; words 1..3 stay zero)
}

// for hosts that move the messages themselves (another transport; tests that play several ranks on one GPU): rank r's message into this
// rank's receive buffer
int uh_fstream_put_message(uh_fstream* f, int rank, const void* d_message) {
    UH_REQUIRE(f && d_message && rank >= 0 && rank < f->p.world, "uh_fstream_put_message: bad argument");
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    UH_HIP_CHECK(hipMemcpyAsync(f->recv.as<uint8_t>() + (size_t)rank * f->L.total, d_message, f->L.total, hipMemcpyDeviceToDevice, f->ctx->stream));
    return UH_OK;
}

// THE collective of a frame
int uh_fstream_exchange(uh_fstream* f) {
    UH_REQUIRE(f, "uh_fstream_exchange: NULL");
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    if (f->p.world == 1 && !f->comm) {
        UH_HIP_CHECK(hipMemcpyAsync(f->recv.p, f->send.p, f->L.total, hipMemcpyDeviceToDevice, f->ctx->stream));
        return UH_OK;
    }
    UH_REQUIRE(f->comm, "uh_fstream_exchange: no communicator (uh_fstream_comm_init / uh_fstream_set_comm, or move the messages yourself between uh_fstream_send_buffer and uh_fstream_recv_buffer)");
    const int e = g_rccl.AllGather(f->send.p, f->recv.p, f->L.total, /* ncclUint8 */ 1, f->comm, f->ctx->stream);
    if (e) { uh::set_error("ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error"); return UH_ENODEVICE; }
    return UH_OK;
}

// every rank's message (rank order in the receive buffer) -> frame t complete, frame t-1 matched
int uh_fstream_finish_dev(uh_fstream* f, uh_keypoint* d_kps, uint8_t* d_desc, int32_t* d_count, int32_t* d_prev_indices, int32_t* d_prev_distances,
                          int32_t* d_prev_count, uint32_t* d_bow_word, float* d_bow_weight, uint32_t* d_bow_node, uint8_t* d_bow_valid, int32_t* d_overflow) {
    UH_REQUIRE(f && d_overflow, "uh_fstream_finish_dev: NULL argument");
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    const int F = f->p.max_features, W = f->p.world;
    const uint8_t* rv = f->recv.as<uint8_t>();
    UH_HIP_CHECK(hipMemsetAsync(d_overflow, 0, 4, f->ctx->stream));
    int32_t* cnts = f->cnt.as<int32_t>();   // [0], [1]: row count of rows[0], rows[1]
    if (f->sent_prev) {   // the lists in the messages are for rows[cur] (frame t-1); replay them where they lie
        UH_REQUIRE(d_prev_indices && d_prev_distances, "uh_fstream_finish_dev: NULL row buffers");
        int rc = uh_knn_replay_tiles_strided_dev(f->tile, f->desc_of(f->cur), F, f->p.nn, f->p.sorted, -1, reinterpret_cast<const uint64_t*>(rv + f->L.cand), f->L.total / 8,
                                                 reinterpret_cast<const int32_t*>(rv + f->L.counts), f->L.total / 4, W, f->p.cand_cap, d_prev_indices, d_prev_distances, d_overflow);
        if (rc) return rc;
    }
    const int nxt = f->cur ^ 1;
    const int chunks = std::max(1, std::min(16, uh_div_up(F * 8, 256 * 4)));
    UH_LAUNCH(f->ctx, fstream_unpack_kernel, dim3(chunks, W), dim3(256), 0, rv, f->L, W, F, reinterpret_cast<uint32_t*>(d_kps), reinterpret_cast<uint32_t*>(d_desc),
              f->rows[nxt].as<uint32_t>(), reinterpret_cast<uint32_t*>(f->desc_of(nxt)), d_count, cnts + nxt, d_bow_word, d_bow_weight, d_bow_node, d_bow_valid,
              (f->sent_prev && f->voc) ? 1 : 0, d_overflow, f->sent_prev ? cnts + f->cur : nullptr, d_prev_count);
    UH_HIP_CHECK(hipGetLastError());
    f->cur = nxt;
    f->have_prev = true;
    return UH_OK;
}

}  // extern "C"
