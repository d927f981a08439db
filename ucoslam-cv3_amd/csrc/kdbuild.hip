// Frame::create_kdtree on MI355X (gfx950): the launch that builds picoflann's kd-tree of a frame's undistorted keypoints in HBM
// (src/utils/frameextractor.cpp:4258, src/basictypes/picoflann.h:150-163,238-345) and the device-resident frame object around it.
// The algorithm is in kdbuild.hpp; this unit owns the kernel, the uh_dev_frame object and the test hooks.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "devframe.hpp"

namespace {

__device__ __forceinline__ int kd_point_count(int n_direct, const int* __restrict__ level_counts, int nlevels, int cap) {
    if (!level_counts) return n_direct;
    int t = 0;   // the extractor's per-level counts (select_kernel), clipped exactly as describe_kernel clips its slots
    for (int l = 0; l < nlevels; l++) t += level_counts[l];
    return t < cap ? t : cap;
}

// SPLIT = false: the whole build in this launch.  SPLIT = true: the first of three launches — it may still finish a small or shallow tree alone
// (dump->nsub = 0 then, and the two launches behind it return at once).
template <bool SPLIT>
__global__ __launch_bounds__(512) void kd_build_kernel(const float4* __restrict__ in, int n_direct, const int* __restrict__ level_counts, int nlevels,
                                                        int cap, int n_cap, uh_kd::Node24* __restrict__ nodes, float4* __restrict__ leaf, uh_kd::Meta* meta,
                                                        unsigned long long word, long long* clk, uh_kd::TopDump* dump, float4* pts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_kd[];
    const int n = kd_point_count(n_direct, level_counts, nlevels, cap);
    if (n > n_cap || n < 0) {   // cannot happen through the entry points (n_cap = the extractor's maxFeatures); refuse loudly instead of overrunning LDS
        if (threadIdx.x == 0) {
            if (SPLIT) dump->nsub = 0;
            meta->n = n; meta->n_nodes = -1; meta->max_depth = 0;
            __hip_atomic_store(&meta->word, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    uh_kd::build_workgroup<SPLIT ? uh_kd::kTop : uh_kd::kFull>(s_kd, n_cap, in, n, nodes, leaf, meta, word, clk, dump, pts, uh_kd::SubArgs{});
}

// second launch: workgroup j builds the subtree under the j-th node of the level kd_build_kernel<true> stopped at, on a compute unit of its own
__global__ __launch_bounds__(256) void kd_sub_kernel(const float4* __restrict__ in, int n_cap, const uh_kd::TopDump* __restrict__ dump, const float4* __restrict__ pts,
                                                      uh_kd::Node24* __restrict__ sub_nodes, int node_stride, float4* __restrict__ leaf, uh_kd::SubSum* __restrict__ sums,
                                                      long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_kd[];
    const int j = blockIdx.x;
    if (j >= dump->nsub) return;
    const uh_kd::TopNode t = dump->node[dump->lvl_b + j];
    uh_kd::SubArgs a;
    a.pts = pts; a.pos0 = (int)(t.nbe & 0xffffu); a.root_flags = t.flag; a.depth0 = dump->depth; a.exact = dump->exact; a.sum = sums + j;
    const int c = (int)(t.nbe >> 16) - a.pos0;
    uh_kd::build_workgroup<uh_kd::kSub>(s_kd, n_cap, in, c, sub_nodes + (size_t)j * node_stride, leaf, nullptr, 0ull, j == 0 ? clk : nullptr, nullptr, nullptr, a);
}

// third launch: numbers across the subtrees, the records to their places, meta and the completion word
__global__ __launch_bounds__(256) void kd_join_kernel(const uh_kd::TopDump* __restrict__ dump, const uh_kd::SubSum* __restrict__ sums, const uh_kd::Node24* __restrict__ sub_nodes,
                                                       int node_stride, uh_kd::Node24* __restrict__ nodes, uh_kd::Meta* meta, unsigned long long word) {
    uh_kd::join_workgroup(dump, sums, sub_nodes, node_stride, nodes, meta, word);
}

__global__ void kd_pack_xy_kernel(const float2* __restrict__ xy, int n, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float4(xy[i].x, xy[i].y, 0.f, 0.f);
}

}  // namespace

namespace uh {

int dev_frame_reserve(uh_dev_frame* f, int n_cap) {
    UH_REQUIRE(n_cap >= 1 && n_cap <= uh_kd::kDevMaxPoints, "device frame: %d keypoints exceed the device kd-tree builder's %d (use uh_projmatch_set_frame)", n_cap, uh_kd::kDevMaxPoints);
    if (n_cap <= f->n_cap) return UH_OK;
    static const int env_threads = [] { const char* e = getenv("UH_KD_THREADS"); const int v = e ? atoi(e) : 0; return (v == 256 || v == 512) ? v : 0; }();
    if (env_threads) { f->threads = env_threads; f->split = false; }
    static const bool env_single = [] { const char* e = getenv("UH_KD_SPLIT"); return e && e[0] == '0'; }();
    if (env_single) f->split = false;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t m = (size_t)uh_kd::node_cap(n_cap, 16);
    f->o_desc = 0;
    f->o_in = al(f->o_desc + 32 * (size_t)n_cap);
    f->o_nodes = al(f->o_in + 16 * (size_t)n_cap);
    f->o_leaf = al(f->o_nodes + sizeof(uh_kd::Node24) * m);
    f->o_pts = al(f->o_leaf + 16 * (size_t)n_cap);
    f->o_dump = al(f->o_pts + 16 * (size_t)n_cap);
    f->o_sums = al(f->o_dump + sizeof(uh_kd::TopDump));
    f->o_sub = al(f->o_sums + sizeof(uh_kd::SubSum) * uh_kd::kSubMax);
    f->sub_stride = (int)m;
    const size_t total = al(f->o_sub + sizeof(uh_kd::Node24) * m * uh_kd::kSubMax);
    int rc = f->buf.reserve(total);
    if (rc) return rc;
    if ((rc = f->meta.reserve(sizeof(uh_kd::Meta) + 64))) return rc;
    f->n_cap = n_cap;
    return UH_OK;
}

int kd_build_launch(uh_dev_frame* f, const int* d_level_counts, int nlevels, int cap, int n_direct) {
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    const int top_threads = f->split ? 512 : f->threads;
    const size_t lds = uh_kd::lds_bytes(f->n_cap, top_threads / 64), lds_sub = uh_kd::lds_bytes(f->n_cap, 4);
    UH_REQUIRE(lds <= 160 * 1024 - 2048 && lds_sub <= 160 * 1024 - 2048, "device kd-tree builder: %zu bytes of LDS for %d keypoints", lds, f->n_cap);
    if (!f->attr_set) {
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kd_build_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kd_build_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kd_sub_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        f->attr_set = true;
    }
    static const bool want_clk = getenv("UH_KD_CLK") != nullptr;
    if (want_clk && !f->d_clk.p) { int rc = f->d_clk.reserve(2 * 192 * 8); if (rc) return rc; }
    if (f->d_clk.p) UH_HIP_CHECK(hipMemsetAsync(f->d_clk.p, 0, 2 * 192 * 8, f->ctx->stream));
    const unsigned long long word = ++f->seq;
    if (f->split) {
        // three launches: the workgroup-wide levels, one workgroup per subtree (each on a compute unit of its own), the join
        UH_LAUNCH(f->ctx, kd_build_kernel<true>, dim3(1), dim3(512), lds, (const float4*)f->kd_in(), n_direct, d_level_counts, nlevels, cap, f->n_cap, f->nodes(), f->leaf(),
                  f->meta.dev<uh_kd::Meta>(), word, f->d_clk.as<long long>(), f->dump(), f->pts());
        UH_LAUNCH(f->ctx, kd_sub_kernel, dim3(uh_kd::kSubMax), dim3(256), lds_sub, (const float4*)f->kd_in(), f->n_cap, (const uh_kd::TopDump*)f->dump(), (const float4*)f->pts(),
                  f->sub_nodes(), f->sub_stride, f->leaf(), f->sums(), f->d_clk.p ? f->d_clk.as<long long>() + 192 : nullptr);
        UH_LAUNCH(f->ctx, kd_join_kernel, dim3(1), dim3(256), 0, (const uh_kd::TopDump*)f->dump(), (const uh_kd::SubSum*)f->sums(), (const uh_kd::Node24*)f->sub_nodes(), f->sub_stride,
                  f->nodes(), f->meta.dev<uh_kd::Meta>(), word);
    } else {
        UH_LAUNCH(f->ctx, kd_build_kernel<false>, dim3(1), dim3(f->threads), lds, (const float4*)f->kd_in(), n_direct, d_level_counts, nlevels, cap, f->n_cap, f->nodes(),
                  f->leaf(), f->meta.dev<uh_kd::Meta>(), word, f->d_clk.as<long long>(), (uh_kd::TopDump*)nullptr, (float4*)nullptr);
    }
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

int dev_frame_wait(uh_dev_frame* f, const uh_kd::Meta** meta, const char* what) {
    UH_REQUIRE(f->seq != 0, "%s: the device frame holds no extraction yet", what);
    int rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(f->meta.host<char>()), f->seq, f->ctx->stream, what);
    if (rc) return rc;
    const uh_kd::Meta* m = f->meta.host<uh_kd::Meta>();
    UH_REQUIRE(m->n_nodes >= 0, "%s: the device kd-tree build refused %d keypoints (capacity %d)", what, m->n, f->n_cap);
    *meta = m;
    return UH_OK;
}

}  // namespace uh

extern "C" {

int uh_dev_frame_create(uh_ctx* ctx, uh_dev_frame** out) {
    UH_REQUIRE(ctx && out, "uh_dev_frame_create: NULL argument");
    uh_dev_frame* f = new uh_dev_frame();
    f->ctx = ctx;
    *out = f;
    return UH_OK;
}

void uh_dev_frame_destroy(uh_dev_frame* f) { delete f; }

// Who builds the frame's kd-tree: 0 (default) the build launches behind the extraction, 1 the host core inside uh_projmatch_set_frame_dev
// (51 us on one core against 82 us of launches for 2000 keypoints: the faster route while a core is free; the frame stays resident either way)
int uh_dev_frame_set_tree_builder(uh_dev_frame* f, int32_t on_host) {
    UH_REQUIRE(f, "uh_dev_frame_set_tree_builder: NULL frame");
    f->host_tree = on_host != 0;
    return UH_OK;
}

// A frame that was not extracted here (read from a file, produced by another extractor) becomes a device frame: descriptors and undistorted
// keypoints are uploaded, the tree is built as for an extracted one (the build launches, or later the host core: uh_dev_frame_set_tree_builder).
// Synchronous (a utility, not the per-frame path).
int uh_dev_frame_upload(uh_dev_frame* f, const uh_keypoint* und_kpts, int32_t n, const uint8_t* desc) {
    UH_REQUIRE(f && n >= 0 && (n == 0 || (und_kpts && desc)), "uh_dev_frame_upload: bad arguments");
    int rc = uh::dev_frame_reserve(f, std::max(n, 1));
    if (rc) return rc;
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    std::vector<float> in(4 * (size_t)std::max(n, 1));
    for (int i = 0; i < n; i++) {
        UH_REQUIRE(und_kpts[i].octave >= 0 && und_kpts[i].octave < 16, "uh_dev_frame_upload: octave %d of keypoint %d outside [0,16)", und_kpts[i].octave, i);
        in[4 * (size_t)i] = und_kpts[i].x; in[4 * (size_t)i + 1] = und_kpts[i].y;
        const int32_t o = und_kpts[i].octave;
        std::memcpy(&in[4 * (size_t)i + 2], &o, 4); in[4 * (size_t)i + 3] = 0.f;
    }
    if (n) {
        UH_HIP_CHECK(hipMemcpyAsync(f->kd_in(), in.data(), 16 * (size_t)n, hipMemcpyHostToDevice, f->ctx->stream));
        UH_HIP_CHECK(hipMemcpyAsync(f->desc(), desc, 32 * (size_t)n, hipMemcpyHostToDevice, f->ctx->stream));
    }
    if (!f->host_tree && (rc = uh::kd_build_launch(f, nullptr, 0, 0, n))) return rc;
    UH_HIP_CHECK(hipStreamSynchronize(f->ctx->stream));
    return UH_OK;
}

// test hook / inspection: the tree of the frame's latest extraction, copied to the host (waits for the build)
int uh_dev_frame_tree(uh_dev_frame* f, int32_t* n_kpts, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out, float* leaf_xy_out,
                      int32_t* leaf_octave_out, double* root_box4, int32_t* max_depth) {
    UH_REQUIRE(f, "uh_dev_frame_tree: NULL frame");
    const uh_kd::Meta* m = nullptr;
    int rc = uh::dev_frame_wait(f, &m, "uh_dev_frame_tree");
    if (rc) return rc;
    if (n_kpts) *n_kpts = m->n;
    if (n_nodes) *n_nodes = m->n_nodes;
    if (max_depth) *max_depth = m->max_depth;
    if (root_box4) std::memcpy(root_box4, m->box, 32);
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    if (nodes24_out && m->n_nodes > 0) UH_HIP_CHECK(hipMemcpy(nodes24_out, f->nodes(), sizeof(uh_kd::Node24) * (size_t)m->n_nodes, hipMemcpyDeviceToHost));
    if ((leaf_idx_out || leaf_xy_out || leaf_octave_out) && m->n > 0) {
        std::vector<float> lr(4 * (size_t)m->n);
        UH_HIP_CHECK(hipMemcpy(lr.data(), f->leaf(), 16 * (size_t)m->n, hipMemcpyDeviceToHost));
        for (int i = 0; i < m->n; i++) {
            uint32_t io;
            std::memcpy(&io, &lr[4 * (size_t)i + 2], 4);
            if (leaf_idx_out) leaf_idx_out[i] = io >> 4;
            if (leaf_octave_out) leaf_octave_out[i] = (int32_t)(io & 15u);
            if (leaf_xy_out) { leaf_xy_out[2 * i] = lr[4 * (size_t)i]; leaf_xy_out[2 * i + 1] = lr[4 * (size_t)i + 1]; }
        }
    }
    return UH_OK;
}

// test hook: the device build of n host points (octave 0), same outputs as uh_kdtree_build_host; threads = 0 (default) / 256 / 512
int uh_kdtree_build_dev(uh_ctx* ctx, const float* xy, int32_t n, int32_t threads, int32_t* n_nodes, void* nodes24_out, uint32_t* leaf_idx_out,
                        double* root_box4, int32_t* max_depth) {
    UH_REQUIRE(ctx && n >= 0 && (n == 0 || xy) && n_nodes && nodes24_out && leaf_idx_out && root_box4, "uh_kdtree_build_dev: bad arguments");
    UH_REQUIRE(threads == 0 || threads == 256 || threads == 512, "uh_kdtree_build_dev: %d threads (256 / 512)", threads);
    uh_dev_frame f;
    f.ctx = ctx;
    int rc = uh::dev_frame_reserve(&f, std::max(n, 1));
    if (rc) return rc;
    if (threads) { f.threads = threads; f.split = false; }
    UH_HIP_CHECK(hipSetDevice(ctx->device));
    uh::DevBuf d_xy;
    if ((rc = d_xy.reserve(8 * (size_t)std::max(n, 1)))) return rc;
    if (n) {
        UH_HIP_CHECK(hipMemcpyAsync(d_xy.p, xy, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        UH_LAUNCH(ctx, kd_pack_xy_kernel, dim3(uh_div_up(n, 256)), dim3(256), 0, (const float2*)d_xy.p, n, f.kd_in());
    }
    if ((rc = uh::kd_build_launch(&f, nullptr, 0, 0, n))) return rc;
    rc = uh_dev_frame_tree(&f, nullptr, n_nodes, nodes24_out, leaf_idx_out, nullptr, nullptr, root_box4, max_depth);
    UH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (f.d_clk.p) {   // UH_KD_CLK=1: wall-clock stamps (10 ns units) of workgroup thread 0 (split build: of the first launch, then of subtree 0's workgroup)
        long long cc[2 * 192];
        UH_HIP_CHECK(hipMemcpy(cc, f.d_clk.p, sizeof(cc), hipMemcpyDeviceToHost));
        for (int half = 0; half < (f.split ? 2 : 1); half++) {
            const long long* c = cc + 192 * half;
            const int nw = half ? 4 : (f.split ? 8 : f.threads / 64);
            fprintf(stderr, "kd build n=%d %s [us]: load %.2f  wg-levels %.2f  wave-levels(w0) %.2f  join %.2f  zero %.2f  climb %.2f  nodes %.2f  leaves %.2f  total %.2f\n", n,
                    f.split ? (half ? "subtree 0" : "top launch") : (f.threads == 512 ? "512 threads" : "256 threads"), (c[1] - c[0]) * 0.01, (c[2] - c[1]) * 0.01, (c[3] - c[2]) * 0.01,
                    (c[4] - c[3]) * 0.01, (c[5] - c[4]) * 0.01, (c[6] - c[5]) * 0.01, (c[7] - c[6]) * 0.01, (c[8] - c[7]) * 0.01, (c[8] - c[0]) * 0.01);
            fprintf(stderr, "  waves done after the hand-over [us] (points):");
            for (int w = 0; w < nw; w++) fprintf(stderr, " %.1f (%lld)", (c[128 + w] - c[2]) * 0.01, c[144 + w]);
            fprintf(stderr, "\n");
            for (int part = 0; part < 2; part++)
                for (int lv = 0; lv < 6; lv++) {
                    const long long* q = c + (part ? 64 : 16) + lv * 8;
                    if (!q[0] || !q[6]) continue;
                    fprintf(stderr, "  %s level %d: moments %.2f  pass0 %.2f  pass1 %.2f  split %.2f  fallback %.2f  children %.2f\n", part ? "wave0" : "wg", lv, (q[1] - q[0]) * 0.01,
                            ((q[2] ? q[2] : q[3]) - q[1]) * 0.01, (q[2] ? (q[3] - q[2]) * 0.01 : 0.0), (q[4] - q[3]) * 0.01, (q[5] - q[4]) * 0.01, (q[6] - q[5]) * 0.01);
                }
        }
    }
    return rc;
}

// test hook, host only: the permutation libstdc++'s std::sort leaves on n float keys, computed the way the device builder computes it
// (uh_kd::sort_phase + a stable rank) — tests compare it with std::sort itself (oracle_std_sort_perm)
int uh_kdtree_sort_restated_host(const float* keys, int32_t n, uint32_t* perm_out) {
    UH_REQUIRE(n >= 0 && (n == 0 || (keys && perm_out)), "uh_kdtree_sort_restated_host: bad arguments");
    struct Acc {
        std::vector<uh_kd::Elem> v;
        float key(int i) const { return v[i].x; }
        float ekey(const uh_kd::Elem& e) const { return e.x; }
        uh_kd::Elem get(int i) const { return v[i]; }
        void set(int i, const uh_kd::Elem& e) { v[i] = e; }
        void swap(int i, int j) { std::swap(v[i], v[j]); }
    } a;
    a.v.resize(n);
    for (int i = 0; i < n; i++) a.v[i] = uh_kd::Elem{keys[i], 0.f, (uint32_t)i};
    uh_kd::sort_phase(a, 0, n);
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int j = 0; j < n; j++) r += (a.v[j].x < a.v[i].x || (a.v[j].x == a.v[i].x && j < i)) ? 1 : 0;
        perm_out[r] = a.v[i].id;
    }
    return UH_OK;
}

}  // extern "C"
