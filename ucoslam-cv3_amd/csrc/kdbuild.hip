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

__global__ __launch_bounds__(512) void kd_build_kernel(const float4* __restrict__ in, int n_direct, const int* __restrict__ level_counts, int nlevels,
                                                        int cap, int n_cap, uh_kd::Node24* __restrict__ nodes, float4* __restrict__ leaf, uh_kd::Meta* meta,
                                                        unsigned long long word, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_kd[];
    int n = n_direct;
    if (level_counts) {   // the extractor's per-level counts (select_kernel), clipped exactly as describe_kernel clips its slots
        int t = 0;
        for (int l = 0; l < nlevels; l++) t += level_counts[l];
        n = t < cap ? t : cap;
    }
    if (n > n_cap || n < 0) {   // cannot happen through the entry points (n_cap = the extractor's maxFeatures); refuse loudly instead of overrunning LDS
        if (threadIdx.x == 0) {
            meta->n = n; meta->n_nodes = -1; meta->max_depth = 0;
            __hip_atomic_store(&meta->word, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    uh_kd::build_workgroup(s_kd, n_cap, in, n, nodes, leaf, meta, word, clk);
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
    if (env_threads) f->threads = env_threads;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t m = (size_t)uh_kd::node_cap(n_cap, 16);
    f->o_desc = 0;
    f->o_in = al(f->o_desc + 32 * (size_t)n_cap);
    f->o_nodes = al(f->o_in + 16 * (size_t)n_cap);
    f->o_leaf = al(f->o_nodes + sizeof(uh_kd::Node24) * m);
    const size_t total = al(f->o_leaf + 16 * (size_t)n_cap);
    int rc = f->buf.reserve(total);
    if (rc) return rc;
    if ((rc = f->meta.reserve(sizeof(uh_kd::Meta) + 64))) return rc;
    f->n_cap = n_cap;
    return UH_OK;
}

int kd_build_launch(uh_dev_frame* f, const int* d_level_counts, int nlevels, int cap, int n_direct) {
    UH_HIP_CHECK(hipSetDevice(f->ctx->device));
    const size_t lds = uh_kd::lds_bytes(f->n_cap, f->threads / 64);
    UH_REQUIRE(lds <= 160 * 1024 - 2048, "device kd-tree builder: %zu bytes of LDS for %d keypoints", lds, f->n_cap);
    if (!f->attr_set) {
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kd_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        f->attr_set = true;
    }
    static const bool want_clk = getenv("UH_KD_CLK") != nullptr;
    if (want_clk && !f->d_clk.p) { int rc = f->d_clk.reserve(192 * 8); if (rc) return rc; }
    if (f->d_clk.p) UH_HIP_CHECK(hipMemsetAsync(f->d_clk.p, 0, 192 * 8, f->ctx->stream));
    const unsigned long long word = ++f->seq;
    UH_LAUNCH(f->ctx, kd_build_kernel, dim3(1), dim3(f->threads), lds, (const float4*)f->kd_in(), n_direct, d_level_counts, nlevels, cap, f->n_cap, f->nodes(),
              f->leaf(), f->meta.dev<uh_kd::Meta>(), word, f->d_clk.as<long long>());
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
    if (threads) f.threads = threads;
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
    if (f.d_clk.p) {   // UH_KD_CLK=1: wall-clock stamps (10 ns units) of workgroup thread 0
        long long c[192];
        UH_HIP_CHECK(hipMemcpy(c, f.d_clk.p, sizeof(c), hipMemcpyDeviceToHost));
        fprintf(stderr, "kd build n=%d threads=%d [us]: load %.2f  wg-levels %.2f  wave-levels(w0) %.2f  join %.2f  zero %.2f  climb %.2f  nodes %.2f  leaves %.2f  total %.2f\n", n, f.threads,
                (c[1] - c[0]) * 0.01, (c[2] - c[1]) * 0.01, (c[3] - c[2]) * 0.01, (c[4] - c[3]) * 0.01, (c[5] - c[4]) * 0.01, (c[6] - c[5]) * 0.01, (c[7] - c[6]) * 0.01, (c[8] - c[7]) * 0.01, (c[8] - c[0]) * 0.01);
        fprintf(stderr, "  waves done after the hand-over [us] (points):");
        for (int w = 0; w < f.threads / 64; w++) fprintf(stderr, " %.1f (%lld)", (c[128 + w] - c[2]) * 0.01, c[144 + w]);
        fprintf(stderr, "\n");
        for (int part = 0; part < 2; part++)
            for (int lv = 0; lv < 6; lv++) {
                const long long* q = c + (part ? 64 : 16) + lv * 8;
                if (!q[0] || !q[6]) continue;
                fprintf(stderr, "  %s level %d: moments %.2f  pass0 %.2f  pass1 %.2f  split %.2f  fallback %.2f  children %.2f\n", part ? "wave0" : "wg", lv, (q[1] - q[0]) * 0.01, ((q[2] ? q[2] : q[3]) - q[1]) * 0.01,
                        (q[2] ? (q[3] - q[2]) * 0.01 : 0.0), (q[4] - q[3]) * 0.01, (q[5] - q[4]) * 0.01, (q[6] - q[5]) * 0.01);
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
