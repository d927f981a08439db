// uh_track_pose: the tracker's pose estimation for one frame as ONE host call — what System does between FrameExtractor::process and the
// keyframe decision (src/utils/system.cpp, line numbers = statement starts after preprocessing):
//   :5930-6460 / :6559-6565   projection search against the previous frame            (uh_projmatch_match_prev)
//   :6566-                    PnPSolver::solvePnp over those matches                   (uh_pnp_solve; pnpsolver.cpp:116-409, look-ups :199-232)
//   :6762-6881                >= 30 inliers: keep the refined pose, search the local map in a 4 px disc; else predicted pose, projDistThr
//                             Map::matchFrameToMapPoints                               (uh_projmatch_match)
//   :6897-6954                inliers of the first set + the new matches, filter_ambiguous_query over the union, per-match look-ups
//                             PnPSolver::solvePnp                                      (uh_pnp_solve)
// Called one after the other through the C ABI the four operators cost four host round trips (launch latency + completion word + unpacking +
// the host's look-ups in between: ~55 of the frame's ~430 us were no kernel's).  Here the host stages both candidate sets, enqueues
// SIX launches on the context stream and waits once; what the host did between the calls runs on the device:
//   projmatch_kernel<prev>  ->  track_select_kernel (matches in item order, filter_ambiguous_query, look-ups for the solve)
//   -> pnp_solve_kernel (match count from device memory; its last thread also takes the decision: pose / radius of the map search)
//   -> projmatch_kernel<map> (pose and radius read from device memory)  ->  track_select_kernel (its own filter, the union with the
//   first solve's inliers, filter, look-ups)  ->  pnp_solve_kernel  ->  track_publish_kernel (everything the sequence of calls returns).
// The look-ups gather from HBM only: the searches leave every candidate's position and its {id, map row, weight} record there again
// (PmPoints::pos_out / aux_out: one more 16-byte read per candidate inside launches that wait on such reads anyway; gathering from pinned
// memory in the select launches was one host-link request per match — 31 us for the second one — and staging the tables there 5 us).
// Same results as the four calls, bit for bit (tests/test_track.py, tests/test_cpp_host.py): the same kernels do the searching and
// the solving; the list logic is integer work (stable minimum per keypoint = uh_filter_ambiguous, matcher.hip keep_best_per_key).
// This file is included at the end of projmatch.hip (it uses that unit's internals).
#pragma once

namespace uh {
int pnp_enqueue_dev(uh_pnp* p, const float* d_pose, const float* d_intr4, int n_cap, const int* d_n, const float* d_p3d, const float* d_kp, const float* d_inv_sigma,
                    const float* d_weight, float* d_pose_out, unsigned char* d_bad_out, int* d_result5, const PnpDecide* dec);
uh_ctx* pnp_ctx(uh_pnp* p);
}

namespace {

constexpr int kTrkThreads = 1024;
struct TrkElem { int query; unsigned id; float dist; int src; };   // src: bit 30 = candidate of the map search, low bits = its index in its candidate set

// header of the device block (ints)
enum : int { kTrkN1 = 0, kTrkN2 = 1, kTrkNA = 2, kTrkTracked = 3, kTrkRes1 = 4 /* 5 ints */, kTrkRes2 = 9 /* 5 ints */, kTrkHdrInts = 16 };

struct TrkSelect {
    // the search whose results become a list (candidate order)
    int nB; const int* bk; const float* bd; const uint4* aux; int map_kind;   // aux: {id, map row, weight} per candidate (HBM, left by the search)
    // second form only: the list carried over from the first solve (its inliers enter the union when the frame counts as tracked)
    const uh_dmatch* carry; const int* carry_src; const unsigned char* carry_bad; int carry_cap;
    int* hdr;                       // counts in / out (see the enum)
    int fresh_n_slot, final_n_slot; // hdr slots of the fresh list's and the final list's length (equal in the first form)
    uh_dmatch* fresh_out; int* fresh_src;   // the search's own matches (after filter_ambiguous_query)
    uh_dmatch* final_out;                   // second form: the union after its filter (NULL in the first form)
    int* final_src;                         // scratch: source of each element of the final list
    // look-ups for the solve over the final list
    const float4* pos_prev; const float4* pos_map;   // HBM: the candidates' positions as the searches left them
    const uint4* aux_prev; const uint4* aux_map; int prefer_map_row;
    const float4* kp_xyo; const float* inv_sigma_lv; int n_levels;
    float* p3d; float* kp; float* isg; float* wgt;
    int n_kpts;
    TrkElem* scratch_a; TrkElem* scratch_b;
    long long* clk;   // UH_TRK_CLK: 8 wall-clock stamps (10 ns) of thread 0
};

// Exclusive prefix over the workgroup's per-thread counts; returns the thread's offset and (total) the sum.  Two barriers.
__device__ __forceinline__ int trk_block_offset(int count, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = uh_kd::wave_incl_scan(count);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kTrkThreads / 64; w++) { const int v = s_wave[w]; base += w < wave ? v : 0; tot += v; }
    __syncthreads();
    total = tot;
    return base + inc - count;
}

constexpr int kTrkItems = 4;   // consecutive list positions per thread and pass: lists of up to 4096 elements take ONE scan (two barriers of sixteen waves) per step

// uh_filter_ambiguous(list, n, by query) on the device: the first strict minimum of the distance per keypoint stays, in list order
// (matcher.hip keep_best_per_key).  in / out may not alias.  Returns the new length (every thread).
__device__ int trk_filter(const TrkElem* in, int n, TrkElem* out, unsigned long long* s_best, int n_kpts, int* s_wave) {
    for (int k = threadIdx.x; k < n_kpts; k += kTrkThreads) s_best[k] = ~0ull;
    __syncthreads();
    for (int pos = threadIdx.x; pos < n; pos += kTrkThreads) {
        const TrkElem e = in[pos];
        atomicMin(&s_best[e.query], ((unsigned long long)__float_as_uint(e.dist) << 32) | (unsigned)pos);   // distances are >= 0: their bit patterns order like the values
    }
    __syncthreads();
    int kept = 0;
    for (int p0 = 0; p0 < n; p0 += kTrkThreads * kTrkItems) {
        TrkElem e[kTrkItems];
        int keep[kTrkItems], cnt = 0;
#pragma unroll
        for (int u = 0; u < kTrkItems; u++) {
            const int pos = p0 + (int)threadIdx.x * kTrkItems + u;
            keep[u] = 0;
            if (pos < n) { e[u] = in[pos]; keep[u] = (unsigned)(s_best[e[u].query] & 0xffffffffull) == (unsigned)pos ? 1 : 0; }
            cnt += keep[u];
        }
        int tot;
        int o = kept + trk_block_offset(cnt, s_wave, tot);
#pragma unroll
        for (int u = 0; u < kTrkItems; u++) if (keep[u]) out[o++] = e[u];
        kept += tot;
    }
    __syncthreads();
    return kept;
}

// LDS_LISTS: the two working lists live in dynamic LDS (cap elements each) instead of the HBM scratch (round trips through L2 between the steps)
template <bool LDS_LISTS>
__global__ __launch_bounds__(kTrkThreads) void track_select_kernel(TrkSelect a, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ unsigned long long s_best[4096];
    __shared__ int s_wave[kTrkThreads / 64];
    __shared__ float s_isl[16];
    if (threadIdx.x < 16) s_isl[threadIdx.x] = (int)threadIdx.x < a.n_levels ? a.inv_sigma_lv[threadIdx.x] : 0.f;   // (pinned memory: read once, not once per match)
    TrkElem* const LA = LDS_LISTS ? reinterpret_cast<TrkElem*>(s_dyn) : a.scratch_a;
    TrkElem* const LB = LDS_LISTS ? reinterpret_cast<TrkElem*>(s_dyn) + cap : a.scratch_b;
    const int tid = threadIdx.x;
#define UH_TRK_STAMP(j) do { if (a.clk && tid == 0) a.clk[j] = wall_clock64(); } while (0)
    UH_TRK_STAMP(0);
    UH_TRK_STAMP(1);
    // ---- the search's hits in candidate order
    int nF = 0;
    for (int p0 = 0; p0 < a.nB; p0 += kTrkThreads * kTrkItems) {
        int kpi[kTrkItems], cnt = 0;
        unsigned id[kTrkItems];
        float dist[kTrkItems];
#pragma unroll
        for (int u = 0; u < kTrkItems; u++) {
            const int i = p0 + tid * kTrkItems + u;
            kpi[u] = -1; id[u] = 0; dist[u] = 0.f;
            if (i < a.nB) { kpi[u] = a.bk[i]; id[u] = a.aux[i].x; dist[u] = a.bd[i]; }
            cnt += kpi[u] >= 0 ? 1 : 0;
        }
        int tot;
        int o = nF + trk_block_offset(cnt, s_wave, tot);
#pragma unroll
        for (int u = 0; u < kTrkItems; u++) if (kpi[u] >= 0) LA[o++] = TrkElem{kpi[u], id[u], dist[u], (a.map_kind << 30) | (p0 + tid * kTrkItems + u)};
        nF += tot;
    }
    __syncthreads();
    UH_TRK_STAMP(2);
    // ---- filter_ambiguous_query of the search's own list (uh_projmatch_match / _match_prev end with it)
    const int n_fresh = trk_filter(LA, nF, LB, s_best, a.n_kpts, s_wave);
    UH_TRK_STAMP(3);
    for (int p = tid; p < n_fresh; p += kTrkThreads) {
        const TrkElem e = LB[p];
        a.fresh_out[p] = uh_dmatch{e.query, (int)e.id, -1, e.dist};
        a.fresh_src[p] = e.src;
    }
    const TrkElem* fin = LB;
    int n_fin = n_fresh;
    UH_TRK_STAMP(4);
    if (a.carry) {
        // ---- the union: inliers of the first solve (when the frame counts as tracked), then the new matches; filter again
        const int tracked = a.hdr[kTrkTracked];
        const int nc = tracked ? min(a.hdr[kTrkN1], a.carry_cap) : 0;
        int nU = 0;
        for (int p0 = 0; p0 < nc; p0 += kTrkThreads * kTrkItems) {
            int flag[kTrkItems], cnt = 0;
#pragma unroll
            for (int u = 0; u < kTrkItems; u++) { const int p = p0 + tid * kTrkItems + u; flag[u] = p < nc && !a.carry_bad[p] ? 1 : 0; cnt += flag[u]; }
            int tot;
            int o = nU + trk_block_offset(cnt, s_wave, tot);
#pragma unroll
            for (int u = 0; u < kTrkItems; u++)
                if (flag[u]) { const int p = p0 + tid * kTrkItems + u; const uh_dmatch m = a.carry[p]; LA[o++] = TrkElem{m.queryIdx, (unsigned)m.trainIdx, m.distance, a.carry_src[p]}; }
            nU += tot;
        }
        __syncthreads();
        for (int p = tid; p < n_fresh; p += kTrkThreads) LA[nU + p] = LB[p];
        __syncthreads();
        nU += n_fresh;
        // (LB is free again: its content lives in LA now)
        n_fin = trk_filter(LA, nU, LB, s_best, a.n_kpts, s_wave);
        for (int p = tid; p < n_fin; p += kTrkThreads) { const TrkElem e = LB[p]; a.final_out[p] = uh_dmatch{e.query, (int)e.id, -1, e.dist}; }
    }
    UH_TRK_STAMP(5);
    // ---- the solver's per-match look-ups (pnpsolver.cpp:199-232): the point's coordinates and weight, the keypoint, 1 / scaleFactor of its octave
    for (int p = tid; p < n_fin; p += kTrkThreads) {
        const TrkElem e = fin[p];
        const int is_map = (e.src >> 30) & 1, idx = e.src & 0x3fffffff;
        const uint4 ax = is_map ? a.aux_map[idx] : a.aux_prev[idx];
        const int row = a.prefer_map_row ? (int)ax.y : -1;   // (first solve: the candidate's own position, weight 1)
        const float4 pos = row >= 0 ? a.pos_map[row] : a.pos_prev[idx];
        a.p3d[3 * p] = pos.x; a.p3d[3 * p + 1] = pos.y; a.p3d[3 * p + 2] = pos.z;
        a.wgt[p] = row >= 0 ? __uint_as_float(ax.z) : 1.f;
        const float4 k = a.kp_xyo[e.query];
        a.kp[2 * p] = k.x; a.kp[2 * p + 1] = k.y;
        const int oct = (int)(__float_as_uint(k.z) & 15u);
        a.isg[p] = s_isl[oct < a.n_levels ? oct : 0];
        if (a.final_src) a.final_src[p] = e.src;
    }
    if (tid == 0) { a.hdr[a.fresh_n_slot] = n_fresh; a.hdr[a.final_n_slot] = n_fin; }
    UH_TRK_STAMP(6);
#undef UH_TRK_STAMP
}

struct TrkPublish {
    const int* hdr; const float* pose1; const float* pose2;
    const uh_dmatch* m1; const unsigned char* bad1; const uh_dmatch* m2; const uh_dmatch* ma; const unsigned char* bad2;
    int cap1, cap2, capa;
    // pinned twins
    int* h_hdr; float* h_pose1; float* h_pose2; uh_dmatch* h_m1; unsigned char* h_bad1; uh_dmatch* h_m2; uh_dmatch* h_ma; unsigned char* h_bad2;
    unsigned long long* host_done; unsigned long long word;
};

__global__ __launch_bounds__(kTrkThreads) void track_publish_kernel(TrkPublish p) {
    const int tid = threadIdx.x;
    const int n1 = min(p.hdr[kTrkN1], p.cap1), n2 = min(p.hdr[kTrkN2], p.cap2), na = min(p.hdr[kTrkNA], p.capa);
    if (tid < kTrkHdrInts) p.h_hdr[tid] = p.hdr[tid];
    if (tid < 16) { p.h_pose1[tid] = p.pose1[tid]; p.h_pose2[tid] = p.pose2[tid]; }
    const uint4* s; uint4* d;
    s = reinterpret_cast<const uint4*>(p.m1); d = reinterpret_cast<uint4*>(p.h_m1);
    for (int i = tid; i < n1; i += kTrkThreads) d[i] = s[i];
    s = reinterpret_cast<const uint4*>(p.m2); d = reinterpret_cast<uint4*>(p.h_m2);
    for (int i = tid; i < n2; i += kTrkThreads) d[i] = s[i];
    s = reinterpret_cast<const uint4*>(p.ma); d = reinterpret_cast<uint4*>(p.h_ma);
    for (int i = tid; i < na; i += kTrkThreads) d[i] = s[i];
    for (int i = tid; i < n1; i += kTrkThreads) p.h_bad1[i] = p.bad1[i];
    for (int i = tid; i < na; i += kTrkThreads) p.h_bad2[i] = p.bad2[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every thread: its stores into pinned memory before the word
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.host_done, p.word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct uh_track_state {
    uh::DevBuf d;          // header | poses | PmDyn | lists | solver arrays | scratch
    uh::MappedBuf h_par;   // pinned: [completion word | pose0 | intr | inv sigma per level | candidate ids | prev_map_row | map weights]
    uh::MappedBuf h_out;   // pinned: the results
    unsigned long long seq = 0;
    bool attr_set = false;
    uh::DevBuf d_clk;      // UH_TRK_CLK (measurement)
};

uh_projmatch::~uh_projmatch() { delete track; }

extern "C" {

int uh_track_pose(uh_projmatch* h, uh_pnp* pnp, const uh_track_args* a, uh_track_result* r) {
    UH_REQUIRE(h && pnp && a && r, "uh_track_pose: NULL argument");
    UH_REQUIRE(h->have_frame && h->dev, "uh_track_pose: needs a device-resident frame (uh_orb_extract_frame_dev + uh_projmatch_set_frame_dev)");
    UH_REQUIRE(a->pose0 && a->intr4 && a->prev && a->map && a->inv_sigma_levels, "uh_track_pose: NULL input");
    UH_REQUIRE(a->n_levels >= 1 && a->n_levels <= 16, "uh_track_pose: %d levels", a->n_levels);
    const int np = a->prev->n, nm = a->map->n, nk = h->n_kpts;
    UH_REQUIRE(np >= 0 && nm >= 0, "uh_track_pose: negative candidate count");
    UH_REQUIRE(nk <= 4096, "uh_track_pose: %d keypoints exceed the list filter's 4096", nk);
    UH_REQUIRE(a->prev_max_repj_dist > 0 && a->map_radius_tracked > 0 && a->map_radius_lost > 0, "uh_track_pose: search radii must be > 0");
    if (np) UH_REQUIRE(a->prev->ids && a->prev->pos3d && a->prev->octave && a->prev->desc, "uh_track_pose: previous-frame arrays missing");
    if (nm) UH_REQUIRE(a->map->ids && a->map->pos3d && a->map->normal && a->map->min_dist && a->map->max_dist && a->map->desc, "uh_track_pose: map point arrays missing");
    for (int i = 0; i < np; i++)
        UH_REQUIRE(a->prev->octave[i] >= 0 && a->prev->octave[i] < h->n_levels, "uh_track_pose: octave %d of item %d outside [0,%d)", a->prev->octave[i], i, h->n_levels);
    UH_REQUIRE(r->matches_prev && r->bad_prev && r->matches_map && r->matches_all && r->bad_all, "uh_track_pose: output buffers missing");
    UH_REQUIRE(h->ctx == uh::pnp_ctx(pnp), "uh_track_pose: matcher and solver belong to different contexts");
    UH_HIP_CHECK(hipSetDevice(h->ctx->device));
    if (!h->track) h->track = new uh_track_state();
    uh_track_state& T = *h->track;
    int rc;
    const int cap1 = std::max(np, 1), cap2 = std::max(nm, 1), capa = std::max(np + nm, 1), capn = std::min(capa, std::max(nk, 1));   // (a filtered list holds one match per keypoint at most)
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // ---- device block
    size_t o = 0;
    const size_t o_hdr = o; o = al(o + 4 * kTrkHdrInts);
    const size_t o_pose1 = o; o += 64; const size_t o_posem = o; o += 64; const size_t o_pose2 = o; o += 64;
    const size_t o_dyn = o; o = al(o + sizeof(PmDyn));
    const size_t o_m1 = o; o = al(o + 16 * (size_t)cap1); const size_t o_src1 = o; o = al(o + 4 * (size_t)cap1); const size_t o_bad1 = o; o = al(o + (size_t)cap1);
    const size_t o_m2 = o; o = al(o + 16 * (size_t)cap2); const size_t o_src2 = o; o = al(o + 4 * (size_t)cap2);
    const size_t o_ma = o; o = al(o + 16 * (size_t)capa); const size_t o_srca = o; o = al(o + 4 * (size_t)capa); const size_t o_bada = o; o = al(o + (size_t)capa);
    const size_t o_p3d = o; o = al(o + 12 * (size_t)capa); const size_t o_kp = o; o = al(o + 8 * (size_t)capa); const size_t o_isg = o; o = al(o + 4 * (size_t)capa);
    const size_t o_wgt = o; o = al(o + 4 * (size_t)capa);
    const size_t o_sa = o; o = al(o + sizeof(TrkElem) * (size_t)capa); const size_t o_sb = o; o = al(o + sizeof(TrkElem) * (size_t)capa);
    const size_t o_posp = o; o = al(o + 16 * (size_t)cap1); const size_t o_posm = o; o = al(o + 16 * (size_t)cap2);
    const size_t o_auxp = o; o = al(o + 16 * (size_t)cap1); const size_t o_auxm = o; o = al(o + 16 * (size_t)cap2);
    if ((rc = T.d.reserve(o))) return rc;
    char* D = T.d.as<char>();
    // ---- pinned parameter block (read by the launches in place)
    size_t q = 64;
    const size_t q_pose0 = q; q += 64; const size_t q_intr = q; q += 64; const size_t q_isl = q; q += 64;
    if ((rc = T.h_par.reserve(q))) return rc;   // (the previous call's launches are complete: its results were awaited)
    char* hp = T.h_par.host<char>();
    char* dp = T.h_par.dev<char>();
    std::memcpy(hp + q_pose0, a->pose0, 64);
    std::memcpy(hp + q_intr, a->intr4, 16);
    std::memcpy(hp + q_isl, a->inv_sigma_levels, 4 * (size_t)a->n_levels);
    // ---- pinned result block
    size_t w = 0;
    const size_t w_hdr = w; w = al(w + 4 * kTrkHdrInts); const size_t w_pose1 = w; w += 64; const size_t w_pose2 = w; w = al(w + 64);
    const size_t w_m1 = w; w = al(w + 16 * (size_t)cap1); const size_t w_bad1 = w; w = al(w + (size_t)cap1);
    const size_t w_m2 = w; w = al(w + 16 * (size_t)cap2);
    const size_t w_ma = w; w = al(w + 16 * (size_t)capa); const size_t w_bada = w; w = al(w + (size_t)capa);
    if ((rc = T.h_out.reserve(w))) return rc;
    hipStream_t st = h->ctx->stream;
    std::atomic_thread_fence(std::memory_order_release);   // (every header field the publish reads is written by one of the launches below)

    const float4* kp_xyo = h->dev->kd_in();
    int* hdr = reinterpret_cast<int*>(D + o_hdr);
    // ---- 1: the search against the previous frame (slot 0), its list and look-ups, the first solve
    PmPending pd1, pd2;
    if (np) {
        if ((rc = match_enqueue(h, 0, a->pose0, nullptr, np, a->prev->pos3d, nullptr, nullptr, nullptr, a->prev->desc, a->prev->octave, a->prev_min_desc_dist, a->prev_max_repj_dist, &pd1,
                                reinterpret_cast<float4*>(D + o_posp), a->prev->ids, a->prev_map_row, nullptr, a->map_weight, reinterpret_cast<uint4*>(D + o_auxp)))) return rc;
    }
    // (the map candidates are staged now, while the device works on the first search: slot 1 has its own pinned block)
    TrkSelect s1{};
    s1.nB = np; s1.bk = pd1.d_best_kp; s1.bd = pd1.d_best_dist; s1.aux = reinterpret_cast<const uint4*>(D + o_auxp); s1.map_kind = 0;
    s1.carry = nullptr; s1.hdr = hdr; s1.fresh_n_slot = kTrkN1; s1.final_n_slot = kTrkN1;
    s1.fresh_out = reinterpret_cast<uh_dmatch*>(D + o_m1); s1.fresh_src = reinterpret_cast<int*>(D + o_src1); s1.final_out = nullptr; s1.final_src = nullptr;
    s1.pos_prev = reinterpret_cast<const float4*>(D + o_posp); s1.pos_map = reinterpret_cast<const float4*>(D + o_posm);
    s1.aux_prev = reinterpret_cast<const uint4*>(D + o_auxp); s1.aux_map = reinterpret_cast<const uint4*>(D + o_auxm); s1.prefer_map_row = 0;
    s1.kp_xyo = kp_xyo; s1.inv_sigma_lv = reinterpret_cast<const float*>(dp + q_isl); s1.n_levels = a->n_levels;
    s1.p3d = reinterpret_cast<float*>(D + o_p3d); s1.kp = reinterpret_cast<float*>(D + o_kp); s1.isg = reinterpret_cast<float*>(D + o_isg); s1.wgt = reinterpret_cast<float*>(D + o_wgt);
    s1.n_kpts = nk; s1.scratch_a = reinterpret_cast<TrkElem*>(D + o_sa); s1.scratch_b = reinterpret_cast<TrkElem*>(D + o_sb);
    static const bool trk_clk = getenv("UH_TRK_CLK") != nullptr;
    if (trk_clk && !T.d_clk.p) { if ((rc = T.d_clk.reserve(16 * 8))) return rc; }
    s1.clk = trk_clk ? T.d_clk.as<long long>() : nullptr;
    // the select launches' working lists: in LDS while two lists of capa elements fit beside the per-keypoint table
    const size_t sel_lds = 2 * sizeof(TrkElem) * (size_t)capa;
    const bool sel_in_lds = sel_lds <= 120 * 1024;
    if (sel_in_lds && !T.attr_set) {
        UH_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(track_select_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        T.attr_set = true;
    }
    if (sel_in_lds) UH_LAUNCH(h->ctx, track_select_kernel<true>, dim3(1), dim3(kTrkThreads), sel_lds, s1, capa);
    else UH_LAUNCH(h->ctx, track_select_kernel<false>, dim3(1), dim3(kTrkThreads), 0, s1, capa);
    // (the decision — refined pose + small disc or predicted pose + wide radius — rides on the solve's last thread)
    static_assert(sizeof(PmDyn) == 17 * 4, "PmDyn is what pnp_decide writes");
    PmDyn* dyn = reinterpret_cast<PmDyn*>(D + o_dyn);
    const uh::PnpDecide dec{a->min_inliers, a->map_radius_tracked, a->map_radius_lost, reinterpret_cast<float*>(dyn), reinterpret_cast<float*>(D + o_posem), hdr + kTrkTracked};
    if ((rc = uh::pnp_enqueue_dev(pnp, reinterpret_cast<const float*>(dp + q_pose0), reinterpret_cast<const float*>(dp + q_intr), std::min(cap1, capn), hdr + kTrkN1, s1.p3d, s1.kp, s1.isg, s1.wgt,
                                  reinterpret_cast<float*>(D + o_pose1), reinterpret_cast<unsigned char*>(D + o_bad1), hdr + kTrkRes1, &dec))) return rc;
    // ---- 2: the search of the local map at the decided pose / radius (slot 1), the union, the second solve
    if (nm) {
        if ((rc = match_enqueue(h, 1, nullptr, dyn, nm, a->map->pos3d, a->map->normal, a->map->min_dist, a->map->max_dist, a->map->desc, nullptr, a->map_min_desc_dist, a->map_radius_tracked, &pd2,
                                reinterpret_cast<float4*>(D + o_posm), a->map->ids, nullptr, a->map_weight, nullptr, reinterpret_cast<uint4*>(D + o_auxm)))) return rc;
    }
    TrkSelect s2 = s1;
    s2.nB = nm; s2.bk = pd2.d_best_kp; s2.bd = pd2.d_best_dist; s2.aux = reinterpret_cast<const uint4*>(D + o_auxm); s2.map_kind = 1;
    s2.carry = reinterpret_cast<const uh_dmatch*>(D + o_m1); s2.carry_src = reinterpret_cast<const int*>(D + o_src1); s2.carry_bad = reinterpret_cast<const unsigned char*>(D + o_bad1); s2.carry_cap = cap1;
    s2.fresh_n_slot = kTrkN2; s2.final_n_slot = kTrkNA;
    if (trk_clk) s2.clk = T.d_clk.as<long long>() + 8;
    s2.fresh_out = reinterpret_cast<uh_dmatch*>(D + o_m2); s2.fresh_src = reinterpret_cast<int*>(D + o_src2); s2.final_out = reinterpret_cast<uh_dmatch*>(D + o_ma); s2.final_src = reinterpret_cast<int*>(D + o_srca);
    s2.prefer_map_row = 1;
    if (sel_in_lds) UH_LAUNCH(h->ctx, track_select_kernel<true>, dim3(1), dim3(kTrkThreads), sel_lds, s2, capa);
    else UH_LAUNCH(h->ctx, track_select_kernel<false>, dim3(1), dim3(kTrkThreads), 0, s2, capa);
    if ((rc = uh::pnp_enqueue_dev(pnp, reinterpret_cast<const float*>(D + o_posem), reinterpret_cast<const float*>(dp + q_intr), capn, hdr + kTrkNA, s2.p3d, s2.kp, s2.isg, s2.wgt,
                                  reinterpret_cast<float*>(D + o_pose2), reinterpret_cast<unsigned char*>(D + o_bada), hdr + kTrkRes2, nullptr))) return rc;
    // ---- 3: everything back in one block
    char* ho = T.h_out.host<char>();
    char* dout = T.h_out.dev<char>();
    TrkPublish pb{};
    pb.hdr = hdr; pb.pose1 = reinterpret_cast<const float*>(D + o_pose1); pb.pose2 = reinterpret_cast<const float*>(D + o_pose2);
    pb.m1 = s1.fresh_out; pb.bad1 = reinterpret_cast<const unsigned char*>(D + o_bad1); pb.m2 = s2.fresh_out; pb.ma = s2.final_out; pb.bad2 = reinterpret_cast<const unsigned char*>(D + o_bada);
    pb.cap1 = cap1; pb.cap2 = cap2; pb.capa = capa;
    pb.h_hdr = reinterpret_cast<int*>(dout + w_hdr); pb.h_pose1 = reinterpret_cast<float*>(dout + w_pose1); pb.h_pose2 = reinterpret_cast<float*>(dout + w_pose2);
    pb.h_m1 = reinterpret_cast<uh_dmatch*>(dout + w_m1); pb.h_bad1 = reinterpret_cast<unsigned char*>(dout + w_bad1); pb.h_m2 = reinterpret_cast<uh_dmatch*>(dout + w_m2);
    pb.h_ma = reinterpret_cast<uh_dmatch*>(dout + w_ma); pb.h_bad2 = reinterpret_cast<unsigned char*>(dout + w_bada);
    pb.host_done = T.h_par.dev<unsigned long long>(); pb.word = ++T.seq;
    UH_LAUNCH(h->ctx, track_publish_kernel, dim3(1), dim3(kTrkThreads), 0, pb);
    UH_HIP_CHECK(hipGetLastError());
    if ((rc = uh::wait_host_word(reinterpret_cast<volatile unsigned long long*>(hp), pb.word, st, "uh_track_pose"))) {
        h->slot[0].ovf_zeroed = h->slot[1].ovf_zeroed = false;
        return rc;
    }
    h->upload_pending = false;
    if (trk_clk) {
        long long c[16];
        UH_HIP_CHECK(hipMemcpy(c, T.d_clk.p, sizeof(c), hipMemcpyDeviceToHost));
        for (int k = 0; k < 2; k++) {
            const long long* q = c + 8 * k;
            fprintf(stderr, "track_select %d [us]: stage %.2f  candidates %.2f  filter %.2f  outputs %.2f  union+filter %.2f  look-ups %.2f  total %.2f\n", k + 1, (q[1] - q[0]) * 0.01, (q[2] - q[1]) * 0.01,
                    (q[3] - q[2]) * 0.01, (q[4] - q[3]) * 0.01, (q[5] - q[4]) * 0.01, (q[6] - q[5]) * 0.01, (q[6] - q[0]) * 0.01);
        }
    }
    // (both searches posted their own words and walk-overflow flags on the way: the stream is in order, they are long since visible)
    for (int sl = 0; sl < 2; sl++) {
        if ((sl == 0 && !np) || (sl == 1 && !nm)) continue;
        const int ovf = *reinterpret_cast<const int*>(h->slot[sl].h_out.host<char>() + 8);
        UH_REQUIRE(!ovf, "uh_track_pose: kd-tree walk stack overflow");
    }
    const int* hh = reinterpret_cast<const int*>(ho + w_hdr);
    const int n1 = hh[kTrkN1], n2 = hh[kTrkN2], na = hh[kTrkNA];
    UH_REQUIRE(n1 <= r->cap_prev && n2 <= r->cap_map && na <= r->cap_all, "uh_track_pose: %d / %d / %d matches do not fit the output buffers (%d / %d / %d)", n1, n2, na, r->cap_prev, r->cap_map, r->cap_all);
    r->n_prev = n1; r->n_map = n2; r->n_all = na; r->tracked = hh[kTrkTracked];
    r->inliers1 = hh[kTrkRes1]; r->inliers2 = hh[kTrkRes2];
    for (int i = 0; i < 4; i++) { r->iters1[i] = hh[kTrkRes1 + 1 + i]; r->iters2[i] = hh[kTrkRes2 + 1 + i]; }
    std::memcpy(r->pose1, ho + w_pose1, 64); std::memcpy(r->pose2, ho + w_pose2, 64);
    if (n1) { std::memcpy(r->matches_prev, ho + w_m1, 16 * (size_t)n1); std::memcpy(r->bad_prev, ho + w_bad1, (size_t)n1); }
    if (n2) std::memcpy(r->matches_map, ho + w_m2, 16 * (size_t)n2);
    if (na) { std::memcpy(r->matches_all, ho + w_ma, 16 * (size_t)na); std::memcpy(r->bad_all, ho + w_bada, (size_t)na); }
    return UH_OK;
}

}  // extern "C"
