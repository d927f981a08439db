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
// SEVEN launches on the context stream and waits once; what the host did between the calls runs on the device:
//   projmatch_kernel<prev>  ->  track_select_kernel (matches in item order, filter_ambiguous_query, look-ups for the solve)
//   -> pnp_solve_kernel (match count from device memory)  ->  track_decide_kernel (pose / radius of the map search)
//   -> projmatch_kernel<map> (pose and radius read from device memory)  ->  track_select_kernel (its own filter, the union with the
//   first solve's inliers, filter, look-ups)  ->  pnp_solve_kernel  ->  track_publish_kernel (everything the sequence of calls returns).
// Same results as the four calls, bit for bit (tests/test_track.py, tests/test_cpp_host.py): the same kernels do the searching and
// the solving; the list logic is integer work (stable minimum per keypoint = uh_filter_ambiguous, matcher.hip keep_best_per_key).
// This file is included at the end of projmatch.hip (it uses that unit's internals).
#pragma once

namespace uh {
int pnp_enqueue_dev(uh_pnp* p, const float* d_pose, const float* d_intr4, int n_cap, const int* d_n, const float* d_p3d, const float* d_kp, const float* d_inv_sigma,
                    const float* d_weight, float* d_pose_out, unsigned char* d_bad_out, int* d_result5);
uh_ctx* pnp_ctx(uh_pnp* p);
}

namespace {

constexpr int kTrkThreads = 1024;
struct TrkElem { int query; unsigned id; float dist; int src; };   // src: bit 30 = candidate of the map search, low bits = its index in its candidate set

// header of the device block (ints)
enum : int { kTrkN1 = 0, kTrkN2 = 1, kTrkNA = 2, kTrkTracked = 3, kTrkRes1 = 4 /* 5 ints */, kTrkRes2 = 9 /* 5 ints */, kTrkHdrInts = 16 };

struct TrkSelect {
    // the search whose results become a list (candidate order)
    int nB; const int* bk; const float* bd; const unsigned* ids; int map_kind;
    // second form only: the list carried over from the first solve (its inliers enter the union when the frame counts as tracked)
    const uh_dmatch* carry; const int* carry_src; const unsigned char* carry_bad; int carry_cap;
    int* hdr;                       // counts in / out (see the enum)
    int fresh_n_slot, final_n_slot; // hdr slots of the fresh list's and the final list's length (equal in the first form)
    uh_dmatch* fresh_out; int* fresh_src;   // the search's own matches (after filter_ambiguous_query)
    uh_dmatch* final_out;                   // second form: the union after its filter (NULL in the first form)
    int* final_src;                         // scratch: source of each element of the final list
    // look-ups for the solve over the final list
    const float* rec_prev; const float* rec_map;   // 64-byte candidate records (position = floats 0..2)
    const int* prev_map_row; const float* map_weight; int prefer_map_row;
    const float4* kp_xyo; const float* inv_sigma_lv; int n_levels;
    float* p3d; float* kp; float* isg; float* wgt;
    int n_kpts;
    TrkElem* scratch_a; TrkElem* scratch_b;
};

// exclusive prefix over the workgroup's flags of one chunk (one flag per thread); returns the thread's rank and the chunk's total
__device__ __forceinline__ int trk_block_rank(int flag, int* s_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = uh_kd::wave_incl_scan(flag);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kTrkThreads / 64; w++) { const int v = s_wave[w]; base += w < wave ? v : 0; tot += v; }
    __syncthreads();
    total = tot;
    return base + inc - flag;
}

// uh_filter_ambiguous(list, n, by query) on the device: the first strict minimum of the distance per keypoint stays, in list order
// (matcher.hip keep_best_per_key).  in / out may not alias.  Returns the new length (every thread).
__device__ int trk_filter(const TrkElem* __restrict__ in, int n, TrkElem* __restrict__ out, unsigned long long* s_best, int n_kpts, int* s_wave) {
    for (int k = threadIdx.x; k < n_kpts; k += kTrkThreads) s_best[k] = ~0ull;
    __syncthreads();
    for (int pos = threadIdx.x; pos < n; pos += kTrkThreads) {
        const TrkElem e = in[pos];
        atomicMin(&s_best[e.query], ((unsigned long long)__float_as_uint(e.dist) << 32) | (unsigned)pos);   // distances are >= 0: their bit patterns order like the values
    }
    __syncthreads();
    int kept = 0;
    for (int p0 = 0; p0 < n; p0 += kTrkThreads) {
        const int pos = p0 + threadIdx.x;
        TrkElem e{};
        int keep = 0;
        if (pos < n) { e = in[pos]; keep = (unsigned)(s_best[e.query] & 0xffffffffull) == (unsigned)pos ? 1 : 0; }
        int tot;
        const int r = trk_block_rank(keep, s_wave, tot);
        if (keep) out[kept + r] = e;
        kept += tot;
    }
    __syncthreads();
    return kept;
}

__global__ __launch_bounds__(kTrkThreads) void track_select_kernel(TrkSelect a) {
    __shared__ unsigned long long s_best[4096];
    __shared__ int s_wave[kTrkThreads / 64];
    const int tid = threadIdx.x;
    // ---- the search's hits in candidate order
    int nF = 0;
    for (int p0 = 0; p0 < a.nB; p0 += kTrkThreads) {
        const int i = p0 + tid;
        int kpi = -1;
        if (i < a.nB) kpi = a.bk[i];
        const int flag = kpi >= 0 ? 1 : 0;
        int tot;
        const int r = trk_block_rank(flag, s_wave, tot);
        if (flag) a.scratch_a[nF + r] = TrkElem{kpi, a.ids[i], a.bd[i], (a.map_kind << 30) | i};
        nF += tot;
    }
    __syncthreads();
    // ---- filter_ambiguous_query of the search's own list (uh_projmatch_match / _match_prev end with it)
    const int n_fresh = trk_filter(a.scratch_a, nF, a.scratch_b, s_best, a.n_kpts, s_wave);
    for (int p = tid; p < n_fresh; p += kTrkThreads) {
        const TrkElem e = a.scratch_b[p];
        a.fresh_out[p] = uh_dmatch{e.query, (int)e.id, -1, e.dist};
        a.fresh_src[p] = e.src;
    }
    const TrkElem* fin = a.scratch_b;
    int n_fin = n_fresh;
    if (a.carry) {
        // ---- the union: inliers of the first solve (when the frame counts as tracked), then the new matches; filter again
        const int tracked = a.hdr[kTrkTracked];
        const int nc = tracked ? min(a.hdr[kTrkN1], a.carry_cap) : 0;
        int nU = 0;
        for (int p0 = 0; p0 < nc; p0 += kTrkThreads) {
            const int p = p0 + tid;
            const int flag = p < nc && !a.carry_bad[p] ? 1 : 0;
            int tot;
            const int r = trk_block_rank(flag, s_wave, tot);
            if (flag) { const uh_dmatch m = a.carry[p]; a.scratch_a[nU + r] = TrkElem{m.queryIdx, (unsigned)m.trainIdx, m.distance, a.carry_src[p]}; }
            nU += tot;
        }
        __syncthreads();
        for (int p = tid; p < n_fresh; p += kTrkThreads) a.scratch_a[nU + p] = a.scratch_b[p];
        __syncthreads();
        nU += n_fresh;
        // (scratch_b is free again: its content lives in scratch_a now)
        n_fin = trk_filter(a.scratch_a, nU, a.scratch_b, s_best, a.n_kpts, s_wave);
        fin = a.scratch_b;
        for (int p = tid; p < n_fin; p += kTrkThreads) { const TrkElem e = fin[p]; a.final_out[p] = uh_dmatch{e.query, (int)e.id, -1, e.dist}; }
    }
    // ---- the solver's per-match look-ups (pnpsolver.cpp:199-232): the point's coordinates and weight, the keypoint, 1 / scaleFactor of its octave
    for (int p = tid; p < n_fin; p += kTrkThreads) {
        const TrkElem e = fin[p];
        const int is_map = (e.src >> 30) & 1, idx = e.src & 0x3fffffff;
        int row = is_map ? idx : -1;
        if (!is_map && a.prefer_map_row && a.prev_map_row) row = a.prev_map_row[idx];
        const float* rec = row >= 0 ? a.rec_map + 16 * (size_t)row : a.rec_prev + 16 * (size_t)idx;
        a.p3d[3 * p] = rec[0]; a.p3d[3 * p + 1] = rec[1]; a.p3d[3 * p + 2] = rec[2];
        a.wgt[p] = row >= 0 && a.prefer_map_row && a.map_weight ? a.map_weight[row] : 1.f;
        const float4 k = a.kp_xyo[e.query];
        a.kp[2 * p] = k.x; a.kp[2 * p + 1] = k.y;
        const int oct = (int)(__float_as_uint(k.z) & 15u);
        a.isg[p] = a.inv_sigma_lv[oct < a.n_levels ? oct : 0];
        if (a.final_src) a.final_src[p] = e.src;
    }
    if (tid == 0) { a.hdr[a.fresh_n_slot] = n_fresh; a.hdr[a.final_n_slot] = n_fin; }
}

// system.cpp:6762-6881: with at least min_inliers inliers the refined pose is kept and the local map is searched in a small disc; otherwise
// the first matches are dropped, the predicted pose stays and the radius is the wide one.  Leaves the pose for the search (as the kernel
// wants it: rows + camera centre, se3transform.h:89-113 — the host's float expressions), for the second solve, and the flag.
__global__ void track_decide_kernel(const float* __restrict__ pose0, const float* __restrict__ pose1, int* hdr, int min_inliers, float r_tracked, float r_lost,
                                    PmDyn* dyn, float* pose_map) {
    if (threadIdx.x != 0) return;
    const int tracked = hdr[kTrkRes1] >= min_inliers ? 1 : 0;
    const float* T = tracked ? pose1 : pose0;
    PmDyn d;
    for (int i = 0; i < 12; i++) d.ps.T[i] = T[i];
    const float m0 = T[0], m1 = T[4], m2 = T[8], m4 = T[1], m5 = T[5], m6 = T[9], m8 = T[2], m9 = T[6], m10 = T[10];
    const float m3 = -(T[3] * m0 + T[7] * m1 + T[11] * m2), m7 = -(T[3] * m4 + T[7] * m5 + T[11] * m6), m11 = -(T[3] * m8 + T[7] * m9 + T[11] * m10);
    d.ps.cc[0] = m0 * 0.f + m1 * 0.f + m2 * 0.f + m3;
    d.ps.cc[1] = m4 * 0.f + m5 * 0.f + m6 * 0.f + m7;
    d.ps.cc[2] = m8 * 0.f + m9 * 0.f + m10 * 0.f + m11;
    d.radius = tracked ? r_tracked : r_lost;
    d.skip = 0;
    *dyn = d;
    for (int i = 0; i < 16; i++) pose_map[i] = T[i];
    hdr[kTrkTracked] = tracked;
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
    if ((rc = T.d.reserve(o))) return rc;
    char* D = T.d.as<char>();
    // ---- pinned parameter block (read by the launches in place)
    size_t q = 64;
    const size_t q_pose0 = q; q += 64; const size_t q_intr = q; q += 64; const size_t q_isl = q; q += 64;
    const size_t q_idp = q; q = al(q + 4 * (size_t)cap1); const size_t q_idm = q; q = al(q + 4 * (size_t)cap2);
    const size_t q_row = q; q = al(q + 4 * (size_t)cap1); const size_t q_w = q; q = al(q + 4 * (size_t)cap2);
    if ((rc = T.h_par.reserve(q))) return rc;   // (the previous call's launches are complete: its results were awaited)
    char* hp = T.h_par.host<char>();
    char* dp = T.h_par.dev<char>();
    std::memcpy(hp + q_pose0, a->pose0, 64);
    std::memcpy(hp + q_intr, a->intr4, 16);
    std::memcpy(hp + q_isl, a->inv_sigma_levels, 4 * (size_t)a->n_levels);
    if (np) std::memcpy(hp + q_idp, a->prev->ids, 4 * (size_t)np);
    if (nm) std::memcpy(hp + q_idm, a->map->ids, 4 * (size_t)nm);
    if (a->prev_map_row && np) std::memcpy(hp + q_row, a->prev_map_row, 4 * (size_t)np);
    if (a->map_weight && nm) std::memcpy(hp + q_w, a->map_weight, 4 * (size_t)nm);
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
        if ((rc = match_enqueue(h, 0, a->pose0, nullptr, np, a->prev->pos3d, nullptr, nullptr, nullptr, a->prev->desc, a->prev->octave, a->prev_min_desc_dist, a->prev_max_repj_dist, &pd1))) return rc;
    }
    // (the map candidates are staged now, while the device works on the first search: slot 1 has its own pinned block)
    TrkSelect s1{};
    s1.nB = np; s1.bk = pd1.d_best_kp; s1.bd = pd1.d_best_dist; s1.ids = reinterpret_cast<const unsigned*>(dp + q_idp); s1.map_kind = 0;
    s1.carry = nullptr; s1.hdr = hdr; s1.fresh_n_slot = kTrkN1; s1.final_n_slot = kTrkN1;
    s1.fresh_out = reinterpret_cast<uh_dmatch*>(D + o_m1); s1.fresh_src = reinterpret_cast<int*>(D + o_src1); s1.final_out = nullptr; s1.final_src = nullptr;
    s1.rec_prev = pd1.d_rec; s1.rec_map = nullptr; s1.prev_map_row = nullptr; s1.map_weight = nullptr; s1.prefer_map_row = 0;
    s1.kp_xyo = kp_xyo; s1.inv_sigma_lv = reinterpret_cast<const float*>(dp + q_isl); s1.n_levels = a->n_levels;
    s1.p3d = reinterpret_cast<float*>(D + o_p3d); s1.kp = reinterpret_cast<float*>(D + o_kp); s1.isg = reinterpret_cast<float*>(D + o_isg); s1.wgt = reinterpret_cast<float*>(D + o_wgt);
    s1.n_kpts = nk; s1.scratch_a = reinterpret_cast<TrkElem*>(D + o_sa); s1.scratch_b = reinterpret_cast<TrkElem*>(D + o_sb);
    UH_LAUNCH(h->ctx, track_select_kernel, dim3(1), dim3(kTrkThreads), 0, s1);
    if ((rc = uh::pnp_enqueue_dev(pnp, reinterpret_cast<const float*>(dp + q_pose0), reinterpret_cast<const float*>(dp + q_intr), std::min(cap1, capn), hdr + kTrkN1, s1.p3d, s1.kp, s1.isg, s1.wgt,
                                  reinterpret_cast<float*>(D + o_pose1), reinterpret_cast<unsigned char*>(D + o_bad1), hdr + kTrkRes1))) return rc;
    // ---- 2: the decision, the search of the local map at the decided pose / radius (slot 1), the union, the second solve
    PmDyn* dyn = reinterpret_cast<PmDyn*>(D + o_dyn);
    UH_LAUNCH(h->ctx, track_decide_kernel, dim3(1), dim3(64), 0, reinterpret_cast<const float*>(dp + q_pose0), reinterpret_cast<const float*>(D + o_pose1), hdr, a->min_inliers, a->map_radius_tracked,
              a->map_radius_lost, dyn, reinterpret_cast<float*>(D + o_posem));
    if (nm) {
        if ((rc = match_enqueue(h, 1, nullptr, dyn, nm, a->map->pos3d, a->map->normal, a->map->min_dist, a->map->max_dist, a->map->desc, nullptr, a->map_min_desc_dist, a->map_radius_tracked, &pd2))) return rc;
    }
    TrkSelect s2 = s1;
    s2.nB = nm; s2.bk = pd2.d_best_kp; s2.bd = pd2.d_best_dist; s2.ids = reinterpret_cast<const unsigned*>(dp + q_idm); s2.map_kind = 1;
    s2.carry = reinterpret_cast<const uh_dmatch*>(D + o_m1); s2.carry_src = reinterpret_cast<const int*>(D + o_src1); s2.carry_bad = reinterpret_cast<const unsigned char*>(D + o_bad1); s2.carry_cap = cap1;
    s2.fresh_n_slot = kTrkN2; s2.final_n_slot = kTrkNA;
    s2.fresh_out = reinterpret_cast<uh_dmatch*>(D + o_m2); s2.fresh_src = reinterpret_cast<int*>(D + o_src2); s2.final_out = reinterpret_cast<uh_dmatch*>(D + o_ma); s2.final_src = reinterpret_cast<int*>(D + o_srca);
    s2.rec_prev = pd1.d_rec; s2.rec_map = pd2.d_rec;
    s2.prev_map_row = a->prev_map_row ? reinterpret_cast<const int*>(dp + q_row) : nullptr; s2.map_weight = a->map_weight ? reinterpret_cast<const float*>(dp + q_w) : nullptr; s2.prefer_map_row = 1;
    UH_LAUNCH(h->ctx, track_select_kernel, dim3(1), dim3(kTrkThreads), 0, s2);
    if ((rc = uh::pnp_enqueue_dev(pnp, reinterpret_cast<const float*>(D + o_posem), reinterpret_cast<const float*>(dp + q_intr), capn, hdr + kTrkNA, s2.p3d, s2.kp, s2.isg, s2.wgt,
                                  reinterpret_cast<float*>(D + o_pose2), reinterpret_cast<unsigned char*>(D + o_bada), hdr + kTrkRes2))) return rc;
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
