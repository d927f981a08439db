// FrameMatcher_Flann post-filter (host code): what UcoSLAM does with the kNN rows the index returns.
//
// Reference: src/utils/framematcher.cpp:228-319 (matchEpipolar body after the search), :67-108 (computeThreeMaxima),
//            src/basictypes/misc.cpp:105-107 (remove_unused_matches), :153-185 (filter_ambiguous_train), :117-150
//            (filter_ambiguous_query), src/basictypes/misc.h:72-81 (epipolarLineSqDist).
// This is sequential policy over <= nq*nn candidates (microseconds on a host core); it stays on the host exactly like the
// reference, consuming the (bit-exact, heap-ordered) rows of uh_knn_search.  The result depends on the column ORDER of the
// unsorted rows, which is why uh_knn_search reproduces the reference heap layout.
//
// FrameMatcher_BoW (src/utils/framematcher.cpp:407-535): the second matcher type.  Features of the two frames that fell into
// the same vocabulary node (fbow::fBow2 of Vocabulary::transform at level 3, keyframedatabase.cpp:319) are compared all against
// all; the Hamming distances and the per-query-feature best / "last not better" bookkeeping run on the GPU (bow_match_kernel,
// one lane per query feature, candidates in the node's list order — the rule is order dependent), the rest of the chain
// (filter_ambiguous_train, orientation histogram) is the host code shared with the FLANN matcher.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

// squared distance of kp2 from the epipolar line that F assigns to kp1 (misc.h:72-81; F row-major 3x3, float throughout)
inline float epipolar_sq_dist(const float* kp1, const float* kp2, const float* F) {
    const float a = kp1[0] * F[0] + kp1[1] * F[3] + F[6];
    const float b = kp1[0] * F[1] + kp1[1] * F[4] + F[7];
    const float den = a * a + b * b;
    if (den == 0) return std::numeric_limits<float>::max();
    const float c = kp1[0] * F[2] + kp1[1] * F[5] + F[8];
    const float num = a * kp2[0] + b * kp2[1] + c;
    return num * num / den;
}

// ---- one-to-one filter --------------------------------------------------------------------------------------------------------
// Contract (basictypes/misc.cpp:117-185, both directions): among the matches that share a key (their train index, or their query
// index) only the one with the smallest distance survives; of several with that same smallest distance the EARLIEST in the list
// survives; survivors keep their relative order.  Formulated as two flat passes: pass 1 records, per key, the position of the
// group's stable minimum; pass 2 compacts the list to the positions that are their group's minimum.
inline int key_of(const uh_dmatch& m, bool by_train) { return by_train ? m.trainIdx : m.queryIdx; }

void keep_best_per_key(std::vector<uh_dmatch>& list, bool by_train) {
    const int n = (int)list.size();
    if (n == 0) return;
    int key_span = 0;
    for (const uh_dmatch& m : list) key_span = std::max(key_span, key_of(m, by_train) + 1);
    std::vector<int> group_min(key_span, -1);   // position of the group's stable minimum so far
    for (int pos = 0; pos < n; pos++) {
        int& holder = group_min[key_of(list[pos], by_train)];
        if (holder < 0 || list[pos].distance < list[holder].distance) holder = pos;   // strict '<': an equal later match never displaces
    }
    int kept = 0;
    for (int pos = 0; pos < n; pos++)
        if (group_min[key_of(list[pos], by_train)] == pos) list[kept++] = list[pos];
    list.resize(kept);
}

// ---- orientation consensus ----------------------------------------------------------------------------------------------------
// Contract (framematcher.cpp:290-316 with computeThreeMaxima :67-108): every match votes with the rotation train.angle - query.angle
// (wrapped into [0, 360)) for slot round(rotation / 30) of a 30-slot table (slot 30 wraps to 0, so only slots 0..12 ever fill);
// the three fullest slots win, an earlier slot beating a later one with the same count; the runner-up slots are dropped when they
// hold less than a tenth of the winner (if the second falls, the third falls with it); matches outside the winning slots go.
constexpr int kRotSlots = 30;

void keep_dominant_rotations(std::vector<uh_dmatch>& list, const float* t_angle, const float* q_angle) {
    const int n = (int)list.size();
    std::vector<unsigned char> slot_of(n);
    int votes[kRotSlots] = {0};
    const float per_slot = 1.0f / float(kRotSlots);
    for (int pos = 0; pos < n; pos++) {
        float rot = t_angle[list[pos].trainIdx] - q_angle[list[pos].queryIdx];
        if (rot < 0.0) rot += 360.0f;
        size_t slot = (size_t)std::round(rot * per_slot);
        if (slot == (size_t)kRotSlots) slot = 0;
        slot_of[pos] = (unsigned char)slot;
        votes[slot]++;
    }
    // podium of three: a slot enters at the first rank whose count it strictly exceeds and pushes the lower ranks down
    int rank_slot[3] = {-1, -1, -1}, rank_votes[3] = {0, 0, 0};
    for (int slot = 0; slot < kRotSlots; slot++) {
        for (int r = 0; r < 3; r++) {
            if (votes[slot] > rank_votes[r]) {
                for (int d = 2; d > r; d--) { rank_votes[d] = rank_votes[d - 1]; rank_slot[d] = rank_slot[d - 1]; }
                rank_votes[r] = votes[slot]; rank_slot[r] = slot;
                break;
            }
        }
    }
    const float tenth = 0.1f * (float)rank_votes[0];
    if ((float)rank_votes[1] < tenth) rank_slot[1] = rank_slot[2] = -1;
    else if ((float)rank_votes[2] < tenth) rank_slot[2] = -1;
    bool winner[kRotSlots] = {false};
    for (int r = 0; r < 3; r++) if (rank_slot[r] >= 0) winner[rank_slot[r]] = true;
    int kept = 0;
    for (int pos = 0; pos < n; pos++)
        if (winner[slot_of[pos]]) list[kept++] = list[pos];
    list.resize(kept);
}

// the tail both matcher types share (framematcher.cpp:288-316 and :504-531): one match per train keypoint, then the rotation consensus
void finish_matches(std::vector<uh_dmatch>& matches, const float* t_angle, const float* q_angle, int check_orientation) {
    keep_best_per_key(matches, true);
    if (check_orientation) keep_dominant_rotations(matches, t_angle, q_angle);
}

// ---- choosing a query row's match from its kNN columns --------------------------------------------------------------------------
// Contract (framematcher.cpp:248-286).  The columns are visited in the order the index returned them (unsorted heap order: the result
// depends on it).  A column is considered only while its distance is below the current RIVAL distance; a considered
// column that passes the gates either becomes the new LEADER (distance strictly below the leader's; the old leader is NOT demoted to
// rival) or overwrites the rival.  The leader is accepted unless the rival has the query's own octave and the leader is not better
// than ratio x rival.
struct RowChoice {
    float lead_dist, rival_dist = std::numeric_limits<float>::max();
    int lead_train = -1, rival_octave = -1;
    explicit RowChoice(float ceiling) : lead_dist(ceiling) {}
    bool still_open(float dist) const { return dist < rival_dist; }
    void offer(float dist, int train, int train_octave) {
        if (dist < lead_dist) { lead_dist = dist; lead_train = train; }
        else { rival_dist = dist; rival_octave = train_octave; }
    }
    bool accepted(int query_octave, float ratio) const {
        return lead_train >= 0 && !(rival_octave == query_octave && lead_dist > rival_dist * ratio);
    }
};

// ------------------------------------------------------------------------------------------------ BoW matcher kernel
struct BowMatchDev {
    int n_entries;                   // query features listed under nodes both frames have, in (node, list) order
    const unsigned int* q_feat;      // [n_entries] query keypoint index
    const int* t_begin;              // [n_entries] first candidate in t_feat
    const int* t_count;              // [n_entries]
    const unsigned int* t_feat;      // candidates: train keypoint indices, node lists back to back
    const uint64_t* q_desc; const uint64_t* t_desc;        // n x 4 words
    const int* q_octave; const int* t_octave;
    const float* q_pt; const float* t_pt;                  // (x, y) pairs, only read with F12
    const unsigned char* q_used; const unsigned char* t_used;   // isUsed(frame, idx, mode) per keypoint
    const float* scale2;             // scaleFactors[i]^2 of the query frame
    float F[9]; int have_F;
    float min_desc_dist; int max_octave_diff;
    int* best_train; float* best_dist; float* best_dist2; int* octave_best2;   // [n_entries] outputs
};

__device__ __forceinline__ float epipolar_sq_dist_dev(const float* kp1, const float* kp2, const float* F) {   // misc.h:72-81
    const float a = kp1[0] * F[0] + kp1[1] * F[3] + F[6];
    const float b = kp1[0] * F[1] + kp1[1] * F[4] + F[7];
    const float den = a * a + b * b;
    if (den == 0) return 3.402823466e+38f;
    const float c = kp1[0] * F[2] + kp1[1] * F[5] + F[8];
    const float num = a * kp2[0] + b * kp2[1] + c;
    return num * num / den;
}

// framematcher.cpp:441-476 for one query feature: candidates in list order; a candidate below the best REPLACES it, any other
// one (that passed the gates) overwrites bestDist2 / octaveBest2 — "last not better", not "second best"
__global__ __launch_bounds__(256) void bow_match_kernel(BowMatchDev a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n_entries) return;
    const unsigned int qi = a.q_feat[e];
    int bestTrain = -1, octaveBest2 = -1;
    float bestDist = a.min_desc_dist, bestDist2 = 3.402823466e+38f;
    if (a.q_used[qi]) {
        const uint64_t* qd = a.q_desc + 4 * (size_t)qi;
        const uint64_t q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
        const int qo = a.q_octave[qi];
        const int b = a.t_begin[e], n = a.t_count[e];
        for (int j = 0; j < n; j++) {
            const unsigned int ti = a.t_feat[b + j];
            if (!a.t_used[ti]) continue;
            const int to = a.t_octave[ti];
            if (abs(to - qo) > a.max_octave_diff) continue;
            if (a.have_F) {
                if (epipolar_sq_dist_dev(a.t_pt + 2 * (size_t)ti, a.q_pt + 2 * (size_t)qi, a.F) >= 3.84 * (double)a.scale2[qo]) continue;
            }
            const uint64_t* td = a.t_desc + 4 * (size_t)ti;
            const float dist = (float)(__popcll(q0 ^ td[0]) + __popcll(q1 ^ td[1]) + __popcll(q2 ^ td[2]) + __popcll(q3 ^ td[3]));
            if (dist < bestDist) { bestDist = dist; bestTrain = (int)ti; }
            else { bestDist2 = dist; octaveBest2 = to; }
        }
    }
    a.best_train[e] = bestTrain; a.best_dist[e] = bestDist; a.best_dist2[e] = bestDist2; a.octave_best2[e] = octaveBest2;
}

}  // namespace

extern "C" {

int uh_filter_ambiguous(uh_dmatch* matches, int n, int by_train) {
    UH_REQUIRE(n >= 0 && (n == 0 || matches), "uh_filter_ambiguous: bad arguments");
    std::vector<uh_dmatch> v(matches, matches + n);
    keep_best_per_key(v, by_train != 0);
    std::copy(v.begin(), v.end(), matches);
    return (int)v.size();
}

int uh_match_filter(const uh_match_filter_args* a, uh_dmatch* out, int cap) {
    UH_REQUIRE(a && out, "uh_match_filter: NULL argument");
    UH_REQUIRE(a->nq >= 0 && a->nn >= 1 && a->indices && a->distances, "uh_match_filter: bad kNN rows");
    UH_REQUIRE(a->q_octave && a->q_angle && a->t_octave && a->t_angle, "uh_match_filter: keypoint arrays missing");
    if (a->F12) UH_REQUIRE(a->q_pt && a->t_pt && a->scale_factors, "uh_match_filter: epipolar gate needs points and scale factors");
    // sizes of the keypoint arrays, when the caller states them (0 = unchecked, as in round 1): every index is validated before use
    const int n_train = a->n_train_kpts, n_query = a->n_query_kpts, n_levels = a->n_levels;
    std::vector<uh_dmatch> matches;
    for (int row = 0; row < a->nq; row++) {
        const int query = a->map_idx_query ? (int)a->map_idx_query[row] : row;
        UH_REQUIRE(n_query <= 0 || (query >= 0 && query < n_query), "uh_match_filter: query keypoint %d outside [0,%d)", query, n_query);
        const int query_octave = a->q_octave[query];
        if (a->F12) UH_REQUIRE(n_levels <= 0 || (query_octave >= 0 && query_octave < n_levels), "uh_match_filter: octave %d outside [0,%d)", query_octave, n_levels);
        RowChoice pick(a->min_desc_dist);
        const int32_t* cols = a->indices + (size_t)row * a->nn;
        const int32_t* dists = a->distances + (size_t)row * a->nn;
        for (int c = 0; c < a->nn; c++) {
            const float dist = (float)dists[c];                 // the reference converts the distance matrix to float first
            if (dist > a->min_desc_dist || !pick.still_open(dist)) continue;
            if (cols[c] < 0) continue;                          // unfilled slot of the index (-1, distance 0): nothing to map
            const int train = a->map_idx_train ? (int)a->map_idx_train[cols[c]] : cols[c];
            UH_REQUIRE(n_train <= 0 || (train >= 0 && train < n_train), "uh_match_filter: train keypoint %d outside [0,%d)", train, n_train);
            const int train_octave = a->t_octave[train];
            if (std::abs(train_octave - query_octave) > a->max_octave_diff) continue;
            if (a->F12) {                                       // chi-square gate at the query keypoint's scale: 3.84 * sigma^2
                const float sigma = a->scale_factors[query_octave];
                if (epipolar_sq_dist(a->t_pt + 2 * (size_t)train, a->q_pt + 2 * (size_t)query, a->F12) >= 3.84 * (double)(sigma * sigma)) continue;
            }
            pick.offer(dist, train, train_octave);
        }
        if (pick.accepted(query_octave, a->nn_match_ratio)) {
            uh_dmatch m;
            m.queryIdx = query; m.trainIdx = pick.lead_train; m.imgIdx = -1; m.distance = pick.lead_dist;
            matches.push_back(m);
        }
    }
    finish_matches(matches, a->t_angle, a->q_angle, a->check_orientation);
    if ((int)matches.size() > cap) { uh::set_error("uh_match_filter: %zu matches but capacity %d", matches.size(), cap); return UH_ECAPACITY; }
    std::copy(matches.begin(), matches.end(), out);
    return (int)matches.size();
}

}  // extern "C"

struct uh_bowmatch {
    uh_ctx* ctx = nullptr;
    uh::DevBuf d_buf;
    uh::PinBuf h_in, h_out;
};

extern "C" {

int uh_bowmatch_create(uh_ctx* ctx, uh_bowmatch** out) {
    UH_REQUIRE(ctx && out, "uh_bowmatch_create: NULL argument");
    uh_bowmatch* h = new uh_bowmatch();
    h->ctx = ctx;
    *out = h;
    return UH_OK;
}

void uh_bowmatch_destroy(uh_bowmatch* h) { delete h; }

static int check_bow_frame(const uh_bow_frame* f, const char* which) {
    UH_REQUIRE(f->n_kpts >= 0 && f->n_nodes >= 0, "uh_bowmatch_match: %s frame: negative size", which);
    if (f->n_nodes > 0) UH_REQUIRE(f->node_ids && f->node_ptr && f->feat_idx, "uh_bowmatch_match: %s frame: node lists missing", which);
    if (f->n_kpts > 0) UH_REQUIRE(f->desc && f->octave && f->angle, "uh_bowmatch_match: %s frame: keypoint arrays missing", which);
    for (int i = 0; i < f->n_nodes; i++) {
        UH_REQUIRE(f->node_ptr[i + 1] > f->node_ptr[i], "uh_bowmatch_match: %s frame: node %d has an empty list (the reference's loop never advances past one)", which, i);
        if (i) UH_REQUIRE(f->node_ids[i] > f->node_ids[i - 1], "uh_bowmatch_match: %s frame: node ids must ascend (std::map order)", which);
    }
    const int total = f->n_nodes ? f->node_ptr[f->n_nodes] : 0;
    for (int i = 0; i < total; i++) UH_REQUIRE(f->feat_idx[i] < (uint32_t)f->n_kpts, "uh_bowmatch_match: %s frame: feature index %u out of range", which, f->feat_idx[i]);   // unsigned compare: 2^31.. must not pass
    return UH_OK;
}

int uh_bowmatch_match(uh_bowmatch* h, const uh_bow_match_args* a, uh_dmatch* out, int cap) {
    UH_REQUIRE(h && a && out, "uh_bowmatch_match: NULL argument");
    const uh_bow_frame& Q = a->query;
    const uh_bow_frame& T = a->train;
    int rc;
    if ((rc = check_bow_frame(&Q, "query"))) return rc;
    if ((rc = check_bow_frame(&T, "train"))) return rc;
    if (a->F12) UH_REQUIRE(Q.pt && T.pt && a->scale_factors && a->n_levels >= 1, "uh_bowmatch_match: epipolar gate needs points and scale factors");
    // merge-join of the two node maps (framematcher.cpp:422-489): query features of common nodes, with their candidate ranges
    std::vector<unsigned int> q_feat;
    std::vector<int> t_begin, t_count;
    {
        int qi = 0, ti = 0;
        while (qi < Q.n_nodes && ti < T.n_nodes) {
            if (Q.node_ids[qi] == T.node_ids[ti]) {
                for (int k = Q.node_ptr[qi]; k < Q.node_ptr[qi + 1]; k++) {
                    q_feat.push_back(Q.feat_idx[k]);
                    t_begin.push_back(T.node_ptr[ti]);
                    t_count.push_back(T.node_ptr[ti + 1] - T.node_ptr[ti]);
                }
                ++qi; ++ti;
            } else if (Q.node_ids[qi] < T.node_ids[ti]) ++qi;
            else ++ti;
        }
    }
    const int ne = (int)q_feat.size();
    std::vector<uh_dmatch> matches;
    if (ne > 0) {
        if (a->F12) for (int i = 0; i < Q.n_kpts; i++) UH_REQUIRE(Q.octave[i] >= 0 && Q.octave[i] < a->n_levels, "uh_bowmatch_match: query octave %d outside the scale factors", Q.octave[i]);
        UH_HIP_CHECK(hipSetDevice(h->ctx->device));
        hipStream_t st = h->ctx->stream;
        const int nt_feat = T.node_ptr[T.n_nodes];
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off = al(off + bytes); return o; };
        const size_t o_qf = take(4 * (size_t)ne), o_tb = take(4 * (size_t)ne), o_tc = take(4 * (size_t)ne), o_tf = take(4 * (size_t)nt_feat);
        const size_t o_qd = take(32 * (size_t)Q.n_kpts), o_td = take(32 * (size_t)T.n_kpts), o_qo = take(4 * (size_t)Q.n_kpts), o_to = take(4 * (size_t)T.n_kpts);
        const size_t o_qp = take(8 * (size_t)Q.n_kpts), o_tp = take(8 * (size_t)T.n_kpts), o_qu = take(Q.n_kpts), o_tu = take(T.n_kpts);
        const size_t o_s2 = take(4 * (size_t)std::max(a->n_levels, 1));
        const size_t in_bytes = off;
        const size_t o_bt = take(4 * (size_t)ne), o_bd = take(4 * (size_t)ne), o_b2 = take(4 * (size_t)ne), o_o2 = take(4 * (size_t)ne);
        const size_t out_bytes = off - o_bt;
        if ((rc = h->d_buf.reserve(off))) return rc;
        if ((rc = h->h_in.reserve(in_bytes))) return rc;
        if ((rc = h->h_out.reserve(out_bytes))) return rc;
        char* hi = static_cast<char*>(h->h_in.p);
        std::memcpy(hi + o_qf, q_feat.data(), 4 * (size_t)ne);
        std::memcpy(hi + o_tb, t_begin.data(), 4 * (size_t)ne);
        std::memcpy(hi + o_tc, t_count.data(), 4 * (size_t)ne);
        std::memcpy(hi + o_tf, T.feat_idx, 4 * (size_t)nt_feat);
        std::memcpy(hi + o_qd, Q.desc, 32 * (size_t)Q.n_kpts);
        std::memcpy(hi + o_td, T.desc, 32 * (size_t)T.n_kpts);
        std::memcpy(hi + o_qo, Q.octave, 4 * (size_t)Q.n_kpts);
        std::memcpy(hi + o_to, T.octave, 4 * (size_t)T.n_kpts);
        if (a->F12) { std::memcpy(hi + o_qp, Q.pt, 8 * (size_t)Q.n_kpts); std::memcpy(hi + o_tp, T.pt, 8 * (size_t)T.n_kpts); }
        if (Q.used) std::memcpy(hi + o_qu, Q.used, Q.n_kpts); else std::memset(hi + o_qu, 1, Q.n_kpts);
        if (T.used) std::memcpy(hi + o_tu, T.used, T.n_kpts); else std::memset(hi + o_tu, 1, T.n_kpts);
        if (a->F12) for (int i = 0; i < a->n_levels; i++) { const float v = a->scale_factors[i]; reinterpret_cast<float*>(hi + o_s2)[i] = v * v; }
        char* base = h->d_buf.as<char>();
        UH_HIP_CHECK(hipMemcpyAsync(base, hi, in_bytes, hipMemcpyHostToDevice, st));
        BowMatchDev D{};
        D.n_entries = ne;
        D.q_feat = (const unsigned int*)(base + o_qf); D.t_begin = (const int*)(base + o_tb); D.t_count = (const int*)(base + o_tc);
        D.t_feat = (const unsigned int*)(base + o_tf);
        D.q_desc = (const uint64_t*)(base + o_qd); D.t_desc = (const uint64_t*)(base + o_td);
        D.q_octave = (const int*)(base + o_qo); D.t_octave = (const int*)(base + o_to);
        D.q_pt = (const float*)(base + o_qp); D.t_pt = (const float*)(base + o_tp);
        D.q_used = (const unsigned char*)(base + o_qu); D.t_used = (const unsigned char*)(base + o_tu);
        D.scale2 = (const float*)(base + o_s2);
        D.have_F = a->F12 ? 1 : 0;
        if (a->F12) for (int i = 0; i < 9; i++) D.F[i] = a->F12[i];
        D.min_desc_dist = a->min_desc_dist; D.max_octave_diff = a->max_octave_diff;
        D.best_train = (int*)(base + o_bt); D.best_dist = (float*)(base + o_bd); D.best_dist2 = (float*)(base + o_b2); D.octave_best2 = (int*)(base + o_o2);
        UH_LAUNCH(h->ctx, bow_match_kernel, dim3(uh_div_up(ne, 256)), dim3(256), 0, D);
        UH_HIP_CHECK(hipGetLastError());
        char* ho = static_cast<char*>(h->h_out.p);
        UH_HIP_CHECK(hipMemcpyAsync(ho, base + o_bt, out_bytes, hipMemcpyDeviceToHost, st));
        UH_HIP_CHECK(hipStreamSynchronize(st));
        const int* bt = (const int*)ho;
        const float* bd = (const float*)(ho + (o_bd - o_bt));
        const float* b2 = (const float*)(ho + (o_b2 - o_bt));
        const int* o2 = (const int*)(ho + (o_o2 - o_bt));
        for (int e = 0; e < ne; e++) {   // framematcher.cpp:477-487, in (node, list) order
            if (bt[e] < 0) continue;
            const int bq = (int)q_feat[e];
            if (!(o2[e] == Q.octave[bq] && bd[e] > b2[e] * a->nn_match_ratio)) matches.push_back(uh_dmatch{bq, bt[e], -1, bd[e]});
        }
    }
    finish_matches(matches, T.angle, Q.angle, a->check_orientation);
    if ((int)matches.size() > cap) { uh::set_error("uh_bowmatch_match: %zu matches but capacity %d", matches.size(), cap); return UH_ECAPACITY; }
    std::copy(matches.begin(), matches.end(), out);
    return (int)matches.size();
}

}  // extern "C"
