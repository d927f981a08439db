// fbow bag-of-words descent on MI355X (gfx950) behind the C ABI (uh_bow_*).
//
// Reference contract:
//   3rdparty/fbow/fbow/fbow.cpp:51-90   Vocabulary::transform(features, level, fBow&, fBow2&) -> _transform2<L1_32bytes>
//   3rdparty/fbow/fbow/fbow.h:402-447   greedy descent: per block the FIRST child of minimum Hamming distance; leaf -> word
//                                       id + weight; node id at `level` packs ceil(log2 k) bits per level
//   3rdparty/fbow/fbow/fbow.h:137-197   block layout;  fbow.cpp:171-190 stream = u64 sig 55824124 + params(120 B) + blob
//   3rdparty/fbow/fbow/fbow.cpp:192-243 fBow::score
//
// Design: one wave per descriptor, lane c = child c of the current block (k <= 64), so one level costs one coalesced
// 32-byte read per lane, 8 x (xor + bcnt) and a wave arg-min that keeps the lowest child index among equal minima.
// The vocabulary (a few MB) stays resident in HBM/L2; the maps of the reference API are assembled on the host from the
// per-descriptor (word, weight, node) triples, in feature order, which preserves the reference's float summation order.
#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "common.hpp"

namespace {

struct BowParams {   // == fbow::Vocabulary::params (fbow.h:121-131)
    char desc_name[50];
    uint32_t aligment, nblocks;
    uint64_t desc_size_bytes_wp, block_size_bytes_wp, feature_off_start, child_off_start, total_size;
    int32_t desc_type, desc_size;
    uint32_t m_k;
};
static_assert(sizeof(BowParams) == 120, "fbow params layout");

__global__ __launch_bounds__(256) void bow_transform_kernel(const uint8_t* __restrict__ blob, uint32_t nblocks, uint64_t block_size,
                                                            uint64_t feature_off, uint64_t child_off, uint64_t desc_wp, int nbits,
                                                            const uint8_t* __restrict__ desc, int n, int level,
                                                            uint32_t* __restrict__ word, float* __restrict__ weight,
                                                            uint32_t* __restrict__ node, uint8_t* __restrict__ valid) {
    const int lane = threadIdx.x & 63;
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (f >= n) return;
    uint32_t q[8];
    const uint32_t* qp = reinterpret_cast<const uint32_t*>(desc + (size_t)f * 32);
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = __builtin_amdgcn_readfirstlane(qp[j]);
    const uint8_t* block = blob;
    uint32_t lvl = 0, curNode = 0, w_word = 0xFFFFFFFFu, w_node = 0;
    float w_weight = 0.f;
    int w_valid = 0;
    uint32_t best_c = 0;   // fbow carries the last arg-min into an empty block (fbow.h:411-426)
    for (;;) {
        const int N = *reinterpret_cast<const uint16_t*>(block);
        unsigned d = 0xFFFFFFFFu;
        if (lane < N) {
            const uint32_t* fp = reinterpret_cast<const uint32_t*>(block + feature_off + (size_t)lane * desc_wp);
            d = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) d += __popc(fp[j] ^ q[j]);
        }
        // arg-min, lowest lane among equal distances: key = d << 6 | lane
        unsigned long long key = ((unsigned long long)d << 6) | (unsigned)lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(key, o);
            key = other < key ? other : key;
        }
        if (N > 0) best_c = (uint32_t)(key & 63);
        if (lvl == (uint32_t)level) { w_node = curNode; w_valid = 1; }
        const uint32_t* info = reinterpret_cast<const uint32_t*>(block + child_off + (size_t)best_c * 8);
        const uint32_t idc = info[0];
        const bool isleaf = (idc & 0x80000000u) != 0;
        const uint32_t id = idc & 0x7FFFFFFFu;
        if (isleaf) {
            w_word = id;
            w_weight = __uint_as_float(info[1]);
            if (lvl < (uint32_t)level) { w_node = curNode; w_valid = 1; }
            break;
        }
        if (id >= nblocks) break;   // corrupt vocabulary: stop instead of reading outside the blob
        block = blob + (size_t)id * block_size;
        curNode = (curNode << nbits) | best_c;
        lvl++;
        if (id == 0) break;          // fbow's loop condition `getId()!=0`
    }
    if (lane == 0) { word[f] = w_word; weight[f] = w_weight; node[f] = w_node; valid[f] = (uint8_t)w_valid; }
}

}  // namespace

struct uh_bow {
    uh_ctx* ctx = nullptr;
    BowParams P{};
    bool loaded = false;
    uh::DevBuf d_blob, d_desc, d_word, d_weight, d_node, d_valid;
};

extern "C" {

int uh_bow_create(uh_ctx* ctx, uh_bow** out) {
    UH_REQUIRE(ctx && out, "uh_bow_create: NULL argument");
    uh_bow* b = new uh_bow();
    b->ctx = ctx;
    *out = b;
    return UH_OK;
}
void uh_bow_destroy(uh_bow* b) { delete b; }

int uh_bow_set(uh_bow* b, const void* params120, const void* blob) {
    UH_REQUIRE(b && params120 && blob, "uh_bow_set: NULL argument");
    BowParams P;
    memcpy(&P, params120, sizeof(P));
    UH_REQUIRE(P.desc_type == 0 && P.desc_size == 32, "Vocabulary: only CV_8UC1 32-byte (ORB) vocabularies are supported (type %d size %d)", P.desc_type, P.desc_size);
    UH_REQUIRE(P.m_k >= 1 && P.m_k <= 64, "Vocabulary: branching factor %u outside [1,64]", P.m_k);
    UH_REQUIRE(P.nblocks >= 1 && P.total_size == P.block_size_bytes_wp * P.nblocks, "Vocabulary: inconsistent sizes");
    UH_REQUIRE(P.desc_size_bytes_wp >= 32 && P.desc_size_bytes_wp % 4 == 0 && P.feature_off_start % 4 == 0 && P.child_off_start % 4 == 0 &&
               P.block_size_bytes_wp % 4 == 0 && P.child_off_start + 8ull * P.m_k <= P.block_size_bytes_wp, "Vocabulary: bad block layout");
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    int rc = b->d_blob.reserve(P.total_size);
    if (rc) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(b->d_blob.p, blob, P.total_size, hipMemcpyHostToDevice, b->ctx->stream));
    UH_HIP_CHECK(hipStreamSynchronize(b->ctx->stream));
    b->P = P;
    b->loaded = true;
    return UH_OK;
}

// Vocabulary::fromStream (fbow.cpp:181-190)
int uh_bow_load(uh_bow* b, const void* stream, size_t nbytes) {
    UH_REQUIRE(b && stream, "uh_bow_load: NULL argument");
    UH_REQUIRE(nbytes >= 8 + sizeof(BowParams), "Vocabulary::fromStream: truncated stream");
    uint64_t sig;
    memcpy(&sig, stream, 8);
    UH_REQUIRE(sig == 55824124ull, "Vocabulary::fromStream invalid signature");
    BowParams P;
    memcpy(&P, (const char*)stream + 8, sizeof(P));
    UH_REQUIRE(nbytes >= 8 + sizeof(BowParams) + P.total_size, "Vocabulary::fromStream: truncated blob");
    return uh_bow_set(b, &P, (const char*)stream + 8 + sizeof(BowParams));
}

int uh_bow_get_params(const uh_bow* b, void* params120) {
    UH_REQUIRE(b && params120 && b->loaded, "uh_bow_get_params: vocabulary not loaded");
    memcpy(params120, &b->P, sizeof(BowParams));
    return UH_OK;
}

int uh_bow_transform_dev(uh_bow* b, const uint8_t* d_desc, int n, int level, uint32_t* d_word, float* d_weight, uint32_t* d_node,
                         uint8_t* d_valid) {
    UH_REQUIRE(b && b->loaded, "Vocabulary::transform: vocabulary not loaded");
    UH_REQUIRE(n >= 1, "Vocabulary::transform No input data");   // fbow.cpp:52
    UH_REQUIRE(d_desc && d_word && d_weight && d_node && d_valid, "uh_bow_transform_dev: NULL buffer");
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    const BowParams& P = b->P;
    const int nbits = (int)std::ceil(std::log2((double)P.m_k));
    UH_LAUNCH(b->ctx, bow_transform_kernel, dim3(uh_div_up(n, 4)), dim3(256), 0, b->d_blob.as<uint8_t>(), P.nblocks, P.block_size_bytes_wp,
              P.feature_off_start, P.child_off_start, P.desc_size_bytes_wp, nbits, d_desc, n, level, d_word, d_weight, d_node, d_valid);
    UH_HIP_CHECK(hipGetLastError());
    return UH_OK;
}

int uh_bow_transform(uh_bow* b, const uint8_t* desc, int n, size_t stride, int desc_bytes, int level, uint32_t* word, float* weight,
                     uint32_t* node, uint8_t* valid) {
    UH_REQUIRE(b && b->loaded, "Vocabulary::transform: vocabulary not loaded");
    UH_REQUIRE(n >= 1 && desc, "Vocabulary::transform No input data");
    UH_REQUIRE(desc_bytes == b->P.desc_size, "Vocabulary::transform features are of different size than the vocabulary ones");   // fbow.cpp:54
    UH_REQUIRE(stride >= 32 && word && weight && node && valid, "uh_bow_transform: bad arguments");
    int rc;
    hipStream_t st = b->ctx->stream;
    UH_HIP_CHECK(hipSetDevice(b->ctx->device));
    if ((rc = b->d_desc.reserve((size_t)n * 32))) return rc;
    if ((rc = b->d_word.reserve((size_t)n * 4))) return rc;
    if ((rc = b->d_weight.reserve((size_t)n * 4))) return rc;
    if ((rc = b->d_node.reserve((size_t)n * 4))) return rc;
    if ((rc = b->d_valid.reserve((size_t)n))) return rc;
    UH_HIP_CHECK(hipMemcpy2DAsync(b->d_desc.p, 32, desc, stride, 32, (size_t)n, hipMemcpyHostToDevice, st));
    if ((rc = uh_bow_transform_dev(b, b->d_desc.as<uint8_t>(), n, level, b->d_word.as<uint32_t>(), b->d_weight.as<float>(),
                                   b->d_node.as<uint32_t>(), b->d_valid.as<uint8_t>()))) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(word, b->d_word.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(weight, b->d_weight.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(node, b->d_node.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(valid, b->d_valid.p, (size_t)n, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    return UH_OK;
}

// fBow::score (fbow.cpp:192-243): host arithmetic on two sorted sparse vectors (a handful of flops; not GPU work)
double uh_bow_score(const uint32_t* ids1, const float* w1, int n1, const uint32_t* ids2, const float* w2, int n2) {
    int i = 0, j = 0;
    double score = 0;
    while (i < n1 && j < n2) {
        if (ids1[i] == ids2[j]) { score += w1[i] * w2[j]; ++i; ++j; }
        else if (ids1[i] < ids2[j]) { while (i < n1 && ids1[i] < ids2[j]) ++i; }
        else { while (j < n2 && ids2[j] < ids1[i]) ++j; }
    }
    if (score >= 1) score = 1.0; else score = 1.0 - std::sqrt(1.0 - score);
    return score;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ keyframe database
// KPFrameDataBase (src/map_types/keyframedatabase.cpp): the bags of words of the keyframes and the first half of
// relocalizationCandidates (:195-238) — how many of the query's words every keyframe shares, and fBow::score against those
// that share enough.  The reference walks an inverted word -> frames index on the host; here every keyframe's sorted bag is
// resident in HBM and one lane per keyframe merge-joins it with the query (the products are added in ascending word order, as
// fBow::score does, so the double sum is the reference's).  The covisibility accumulation that follows (:241-275) needs
// CovisGraph and stays with the caller.
namespace {

struct BowDbDev {
    int n_frames;
    const int* ptr;              // n_frames + 1
    const uint32_t* words; const float* weights;
    int nq; const uint32_t* q_words; const float* q_weights;
    uint32_t* nobs; double* score;
};

__global__ __launch_bounds__(64) void bowdb_query_kernel(BowDbDev a) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= a.n_frames) return;
    int i = 0, j = a.ptr[f];
    const int n1 = a.nq, n2 = a.ptr[f + 1];
    double score = 0;
    uint32_t common = 0;
    while (i < n1 && j < n2) {
        const uint32_t w1 = a.q_words[i], w2 = a.words[j];
        if (w1 == w2) { score += a.q_weights[i] * a.weights[j]; ++common; ++i; ++j; }
        else if (w1 < w2) ++i;
        else ++j;
    }
    a.nobs[f] = common;
    a.score[f] = score >= 1 ? 1.0 : 1.0 - sqrt(1.0 - score);
}

}  // namespace

struct uh_bowdb {
    uh_ctx* ctx = nullptr;
    std::map<uint32_t, std::pair<std::vector<uint32_t>, std::vector<float>>> frames;   // ascending frame id, like the reference's maps
    bool dirty = true;
    uh::DevBuf d_db, d_q, d_out;
    std::vector<uint32_t> ids;   // frame id of database row i (as uploaded)
    std::vector<int> ptr;
};

extern "C" {

int uh_bowdb_create(uh_ctx* ctx, uh_bowdb** out) {
    UH_REQUIRE(ctx && out, "uh_bowdb_create: NULL argument");
    uh_bowdb* d = new uh_bowdb();
    d->ctx = ctx;
    *out = d;
    return UH_OK;
}

void uh_bowdb_destroy(uh_bowdb* d) { delete d; }

int uh_bowdb_size(const uh_bowdb* d) { return d ? (int)d->frames.size() : 0; }

// KPFrameDataBase::add (:150-161); the bag is the fBow of uh_bow_transform(desc, 3): words ascending, raw weights
int uh_bowdb_add(uh_bowdb* d, uint32_t frame_id, const uint32_t* words, const float* weights, int n) {
    UH_REQUIRE(d && n >= 0 && (n == 0 || (words && weights)), "uh_bowdb_add: bad arguments");
    UH_REQUIRE(d->frames.count(frame_id) == 0, "uh_bowdb_add: frame %u is already in the database", frame_id);
    for (int i = 1; i < n; i++) UH_REQUIRE(words[i] > words[i - 1], "uh_bowdb_add: words must ascend (std::map order)");
    d->frames[frame_id] = {std::vector<uint32_t>(words, words + n), std::vector<float>(weights, weights + n)};
    d->dirty = true;
    return UH_OK;
}

// KPFrameDataBase::del (:163-180)
int uh_bowdb_del(uh_bowdb* d, uint32_t frame_id) {
    UH_REQUIRE(d, "uh_bowdb_del: NULL");
    UH_REQUIRE(d->frames.erase(frame_id) == 1, "uh_bowdb_del: frame %u is not in the database", frame_id);
    d->dirty = true;
    return UH_OK;
}

// relocalizationCandidates :195-238: returns the frames of `frame_score` (more than 0.8 * max common words shared AND score >
// min_score), ascending frame id, with their common-word counts and scores; < 0 on error
int uh_bowdb_query(uh_bowdb* d, const uint32_t* words, const float* weights, int n, const uint32_t* excluded, int n_excluded,
                   float min_score, uint32_t* frame_ids_out, uint32_t* nobs_out, double* score_out, int cap) {
    UH_REQUIRE(d && n >= 0 && (n == 0 || (words && weights)) && n_excluded >= 0 && (n_excluded == 0 || excluded), "uh_bowdb_query: bad arguments");
    const int nf = (int)d->frames.size();
    if (nf == 0 || n == 0) return 0;
    int rc;
    UH_HIP_CHECK(hipSetDevice(d->ctx->device));
    hipStream_t st = d->ctx->stream;
    if (d->dirty) {
        d->ids.clear(); d->ptr.assign(1, 0);
        std::vector<uint32_t> w; std::vector<float> wt;
        for (auto& kv : d->frames) {
            d->ids.push_back(kv.first);
            w.insert(w.end(), kv.second.first.begin(), kv.second.first.end());
            wt.insert(wt.end(), kv.second.second.begin(), kv.second.second.end());
            d->ptr.push_back((int)w.size());
        }
        const size_t o_w = ((size_t)(nf + 1) * 4 + 255) & ~(size_t)255, o_wt = (o_w + w.size() * 4 + 255) & ~(size_t)255;
        if ((rc = d->d_db.reserve(o_wt + wt.size() * 4 + 256))) return rc;
        char* base = d->d_db.as<char>();
        UH_HIP_CHECK(hipMemcpyAsync(base, d->ptr.data(), (size_t)(nf + 1) * 4, hipMemcpyHostToDevice, st));
        if (!w.empty()) {
            UH_HIP_CHECK(hipMemcpyAsync(base + o_w, w.data(), w.size() * 4, hipMemcpyHostToDevice, st));
            UH_HIP_CHECK(hipMemcpyAsync(base + o_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, st));
        }
        UH_HIP_CHECK(hipStreamSynchronize(st));   // the staging vectors die here
        d->dirty = false;
    }
    const size_t total_words = (size_t)d->ptr.back();
    const size_t o_w = ((size_t)(nf + 1) * 4 + 255) & ~(size_t)255, o_wt = (o_w + total_words * 4 + 255) & ~(size_t)255;
    const size_t oq_w = 0, oq_wt = ((size_t)n * 4 + 255) & ~(size_t)255;
    if ((rc = d->d_q.reserve(oq_wt + (size_t)n * 4 + 256))) return rc;
    const size_t oo_s = ((size_t)nf * 4 + 255) & ~(size_t)255;
    if ((rc = d->d_out.reserve(oo_s + (size_t)nf * 8 + 256))) return rc;
    UH_HIP_CHECK(hipMemcpyAsync(d->d_q.as<char>() + oq_w, words, (size_t)n * 4, hipMemcpyHostToDevice, st));
    UH_HIP_CHECK(hipMemcpyAsync(d->d_q.as<char>() + oq_wt, weights, (size_t)n * 4, hipMemcpyHostToDevice, st));
    BowDbDev a;
    char* base = d->d_db.as<char>();
    a.n_frames = nf; a.ptr = (const int*)base; a.words = (const uint32_t*)(base + o_w); a.weights = (const float*)(base + o_wt);
    a.nq = n; a.q_words = (const uint32_t*)(d->d_q.as<char>() + oq_w); a.q_weights = (const float*)(d->d_q.as<char>() + oq_wt);
    a.nobs = (uint32_t*)d->d_out.as<char>(); a.score = (double*)(d->d_out.as<char>() + oo_s);
    UH_LAUNCH(d->ctx, bowdb_query_kernel, dim3(uh_div_up(nf, 64)), dim3(64), 0, a);
    UH_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> nobs(nf);
    std::vector<double> score(nf);
    UH_HIP_CHECK(hipMemcpyAsync(nobs.data(), a.nobs, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipMemcpyAsync(score.data(), a.score, (size_t)nf * 8, hipMemcpyDeviceToHost, st));
    UH_HIP_CHECK(hipStreamSynchronize(st));
    std::set<uint32_t> excl(excluded, excluded + n_excluded);
    uint32_t maxCommon = 0;
    for (int i = 0; i < nf; i++) if (nobs[i] > 0 && !excl.count(d->ids[i])) maxCommon = std::max(maxCommon, nobs[i]);
    const uint32_t minCommon = (uint32_t)(maxCommon * 0.8f);   // :224
    int k = 0;
    for (int i = 0; i < nf; i++) {
        if (nobs[i] == 0 || excl.count(d->ids[i])) continue;   // frame_nobs has an entry only for frames that share a word (:212-221)
        if (nobs[i] > minCommon && score[i] > min_score) {
            UH_REQUIRE(k < cap, "uh_bowdb_query: more than %d candidate frames", cap);
            frame_ids_out[k] = d->ids[i]; if (nobs_out) nobs_out[k] = nobs[i]; if (score_out) score_out[k] = score[i];
            k++;
        }
    }
    return k;
}

}  // extern "C"
