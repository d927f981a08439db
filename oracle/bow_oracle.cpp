// TEST INFRASTRUCTURE ONLY — CPU oracle for the fbow bag-of-words stage.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// PARITY UNPINNED: fbow's sources include <opencv2/core/core.hpp> (3rdparty/fbow/fbow/fbow.h:5) which this image lacks, the
// reference ships no fbow test vectors and the ORB vocabulary blob is absent (.MISSING_LARGE_BLOBS).  This file restates
//   3rdparty/fbow/fbow/fbow.h:402-447   Vocabulary::_transform2<L1_32bytes>  (greedy first-minimum Hamming descent,
//                                        word weight accumulation, node id at `storeLevel`)
//   3rdparty/fbow/fbow/fbow.h:369-399   Vocabulary::_transform  + fbow.cpp:133-143 (L2 normalisation of the 1-arg transform)
//   3rdparty/fbow/fbow/fbow.h:137-197   block layout,  fbow.cpp:10-49 setParams sizes,  fbow.cpp:171-190 stream format
//   3rdparty/fbow/fbow/fbow.cpp:192-243 fBow::score
// and is checked by hand-built vocabularies with known answers (tests/test_bow.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {

struct Params {   // fbow::Vocabulary::params, natural x86-64 layout (sizeof == 120)
    char desc_name[50];
    uint32_t aligment, nblocks;
    uint64_t desc_size_bytes_wp, block_size_bytes_wp, feature_off_start, child_off_start, total_size;
    int32_t desc_type, desc_size;
    uint32_t m_k;
};
static_assert(sizeof(Params) == 120, "fbow params layout");

inline int hamming32(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

}  // namespace

extern "C" {

// Per-feature outputs (the maps are assembled by the caller in feature order):
//   word[i]   leaf word id reached, 0xFFFFFFFF if the descent ended without a leaf
//   weight[i] that leaf's weight
//   node[i]   node id recorded at `level` (or at the leaf's parent level when the leaf comes earlier), valid[i] says if any
int oracle_bow_transform(const void* params_ptr, const uint8_t* blob, const uint8_t* desc, int n, size_t stride, int level,
                         uint32_t* word, float* weight, uint32_t* node, uint8_t* valid) {
    const Params& P = *reinterpret_cast<const Params*>(params_ptr);
    if (P.desc_size != 32 || P.desc_type != 0) return -1;
    const int nbits = (int)std::ceil(std::log2((double)P.m_k));
    uint32_t best_second = 0;   // fbow keeps best_dist_idx.second across blocks and features (only matters for empty blocks)
    for (int f = 0; f < n; f++) {
        const uint8_t* feat = desc + (size_t)f * stride;
        const uint8_t* block = blob;
        uint32_t lvl = 0, curNode = 0;
        word[f] = 0xFFFFFFFFu; weight[f] = 0.f; node[f] = 0; valid[f] = 0;
        bool isleaf;
        uint32_t id;
        do {
            uint64_t best = 0xFFFFFFFFull;
            const int N = *reinterpret_cast<const uint16_t*>(block);
            for (int c = 0; c < N; c++) {
                const uint64_t d = (uint64_t)hamming32(feat, block + P.feature_off_start + (size_t)c * P.desc_size_bytes_wp);
                if (d < best) { best = d; best_second = (uint32_t)c; }
            }
            if (lvl == (uint32_t)level) { node[f] = curNode; valid[f] = 1; }
            const uint8_t* info = block + P.child_off_start + (size_t)best_second * 8;
            uint32_t idc;
            float w;
            std::memcpy(&idc, info, 4);
            std::memcpy(&w, info + 4, 4);
            isleaf = (idc & 0x80000000u) != 0;
            id = idc & 0x7FFFFFFFu;
            if (isleaf) {
                word[f] = id; weight[f] = w;
                if (lvl < (uint32_t)level) { node[f] = curNode; valid[f] = 1; }
                break;
            }
            block = blob + (size_t)id * P.block_size_bytes_wp;
            curNode = (curNode << nbits) | best_second;
            lvl++;
        } while (!isleaf && id != 0);
    }
    return 0;
}

// fBow::score on two sorted sparse vectors
double oracle_bow_score(const uint32_t* ids1, const float* w1, int n1, const uint32_t* ids2, const float* w2, int n2) {
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

// sizes of setParams (fbow.cpp:10-49) for a writer of synthetic vocabularies
void oracle_bow_make_params(int aligment, int k, int desc_size, int nblocks, void* params_out) {
    Params P;
    std::memset(&P, 0, sizeof(P));
    std::strcpy(P.desc_name, "orb");
    P.aligment = aligment; P.m_k = k; P.desc_type = 0; P.desc_size = desc_size; P.nblocks = nblocks;
    uint64_t al = desc_size / aligment; if (desc_size % aligment) al++;
    P.desc_size_bytes_wp = al * aligment;
    int fo = 8 / aligment; if (8 % aligment) fo++;
    P.feature_off_start = (uint64_t)fo * aligment;
    P.child_off_start = P.feature_off_start + (uint64_t)k * P.desc_size_bytes_wp;
    uint64_t bs = P.feature_off_start + (uint64_t)k * (P.desc_size_bytes_wp + 8);
    uint64_t ba = bs / aligment; if (bs % aligment) ba++;
    P.block_size_bytes_wp = ba * aligment;
    P.total_size = P.block_size_bytes_wp * nblocks;
    std::memcpy(params_out, &P, sizeof(P));
}

}  // extern "C"
