// TEST INFRASTRUCTURE ONLY — builds only where OpenCV's development files exist (fbow.h includes <opencv2/core/core.hpp>; this image has
// none, so this file has never been compiled here: `make -C oracle ref-fbow`, used by tests/golden/make_fbow_golden.py).
// A thin C driver around the REAL fbow (3rdparty/fbow/fbow/fbow.cpp, compiled from /root/reference where it lies): loads a vocabulary
// stream, runs Vocabulary::transform(features, level, fBow&, fBow2&) (fbow.cpp:51-90) and fBow::score (fbow.cpp:192-243) and flattens
// the two maps for the caller — the vectors that pin oracle/bow_oracle.cpp and csrc/bow.hip.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <opencv2/core/core.hpp>
#include "fbow/fbow.h"

extern "C" {

// stream = the vocabulary file's bytes (u64 signature 55824124 + params struct + block data, fbow.cpp:171-190).
// Returns the number of (word, weight) entries of the bag, or -1; the outputs are written up to their capacities:
//   bag_ids / bag_w [cap_bag], node_ids / node_ptr [cap_nodes (+1)], node_feats [cap_feats]; *n_nodes, *n_feats, *self_score.
int fbow_ref_transform(const void* stream, size_t nbytes, const uint8_t* desc, int n, int desc_bytes, int level,
                       uint32_t* bag_ids, float* bag_w, int cap_bag, uint32_t* node_ids, int32_t* node_ptr, int cap_nodes,
                       uint32_t* node_feats, int cap_feats, int* n_nodes, int* n_feats, double* self_score) {
    try {
        std::stringstream ss(std::string(static_cast<const char*>(stream), nbytes), std::ios::in | std::ios::binary);
        fbow::Vocabulary voc;
        voc.fromStream(ss);
        cv::Mat feats(n, desc_bytes, CV_8UC1);
        for (int i = 0; i < n; i++) std::memcpy(feats.ptr<uint8_t>(i), desc + (size_t)i * desc_bytes, desc_bytes);
        fbow::fBow bag;
        fbow::fBow2 nodes;
        voc.transform(feats, level, bag, nodes);
        int nb = 0;
        for (const auto& e : bag) { if (nb < cap_bag) { bag_ids[nb] = e.first; bag_w[nb] = (float)e.second; } nb++; }
        int nn = 0, nf = 0;
        node_ptr[0] = 0;
        for (const auto& e : nodes) {
            if (nn < cap_nodes) node_ids[nn] = e.first;
            for (uint32_t f : e.second) { if (nf < cap_feats) node_feats[nf] = f; nf++; }
            nn++;
            if (nn <= cap_nodes) node_ptr[nn] = nf;
        }
        *n_nodes = nn; *n_feats = nf;
        const fbow::fBow normed = voc.transform(feats);   // the L2-normalised bag (fbow.cpp:92-143)
        *self_score = fbow::fBow::score(normed, normed);
        return nb;
    } catch (...) { return -1; }
}

}  // extern "C"
