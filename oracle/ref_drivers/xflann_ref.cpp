// TEST INFRASTRUCTURE ONLY — thin C driver around the REAL reference xflann (compiled from
// /root/reference/3rdparty/xflann where it lies; output goes to oracle/_ref/, never into git).
// Used to pin oracle/knn_oracle.cpp and to generate tests/golden/knn_*.npz (tests/golden/make_knn_golden.py).
#include <xflann/xflann.h>
#include <cstdint>

extern "C" int xflann_ref_linear_search(const uint8_t* train, int nt, const uint8_t* queries, int nq, int nn, int sorted,
                                        int threads, int32_t* indices, int32_t* distances) {
    try {
        // caller-owned buffers only: an owning xflann::Matrix double-frees when passed by value
        xflann::Matrix T(XFLANN_8U, nt, 32, train);
        xflann::Matrix Q(XFLANN_8U, nq, 32, queries);
        xflann::Matrix I(XFLANN_32S, nq, nn, indices);
        xflann::Matrix D(XFLANN_32S, nq, nn, distances);
        xflann::Index index;
        index.build(T, xflann::LinearParams(1));
        bool ok = index.search(Q, nn, I, D, xflann::KnnSearchParams(-1, sorted, threads));
        return ok ? 0 : 1;
    } catch (const std::exception&) {
        return -1;
    }
}

// ---- hierarchical k-means index, the way FrameMatcher_Flann uses it (framematcher.cpp:213,239): build(HKMeansParams(k, maxIters)),
// search(KnnSearchParams(maxChecks, sorted)).
#include <sstream>
#include <cstring>

// builds the index and returns its serialised form (xflann::Index::toStream: 16-byte Index header, then KMeansIndex::toStream =
// 8-byte signature, the 40-byte params struct, the block data).  Returns the stream size, or -1; writes at most cap bytes.
extern "C" long xflann_ref_hkmeans_stream(const uint8_t* train, int nt, int k, int max_iters, uint8_t* out, long cap) {
    try {
        xflann::Matrix T(XFLANN_8U, nt, 32, train);
        xflann::Index index;
        index.build(T, xflann::HKMeansParams(k, max_iters));
        std::ostringstream ss(std::ios::binary);
        index.toStream(ss);
        const std::string s = ss.str();
        if (out) std::memcpy(out, s.data(), (size_t)std::min<long>(cap, (long)s.size()));
        return (long)s.size();
    } catch (const std::exception&) {
        return -1;
    }
}

extern "C" int xflann_ref_hkmeans_search(const uint8_t* train, int nt, const uint8_t* queries, int nq, int nn, int k, int max_iters,
                                         int max_checks, int sorted, int32_t* indices, int32_t* distances) {
    try {
        xflann::Matrix T(XFLANN_8U, nt, 32, train);
        xflann::Matrix Q(XFLANN_8U, nq, 32, queries);
        xflann::Matrix I(XFLANN_32S, nq, nn, indices);
        xflann::Matrix D(XFLANN_32S, nq, nn, distances);
        xflann::Index index;
        index.build(T, xflann::HKMeansParams(k, max_iters));
        bool ok = index.search(Q, nn, I, D, xflann::KnnSearchParams(max_checks, sorted, 1));
        return ok ? 0 : 1;
    } catch (const std::exception&) {
        return -1;
    }
}
