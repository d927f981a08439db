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
