// TEST INFRASTRUCTURE ONLY — CPU oracle for the Hamming kNN stage.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// Restates, in plain sequential C++, what xflann::Index(Linear) computes for 32-byte binary features:
//   3rdparty/xflann/xflann/impl/linear.h:68-86      scan order, fill of unfilled slots
//   3rdparty/xflann/xflann/impl/resultset.h:64-135  ResultSet::push / down / up (max-heap in the output row)
//   3rdparty/xflann/xflann/impl/distances.h:279-283 d_Hamming_x64_32bytes
//   3rdparty/xflann/xflann/index.h:119-134          Index::sort (exchange sort incl. its handling of -1 slots)
// Pinned against the real xflann build (oracle/_ref/libxflann_ref.so) by tests/test_knn_oracle.py.
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace {

inline int hamming32(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

struct Row {
    int32_t* dist;
    int32_t* idx;
    int n = 0;     // array_size
    int cap;       // maxSize
    int maxv;      // radius bound, <0 = none
    void swp(int a, int b) {
        int32_t t = dist[a]; dist[a] = dist[b]; dist[b] = t;
        t = idx[a]; idx[a] = idx[b]; idx[b] = t;
    }
    void toward_root(int i) {                 // resultset.h "down"
        while (i != 0) {
            int p = (i - 1) / 2;
            if (dist[p] < dist[i]) { swp(i, p); i = p; } else return;
        }
    }
    void toward_leaves(int i) {               // resultset.h "up"
        for (;;) {
            int l = 2 * i + 1, r = 2 * i + 2;
            if (l >= n) return;
            if (r >= n) { if (dist[i] < dist[l]) swp(i, l); return; }
            if (dist[r] < dist[l]) { if (dist[i] < dist[l]) { swp(i, l); i = l; } else return; }
            else                   { if (dist[i] < dist[r]) { swp(i, r); i = r; } else return; }
        }
    }
    void push(int d, int i) {
        if (maxv >= 0 && maxv < d) return;
        if (n >= cap) {
            if (d < dist[0]) { swp(0, n - 1); n--; if (n > 1) toward_leaves(0); }
            else return;
        }
        dist[n] = d; idx[n] = i;
        if (n > 0) toward_root(n);
        n++;
    }
};

}  // namespace

extern "C" {

// t_begin/t_end: scan only train rows [t_begin,t_end) (indices stay global) — used by shard tests.
int oracle_knn_search(const uint8_t* train, int nt, size_t t_stride, const uint8_t* queries, int nq, size_t q_stride,
                      int nn, int sorted, int max_dist, int t_begin, int t_end, int32_t* indices, int32_t* distances) {
    if (nn < 1 || nt < 0 || nq < 0) return -1;
    if (t_begin < 0) t_begin = 0;
    if (t_end < 0 || t_end > nt) t_end = nt;
    for (int q = 0; q < nq; ++q) {
        Row r{distances + (size_t)q * nn, indices + (size_t)q * nn, 0, nn, max_dist};
        const uint8_t* f = queries + (size_t)q * q_stride;
        for (int i = t_begin; i < t_end; ++i) r.push(hamming32(f, train + (size_t)i * t_stride), i);
        for (int i = r.n; i < nn; ++i) { r.idx[i] = -1; r.dist[i] = 0; }
        if (sorted) {
            for (int i = 0; i < nn - 1; ++i)
                if (r.idx[i] != -1)
                    for (int j = i + 1; j < nn; ++j)
                        if (r.dist[i] > r.dist[j]) r.swp(i, j);
        }
    }
    return 0;
}

int oracle_hamming32(const uint8_t* a, const uint8_t* b) { return hamming32(a, b); }

}  // extern "C"
