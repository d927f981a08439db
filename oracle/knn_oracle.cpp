// TEST INFRASTRUCTURE ONLY — CPU oracle for the Hamming kNN stage.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// Restates, in plain sequential C++, what xflann::Index(Linear) computes for 32-byte binary features:
//   3rdparty/xflann/xflann/impl/linear.h:68-86      scan order, fill of unfilled slots
//   3rdparty/xflann/xflann/impl/resultset.h:64-135  ResultSet::push / down / up (max-heap in the output row)
//   3rdparty/xflann/xflann/impl/distances.h:279-283 d_Hamming_x64_32bytes
//   3rdparty/xflann/xflann/index.h:119-134          Index::sort (exchange sort incl. its handling of -1 slots)
// Pinned against the real xflann build (oracle/_ref/libxflann_ref.so) by tests/test_knn_oracle.py.
//
// Second part: the hierarchical k-means index FrameMatcher_Flann really uses (framematcher.cpp:213 HKMeansParams(32,0), :239
// KnnSearchParams(16,false)), 32-byte Hamming only:
//   impl/kmeansindexcreator.h:228-302   createNode: std::shuffle with a fresh default std::mt19937, first k mutually distinct
//                                       rows as centres (:324-366), nearest-centre assignment with first-minimum ties (:305-322),
//                                       maxIters k-means rounds (none for maxIters=0), empty clusters dropped, recursion on
//                                       clusters larger than k
//   impl/kmeansindexcreator.cpp:235-300 convert(): breadth-first block layout (8-byte header, 8-byte node infos, 32-byte features)
//   impl/kmeansindex.h:356-410          _knnsearch_nn: best-bin-first with maxChecks, branch min-heap (impl/heap.h), ResultSet
// Pinned against the real thing by tests/test_hkmeans_oracle.py: the serialised block data is compared byte for byte and the
// search rows element for element.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <deque>
#include <limits>
#include <memory>
#include <random>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------- hierarchical k-means
namespace {

struct KmNode {
    std::vector<std::unique_ptr<KmNode>> children;
    std::vector<uint32_t> assign;
    uint8_t feature[32] = {0};   // centre: a train row, or the bitwise majority of the cluster after k-means rounds
};

struct KmBuilder {
    const uint8_t* train;
    int k, max_iters;
    bool unbounded = false;   // > k identical rows: the reference recurses without end
    int depth_guard = 0;

    void assign_points(KmNode* parent) {   // assignPointsToChildren, kmeansindexcreator.h:305-322
        for (auto& ch : parent->children) ch->assign.clear();
        for (auto fi : parent->assign) {
            KmNode* best = nullptr;
            float bestd = std::numeric_limits<float>::max();
            for (auto& ch : parent->children) {
                const int32_t d = hamming32(ch->feature, train + 32 * (size_t)fi);
                if (d < bestd) { best = ch.get(); bestd = (float)d; }
                if (bestd < 1e-16) break;
            }
            best->assign.push_back(fi);
        }
    }
    static void majority(const uint8_t* train, const std::vector<uint32_t>& idx, uint8_t* out) {   // MC_binary_generic :390-422
        int sum[256] = {0};
        for (auto i : idx) {
            const uint8_t* p = train + 32 * (size_t)i;
            for (int j = 0; j < 32; j++)
                for (int b = 0; b < 8; b++) if (p[j] & (128 >> b)) ++sum[j * 8 + b];
        }
        std::memset(out, 0, 32);
        const int N2 = (int)idx.size() / 2 + (int)(idx.size() % 2);
        for (int i = 0; i < 256; i++) if (sum[i] >= N2) out[i / 8] |= (uint8_t)(1 << (7 - (i % 8)));
    }
    static size_t vhash(const KmNode* parent) {   // kmeansindexcreator.cpp:199-205
        size_t seed = 0;
        for (auto& ch : parent->children)
            for (auto id : ch->assign) seed ^= id + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }

    void create(KmNode* parent) {
        if (++depth_guard > 64) { unbounded = true; --depth_guard; return; }
        std::mt19937 g;
        std::shuffle(parent->assign.begin(), parent->assign.end(), g);
        std::vector<uint32_t> centers;
        size_t next = 0;
        while ((int)centers.size() < k) {   // getInitialClusterCenters
            int32_t sq;
            int sel = -1;
            bool out = false;
            do {
                if (next == parent->assign.size()) { out = true; break; }
                sel = (int)parent->assign[next++];
                sq = std::numeric_limits<int32_t>::max();
                for (auto c : centers) sq = std::min(hamming32(train + 32 * (size_t)sel, train + 32 * (size_t)c), sq);
            } while (sq < 1e-16);
            if (out) break;
            centers.push_back((uint32_t)sel);
        }
        for (auto c : centers) {
            parent->children.emplace_back(new KmNode());
            std::memcpy(parent->children.back()->feature, train + 32 * (size_t)c, 32);
        }
        assign_points(parent);
        // k-means rounds (:245-262): move the centres to the bitwise majority of their clusters until the assignment hash repeats
        size_t prev_hash = 0, cur_hash = 1, niters = 0;
        while (cur_hash != prev_hash && ((max_iters == -1) || (max_iters != -1 && niters++ < (size_t)max_iters))) {
            std::swap(prev_hash, cur_hash);
            int ci = 0;
            for (auto& ch : parent->children) {
                if (ch->assign.empty()) ch->assign.push_back(centers[ci]);
                majority(train, ch->assign, ch->feature);
                ci++;
            }
            assign_points(parent);
            cur_hash = vhash(parent);
        }
        parent->assign.clear();
        for (auto it = parent->children.begin(); it != parent->children.end();) {
            if ((*it)->assign.empty()) it = parent->children.erase(it); else ++it;
        }
        if (parent->children.size() == 1 && (int)parent->children[0]->assign.size() > k) {
            // all rows fell into one cluster: the same split would repeat for ever (identical rows, or a majority centre
            // that attracts everything)
            bool same = true;   // only identical rows repeat exactly; distinct rows get distinct centres next time
            const uint32_t f0 = parent->children[0]->assign[0];
            for (auto fi : parent->children[0]->assign) if (hamming32(train + 32 * (size_t)fi, train + 32 * (size_t)f0) != 0) { same = false; break; }
            if (same) { unbounded = true; --depth_guard; return; }
        }
        for (auto& ch : parent->children)
            if ((int)ch->assign.size() > k) create(ch.get());
        --depth_guard;
    }
};

inline uint32_t pad_to(uint32_t size, uint32_t al) { uint32_t n = size / al; if (size % al) n++; return n * al; }
inline uint64_t km_block_size(uint32_t n) { return pad_to(8 + 8 * n, 8) + (uint64_t)n * 32; }

// block data exactly as KMeansIndex stores it (alignment 8 for binary descriptors)
int km_build_blob(const uint8_t* train, int nt, int k, int max_iters, std::vector<uint8_t>& blob) {
    if (nt <= 0 || k < 1 || max_iters < -1) return -1;
    KmNode root;
    root.assign.resize(nt);
    for (int i = 0; i < nt; i++) root.assign[i] = i;
    KmBuilder b{train, k, max_iters};
    b.create(&root);
    if (b.unbounded) return -2;
    std::deque<KmNode*> queue;
    std::vector<std::pair<KmNode*, uint64_t>> offs;
    uint64_t total = 0;
    queue.push_back(&root);
    while (!queue.empty()) {
        KmNode* nd = queue.front(); queue.pop_front();
        offs.push_back({nd, total});
        uint32_t n;
        if (nd->children.empty()) n = (uint32_t)nd->assign.size();
        else { for (auto& c : nd->children) queue.push_back(c.get()); n = (uint32_t)nd->children.size(); }
        total += km_block_size(n);
    }
    blob.assign(total, 0);
    // children of the i-th visited node are queued in visiting order: their offsets are found by replaying the traversal
    size_t next_child = 1;
    for (size_t i = 0; i < offs.size(); i++) {
        KmNode* nd = offs[i].first;
        uint8_t* blk = blob.data() + offs[i].second;
        const bool leaf = nd->children.empty();
        const uint32_t n = leaf ? (uint32_t)nd->assign.size() : (uint32_t)nd->children.size();
        const uint32_t hs = pad_to(8 + 8 * n, 8);
        const uint16_t n16 = (uint16_t)n;
        std::memcpy(blk, &n16, 2);
        blk[2] = leaf ? 1 : 0;
        std::memcpy(blk + 4, &hs, 4);
        for (uint32_t j = 0; j < n; j++) {
            uint64_t info;
            const uint8_t* feat;
            if (leaf) { info = (uint64_t)nd->assign[j] | 0x8000000000000000ull; feat = train + 32 * (size_t)nd->assign[j]; }
            else { info = offs[next_child++].second; feat = nd->children[j]->feature; }   // centre (train row or majority vector)
            std::memcpy(blk + 8 + 8 * j, &info, 8);
            std::memcpy(blk + hs + 32 * (size_t)j, feat, 32);
        }
    }
    return 0;
}

struct Branch { uint32_t offset; int32_t dist; };

struct BranchHeap {   // impl/heap.h (min-heap on dist, capacity 5000)
    std::vector<Branch> a;
    void sift_to_root(size_t i) {            // heap.h "down"
        while (i != 0) {
            const size_t p = (i - 1) / 2;
            if (a[i].dist < a[p].dist) { std::swap(a[i], a[p]); i = p; } else return;
        }
    }
    void sift_to_leaves(size_t i) {          // heap.h "up"
        for (;;) {
            const size_t l = 2 * i + 1, r = 2 * i + 2, n = a.size();
            if (l >= n) return;
            if (r >= n) { if (a[l].dist < a[i].dist) std::swap(a[i], a[l]); return; }
            if (a[l].dist < a[r].dist) { if (a[l].dist < a[i].dist) { std::swap(a[i], a[l]); i = l; } else return; }
            else { if (a[r].dist < a[i].dist) { std::swap(a[i], a[r]); i = r; } else return; }
        }
    }
    void push(Branch b) {
        if (a.size() >= 5000) return;        // "Heap max size reached": the value is dropped
        a.push_back(b);
        if (a.size() > 1) sift_to_root(a.size() - 1);
    }
    Branch pop() {
        const Branch res = a[0];
        std::swap(a[0], a[a.size() - 1]);
        a.pop_back();
        if (a.size() > 1) sift_to_leaves(0);
        return res;
    }
};

}  // namespace

extern "C" {

// serialised block data of HKMeansParams(k, max_iters) over nt rows; returns its size (call with out = NULL first), -1 bad
// arguments / unsupported max_iters, -2 more than k identical rows (unbounded recursion in the reference)
long oracle_hkmeans_blob(const uint8_t* train, int nt, int k, int max_iters, uint8_t* out, long cap) {
    std::vector<uint8_t> blob;
    const int rc = km_build_blob(train, nt, k, max_iters, blob);
    if (rc) return rc;
    if (out) std::memcpy(out, blob.data(), (size_t)std::min<long>(cap, (long)blob.size()));
    return (long)blob.size();
}

// KMeansIndex::_knnsearch_nn on a block blob (the nn==1&&maxChecks==1 and nn==2&&maxChecks<=2 shortcuts of :216-224 are other
// code paths and not restated); sorted as Index::sort
int oracle_hkmeans_search(const uint8_t* blob, const uint8_t* queries, int nq, int nn, int max_checks, int sorted, int32_t* indices,
                          int32_t* distances) {
    if (nn < 1 || nq < 0 || (nn == 1 && max_checks == 1) || (nn == 2 && max_checks <= 2)) return -1;
    BranchHeap heap;
    for (int q = 0; q < nq; q++) {
        const uint8_t* f = queries + 32 * (size_t)q;
        Row r{distances + (size_t)q * nn, indices + (size_t)q * nn, 0, nn, -1};
        heap.a.clear();
        heap.push({0, 0});
        int nchecks = 0;
        while (nchecks < max_checks && !heap.a.empty()) {
            uint64_t off = heap.pop().offset;
            for (;;) {
                const uint8_t* blk = blob + off;
                uint16_t n; std::memcpy(&n, blk, 2);
                uint32_t hs; std::memcpy(&hs, blk + 4, 4);
                if (blk[2]) break;
                int32_t bestd = std::numeric_limits<int32_t>::max(), besto = -1;
                for (int c = 0; c < n; c++) {
                    const int32_t d = hamming32(f, blk + hs + 32 * (size_t)c);
                    uint64_t info; std::memcpy(&info, blk + 8 + 8 * c, 8);
                    const uint32_t o = (uint32_t)(info & 0x7FFFFFFFFFFFFFFFull);
                    if (d < bestd) {
                        if (besto != -1) heap.push({(uint32_t)besto, bestd});
                        bestd = d; besto = (int32_t)o;
                    } else heap.push({o, d});
                }
                off = (uint64_t)(uint32_t)besto;
            }
            const uint8_t* blk = blob + off;
            uint16_t n; std::memcpy(&n, blk, 2);
            uint32_t hs; std::memcpy(&hs, blk + 4, 4);
            for (int c = 0; c < n; c++) {
                uint64_t info; std::memcpy(&info, blk + 8 + 8 * c, 8);
                r.push(hamming32(f, blk + hs + 32 * (size_t)c), (int)(info & 0x7FFFFFFFFFFFFFFFull));
            }
            nchecks += n;
        }
        for (int i = r.n; i < nn; ++i) { r.idx[i] = -1; r.dist[i] = 0; }   // quiet_NaN() of an integer type is 0
        if (sorted)
            for (int i = 0; i < nn - 1; ++i)
                if (r.idx[i] != -1)
                    for (int j = i + 1; j < nn; ++j)
                        if (r.dist[i] > r.dist[j]) r.swp(i, j);
    }
    return 0;
}

}  // extern "C"
