// TEST INFRASTRUCTURE ONLY — thin C driver around the REAL reference kd-tree (src/basictypes/picoflann.h, header-only),
// compiled where it lies under /root/reference into oracle/_ref/libpicoflann_ref.so (never copied into this repo).
// It instantiates the same template the reference's Frame uses (map_types/frame.h:50-62: KdTreeIndex<2, adapter returning
// pt.x / pt.y as float>) on a plain (x,y) float container, and exposes build + radiusSearch(sorted=false) — the call made by
// Frame::getKeyPointsInRegion (map_types/frame.cpp:102-115) — so that oracle/proj_oracle.cpp's restatement of the tree build
// and of the traversal ORDER can be pinned against the real thing (tests/test_projmatch_oracle.py).
#include <cstdint>
#include <vector>

#include "basictypes/picoflann.h"

namespace {
struct P2 { float x, y; };
struct Adapter {
    inline float operator()(const P2& p, int dim) const { return dim == 0 ? p.x : p.y; }
};
struct Tree {
    std::vector<P2> pts;
    picoflann::KdTreeIndex<2, Adapter> kd;
};
}  // namespace

extern "C" {

void* picoflann_ref_build(const float* xy, int n) {
    Tree* t = new Tree();
    t->pts.resize(n);
    for (int i = 0; i < n; i++) { t->pts[i].x = xy[2 * i]; t->pts[i].y = xy[2 * i + 1]; }
    t->kd.build(t->pts);
    return t;
}

void picoflann_ref_free(void* h) { delete static_cast<Tree*>(h); }

// radiusSearch(container, query, radius, sorted=false): returns the number of hits, writes up to cap (index, squared dist)
int picoflann_ref_radius(void* h, float qx, float qy, double radius, uint32_t* idx_out, double* sqd_out, int cap) {
    Tree* t = static_cast<Tree*>(h);
    P2 q{qx, qy};
    auto res = t->kd.radiusSearch(t->pts, q, radius, false);
    int n = 0;
    for (auto& r : res) {
        if (n < cap) { idx_out[n] = r.first; sqd_out[n] = r.second; }
        n++;
    }
    return n;
}

}  // extern "C"
