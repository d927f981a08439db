// flatten_for_ba against hand-derived vertex / edge sets on a toy map (no GPU: the sink is plain vectors).  The expected sets follow
// from globaloptimizer_g2o.cpp:99-172,191-249 by hand, see the comments at each case.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../include/ucoslam_hip/flatten_ba.hpp"
#include "toy_map.hpp"

#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)
using namespace ucoslam_hip;

static ToyMap make_map() {
    ToyMap m;
    m.frames.resize(7);                       // capacity 7: index 6 is a hole
    for (int f = 0; f < 6; f++) { m.frames[f].valid = true; m.frames[f].pose[3] = 0.1f * f; m.frames[f].intr[0] = 500.f + f; }
    m.order = {0, 1, 2, 3, 4, 5};
    m.points.resize(7);
    for (int p = 0; p < 7; p++) { m.points[p].xyz[0] = (float)p; m.points[p].xyz[2] = 5.f; }
    m.sf = {1.f, 1.2f, 1.44f};
    // frame 0 sees P0, (a keypoint without a map point), P5
    m.observe(0, 0, 10, 11, 0);
    m.frames[0].kpts.push_back({1, 1, 0, 0}); m.frames[0].ids.push_back(0xFFFFFFFFu);
    m.observe(5, 0, 50, 51, 1);
    // frame 1 sees P1 (its only observer), P0, P2
    m.observe(1, 1, 20, 21, 0);
    m.observe(0, 1, 12, 13, 1);
    m.observe(2, 1, 30, 31, 2);
    // frame 2 sees P2, P0, P3 (bad)
    m.observe(2, 2, 32, 33, 0);
    m.observe(0, 2, 14, 15, 2);
    m.observe(3, 2, 40, 41, 0);
    m.observe(3, 1, 42, 43, 0);
    m.points[3].bad = true;
    // frame 3 sees P5; frame 4 sees P4; frame 5 sees P2 and P4
    m.observe(5, 3, 52, 53, 0);
    m.observe(4, 4, 60, 61, 1);
    m.observe(2, 5, 34, 35, 1);
    m.observe(4, 5, 62, 63, 0);
    return m;
}

int main() {
    // ---- case 1: a local window.  used = {0,1,2}, fixFirstFrame (front = 0), fixed_frames = {2, 4} (4 is not used: ignored).
    //   visit f=0 (FIXED_WITHPOINTS still contributes): P0 taken (observers 0,1,2 all used), P5 taken (observer 3 joins, FIXED_WITHOUTPOINTS)
    //   visit f=1: P1 has one observer -> rejected; P0 seen; P2 taken (observer 5 joins, FIXED_WITHOUTPOINTS)
    //   visit f=2: P2, P0 seen; P3 bad -> rejected.  f=3, f=5: FIXED_WITHOUTPOINTS, contribute nothing -> P4 is never reached.
    //   frames (ascending index) 0 1 2 3 5, fixed 1 0 1 1 1; points in taking order 0 5 2;
    //   edges per point in observer order: P0: 0 1 2 | P5: 0 3 | P2: 1 2 5  -> 8
    {
        ToyMap m = make_map();
        BAParamSet ps;
        ps.used_frames = {0, 1, 2}; ps.fixed_frames = {2, 4}; ps.nIters = 5;
        VectorSink sink;
        const FlatBAIndex ix = flatten_for_ba(m, ps, sink);
        EXPECT((ix.frame_of == std::vector<uint32_t>{0, 1, 2, 3, 5}));
        EXPECT((ix.frame_fixed == std::vector<uint8_t>{1, 0, 1, 2, 2}));
        EXPECT((sink.fixed == std::vector<uint8_t>{1, 0, 1, 1, 1}));
        EXPECT((ix.point_of == std::vector<uint32_t>{0, 5, 2}));
        EXPECT(ix.n_obs == 8 && sink.obs.size() == 8);
        const int exp_pt[8] = {0, 0, 0, 1, 1, 2, 2, 2}, exp_fr[8] = {0, 1, 2, 0, 3, 1, 2, 4};
        const float exp_u[8] = {10, 12, 14, 50, 52, 30, 32, 34};
        const int exp_oct[8] = {0, 1, 2, 1, 0, 2, 0, 1};
        for (int e = 0; e < 8; e++) {
            EXPECT(sink.obs[e].point == exp_pt[e] && sink.obs[e].frame == exp_fr[e]);
            EXPECT(sink.obs[e].u == exp_u[e] && sink.obs[e].v == exp_u[e] + 1);
            EXPECT(sink.obs[e].inv_sigma == (double)(float)(1. / m.sf[exp_oct[e]]));   // vector<float> _InvScaleFactors, widened
        }
        EXPECT(sink.points[0] == 0.f && sink.points[3] == 5.f && sink.points[6] == 2.f && sink.points[5] == 5.f);
        EXPECT(sink.poses[16 * 3 + 3] == 0.3f && sink.poses[16 * 4 + 3] == 0.5f && sink.intr[4 * 4] == 505.f);
        // getResults: free frame 1 moves, fixed ones do not; bad association named by (map point id, frame index)
        std::vector<float> poses = sink.poses, pts = sink.points;
        for (auto& v : poses) v += 1.f;
        for (auto& v : pts) v += 2.f;
        std::vector<uint8_t> bad(8, 0);
        bad[4] = 1; bad[7] = 1;
        const auto ba = apply_results(m, ix, poses.data(), pts.data(), bad.data(), sink.obs.data());
        EXPECT(m.frames[1].pose[3] == 1.1f && m.frames[0].pose[3] == 0.0f && m.frames[2].pose[3] == 0.2f && m.frames[3].pose[3] == 0.3f);
        EXPECT(m.points[5].xyz[0] == 7.f && m.points[2].xyz[2] == 7.f && m.points[4].xyz[0] == 4.f);
        EXPECT(ba.size() == 2 && ba[0] == std::make_pair(5u, 3u) && ba[1] == std::make_pair(2u, 5u));
        EXPECT(m.points[0].normals_updated == 1 && m.points[5].normals_updated == 1 && m.points[2].normals_updated == 1 && m.points[4].normals_updated == 0);
    }
    // ---- case 2: used_frames empty = every keyframe; only the first one is fixed.  f=4 now contributes P4 (observers 4, 5 both used).
    //   points 0 5 2 4; edges 3 + 2 + 3 + 2 = 10; nobody joins as FIXED_WITHOUTPOINTS
    {
        ToyMap m = make_map();
        BAParamSet ps;
        VectorSink sink;
        const FlatBAIndex ix = flatten_for_ba(m, ps, sink);
        EXPECT((ix.frame_of == std::vector<uint32_t>{0, 1, 2, 3, 4, 5}));
        EXPECT((sink.fixed == std::vector<uint8_t>{1, 0, 0, 0, 0, 0}));
        EXPECT((ix.point_of == std::vector<uint32_t>{0, 5, 2, 4}) && ix.n_obs == 10);
        EXPECT(sink.obs[8].point == 3 && sink.obs[8].frame == 4 && sink.obs[9].frame == 5);
    }
    // ---- case 3: fixFirstFrame off, nothing fixed; the container's first keyframe is not the lowest index
    {
        ToyMap m = make_map();
        m.order = {3, 0, 1, 2, 4, 5};
        BAParamSet ps;
        ps.fixFirstFrame = false;
        VectorSink sink;
        flatten_for_ba(m, ps, sink);
        EXPECT((sink.fixed == std::vector<uint8_t>{0, 0, 0, 0, 0, 0}));
        ps.fixFirstFrame = true;
        flatten_for_ba(m, ps, sink);
        EXPECT((sink.fixed == std::vector<uint8_t>{0, 0, 0, 1, 0, 0}));   // front() is frame 3
        ps.used_frames = {0, 1};                                            // front() not used: nothing is fixed by fixFirstFrame
        const FlatBAIndex ix = flatten_for_ba(m, ps, sink);
        EXPECT((ix.frame_fixed == std::vector<uint8_t>{0, 0, 2, 2, 2}));  // 0 1 free; 2, 3, 5 join without points
    }
    // ---- case 4: what the monocular, marker-less path refuses
    {
        ToyMap m = make_map();
        m.frames[1].kpts[m.points[0].frames[1]].depth = 2.5f;   // a stereo observation of P0
        BAParamSet ps;
        VectorSink sink;
        bool threw = false;
        try { flatten_for_ba(m, ps, sink); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
        ToyMap m2 = make_map();
        m2.frames[2].valid_markers = true;
        threw = false;
        try { flatten_for_ba(m2, ps, sink); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
        ToyMap m3 = make_map();
        ps.used_frames = {6};   // a hole of the container
        threw = false;
        try { flatten_for_ba(m3, ps, sink); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
    }
    std::printf("flatten ok\n");
    return 0;
}
