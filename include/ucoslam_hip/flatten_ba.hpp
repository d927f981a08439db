// flatten_for_ba: GlobalOptimizerG2O::setParams' graph-selection rules (src/optimization/globaloptimizer_g2o.cpp:77-249) as code.
//
// The reference walks its Map and builds g2o vertices and edges; the HIP optimiser takes the same selection as flat arrays
// (uh_ba_staging, include/ucoslam_hip.h).  This header is that walk, written against a minimal map VIEW (any type with the members
// listed under "MapView" below — the reference's Map satisfies it through the ten-line wrapper shown in INTEGRATION.md §4; the tests
// use a toy map), writing straight into the optimiser's pinned staging block: no intermediate copy, no OpenCV type in sight.
//
// Rules reproduced, with the lines they come from:
//   :108-116  used frames = ParamSet::used_frames, or every keyframe when that set is empty
//   :119-128  fixFirstFrame -> the map's first keyframe is FIXED_WITHPOINTS if it is used; fixed_frames likewise — ONLY if used
//             (the comment there says "add also the fixed ones", the code never adds one)
//   :133-155  frames visited in ascending INDEX order (not in used_frames order); a FIXED_WITHOUTPOINTS frame contributes no points;
//             a point is taken on first sight unless it has fewer than two observers (and is not stereo) or is bad — in which case it
//             is remembered as visited and never looked at again; every observer of a taken point that is not yet used joins as
//             FIXED_WITHOUTPOINTS (and, being visited later in the same loop only if its index is larger, contributes no points)
//   :191-202  frame vertices: every used frame, ascending index; fixed iff FIXED_WITHPOINTS or FIXED_WITHOUTPOINTS
//   :207-219  point vertices in the order the points were taken
//   :224-249  one monocular edge per (taken point, observer that is used) in the point's observer order (ascending frame index:
//             MapPoint::frames is a std::map): measurement = undistorted keypoint, intrinsics of THAT frame, information =
//             I * _InvScaleFactors[octave] with _InvScaleFactors a vector<float> filled with 1./scaleFactor (:96-97) — i.e. the
//             optimiser sees (double)(float)(1. / f).  depth > 0 (stereo / RGB-D observation, :250-) is refused: monocular path.
//   markers (:158-171, :300-398) are refused: marker-less maps only.
// getResults (:466-537) is apply_results below: poses of the non-fixed used frames, coordinates of the taken points, and the bad
// associations as (map point id, frame id) pairs.
#pragma once
#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../ucoslam_hip.h"

namespace ucoslam_hip {

// GlobalOptimizer::ParamSet (globaloptimizer.h:31-47): the members the marker-less monocular path reads
struct BAParamSet {
    std::unordered_set<uint32_t> used_frames;   // which are used; if empty, all
    std::set<uint32_t> fixed_frames;            // which are set as not movable
    bool fixFirstFrame = true;
    int nIters = 100;
    bool verbose = false;
};

/* MapView — what flatten_for_ba needs from a map (all const):
 *   uint32_t frame_capacity()                       map->keyframes.capacity(): frame indices are < this
 *   bool     frame_valid(uint32_t f)                map->keyframes.is(f)
 *   template <class F> void for_each_keyframe(F)    F(uint32_t idx) for every keyframe, in the container's order
 *   uint32_t front_keyframe()                       map->keyframes.front().idx
 *   const float* frame_pose_f2g(f)                  16 floats, row-major 4x4 (Frame::pose_f2g)
 *   void     frame_intrinsics(f, float out[4])      fx fy cx cy (Frame::imageParams)
 *   size_t   frame_n_ids(f); uint32_t frame_id(f,i) Frame::ids (0xFFFFFFFF = no map point)
 *   void     frame_keypoint(f, kp, float& x, float& y, int& octave)   Frame::und_kpts[kp]
 *   float    frame_depth(f, kp)                     Frame::getDepth(kp): <= 0 for a monocular observation
 *   bool     frame_has_valid_markers(f)             any marker of the frame with a valid pose in the map (:158-171)
 *   const std::vector<float>& scale_factors()       map->keyframes.front().scaleFactors
 *   uint32_t point_capacity()
 *   bool     point_bad(p), point_stereo(p)
 *   size_t   point_n_observers(p)
 *   template <class F> void for_each_observer(p, F) F(uint32_t frame, uint32_t kp) in ascending frame index (MapPoint::frames)
 *   void     point_coordinates(p, float out[3])
 * and, for apply_results, non-const:
 *   void set_frame_pose_f2g(f, const float m[16]); void set_point_coordinates(p, const float xyz[3]); void update_point_normal_and_distances(p)
 */

struct FlatBAIndex {                       // what getResults needs to write back
    std::vector<uint32_t> frame_of;        // flattened frame -> map frame index
    std::vector<uint8_t>  frame_fixed;     // 0 free, 1 FIXED_WITHPOINTS, 2 FIXED_WITHOUTPOINTS
    std::vector<uint32_t> point_of;        // flattened point -> map point id (= usedMapPoints)
    int n_obs = 0;
};

constexpr uint32_t kInvalidIdx = 0xFFFFFFFFu;

// Sink = something with `uh_ba_staging begin(int n_frames, int n_points, int n_obs)`: StagingSink(uh_ba*) below maps the optimiser's
// pinned block; tests pass a vector-backed one and run without a GPU.
template <class MapView, class Sink>
FlatBAIndex flatten_for_ba(const MapView& map, const BAParamSet& ps, Sink& sink) {
    enum : uint8_t { UNFIXED = 0, FIXED_WITHPOINTS = 1, FIXED_WITHOUTPOINTS = 2 };
    const uint32_t FC = map.frame_capacity(), PC = map.point_capacity();
    std::vector<uint8_t> used(FC, 0), fixed(FC, UNFIXED);
    std::vector<uint8_t> pstate(PC, 0);   // 0 unseen, 1 taken, 2 visited and rejected
    auto use = [&](uint32_t f) {
        if (f >= FC || !map.frame_valid(f)) throw std::runtime_error("flatten_for_ba: frame " + std::to_string(f) + " is not a keyframe of the map");
        used[f] = 1;
    };
    if (ps.used_frames.empty()) map.for_each_keyframe([&](uint32_t f) { use(f); });
    else for (uint32_t f : ps.used_frames) use(f);
    if (ps.fixFirstFrame) { const uint32_t f0 = map.front_keyframe(); if (f0 < FC && used[f0]) fixed[f0] = FIXED_WITHPOINTS; }
    for (uint32_t f : ps.fixed_frames) if (f < FC && used[f]) fixed[f] = FIXED_WITHPOINTS;

    FlatBAIndex ix;
    for (uint32_t f = 0; f < FC; f++) {   // ascending index: frames that join below with a larger index are reached, and skipped
        if (!used[f] || fixed[f] == FIXED_WITHOUTPOINTS) continue;
        const size_t nids = map.frame_n_ids(f);
        for (size_t i = 0; i < nids; i++) {
            const uint32_t p = map.frame_id(f, i);
            if (p == kInvalidIdx) continue;
            if (p >= PC) throw std::runtime_error("flatten_for_ba: frame " + std::to_string(f) + " references map point " + std::to_string(p) + " beyond the map");
            if (pstate[p]) continue;
            if ((map.point_n_observers(p) < 2 && !map.point_stereo(p)) || map.point_bad(p)) { pstate[p] = 2; continue; }
            pstate[p] = 1;
            ix.point_of.push_back(p);
            map.for_each_observer(p, [&](uint32_t of, uint32_t) {
                if (of >= FC || !map.frame_valid(of)) throw std::runtime_error("flatten_for_ba: map point " + std::to_string(p) + " is observed by a frame that is not in the map");
                if (!used[of]) { used[of] = 1; fixed[of] = FIXED_WITHOUTPOINTS; }
            });
        }
        if (map.frame_has_valid_markers(f))
            throw std::runtime_error("flatten_for_ba: frame " + std::to_string(f) + " sees markers with a valid pose; the HIP optimiser takes marker-less maps only");
    }
    std::vector<int32_t> flat_of(FC, -1);
    for (uint32_t f = 0; f < FC; f++)
        if (used[f]) { flat_of[f] = (int32_t)ix.frame_of.size(); ix.frame_of.push_back(f); ix.frame_fixed.push_back(fixed[f]); }
    int E = 0;
    for (uint32_t p : ix.point_of) map.for_each_observer(p, [&](uint32_t of, uint32_t) { E += used[of] ? 1 : 0; });
    ix.n_obs = E;

    const int K = (int)ix.frame_of.size(), P = (int)ix.point_of.size();
    if (K == 0) throw std::runtime_error("flatten_for_ba: no frame selected");
    uh_ba_staging st = sink.begin(K, P, E);
    for (int k = 0; k < K; k++) {
        const uint32_t f = ix.frame_of[k];
        const float* M = map.frame_pose_f2g(f);
        for (int j = 0; j < 16; j++) st.poses_f2g[16 * k + j] = M[j];
        st.fixed[k] = ix.frame_fixed[k] ? 1 : 0;
        map.frame_intrinsics(f, st.intr + 4 * k);
    }
    // _InvScaleFactors (:96-97): a vector<float> of 1./f, read back into a double information matrix
    const std::vector<float>& sf = map.scale_factors();
    std::vector<float> inv_sf(sf.size());
    for (size_t i = 0; i < sf.size(); i++) inv_sf[i] = (float)(1. / sf[i]);
    int e = 0;
    for (int pi = 0; pi < P; pi++) {
        const uint32_t p = ix.point_of[pi];
        map.point_coordinates(p, st.points + 3 * pi);
        map.for_each_observer(p, [&](uint32_t of, uint32_t kp) {
            if (!used[of]) return;
            if (map.frame_depth(of, kp) > 0)
                throw std::runtime_error("flatten_for_ba: stereo / RGB-D observation (map point " + std::to_string(p) + ", frame " + std::to_string(of) + "); monocular edges only");
            float x, y;
            int octave;
            map.frame_keypoint(of, kp, x, y, octave);
            if (octave < 0 || (size_t)octave >= inv_sf.size()) throw std::runtime_error("flatten_for_ba: keypoint octave outside the scale-factor table");
            uh_ba_obs& o = st.obs[e++];
            o.point = pi; o.frame = flat_of[of]; o.u = x; o.v = y; o.inv_sigma = (double)inv_sf[octave];
        });
    }
    return ix;
}

// getResults (:466-537) onto the map: poses of the free used frames, coordinates of every taken point, then
// updatePointNormalAndDistances; returns the bad associations as (map point id, frame index)
template <class MapView>
std::vector<std::pair<uint32_t, uint32_t>> apply_results(MapView& map, const FlatBAIndex& ix, const float* poses /*K x 16*/, const float* points /*P x 3*/,
                                                         const uint8_t* bad /*E*/, const uh_ba_obs* obs /*E*/) {
    for (size_t k = 0; k < ix.frame_of.size(); k++)
        if (!ix.frame_fixed[k]) map.set_frame_pose_f2g(ix.frame_of[k], poses + 16 * k);
    for (size_t p = 0; p < ix.point_of.size(); p++) map.set_point_coordinates(ix.point_of[p], points + 3 * p);
    std::vector<std::pair<uint32_t, uint32_t>> out;
    for (int e = 0; e < ix.n_obs; e++)
        if (bad[e]) out.push_back({ix.point_of[obs[e].point], ix.frame_of[obs[e].frame]});
    for (uint32_t p : ix.point_of) map.update_point_normal_and_distances(p);
    return out;
}

// The optimiser's own pinned staging block as the sink: the walk above writes where the H2D copy reads
struct StagingSink {
    uh_ba* ba;
    uh_ba_staging st{};
    explicit StagingSink(uh_ba* b) : ba(b) {}
    uh_ba_staging begin(int K, int P, int E) {
        if (uh_ba_map_staging(ba, K, P, E, &st) < 0) throw std::runtime_error(uh_last_error());
        return st;
    }
};

// Plain vectors as the sink (tests; hosts that want the arrays for something else)
struct VectorSink {
    std::vector<float> poses, intr, points;
    std::vector<uint8_t> fixed;
    std::vector<uh_ba_obs> obs;
    uh_ba_staging begin(int K, int P, int E) {
        poses.assign(16 * (size_t)K, 0.f); intr.assign(4 * (size_t)K, 0.f); points.assign(3 * (size_t)P, 0.f); fixed.assign(K, 0); obs.assign(E, uh_ba_obs{});
        return uh_ba_staging{poses.data(), fixed.data(), intr.data(), points.data(), obs.data(), K, P, E};
    }
};

}  // namespace ucoslam_hip
