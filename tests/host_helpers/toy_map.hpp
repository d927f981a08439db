// A toy map for the tests of include/ucoslam_hip/flatten_ba.hpp: the members of the MapView concept over plain vectors, shaped
// like the reference's containers (keyframes indexed by idx with holes, MapPoint::frames as a std::map<frame, keypoint index>).
#pragma once
#include <cstdint>
#include <map>
#include <vector>

struct ToyKeypoint { float x, y; int octave; float depth; };
struct ToyFrame {
    bool valid = false;
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float intr[4] = {500, 500, 320, 240};
    std::vector<uint32_t> ids;            // Frame::ids
    std::vector<ToyKeypoint> kpts;        // Frame::und_kpts (+ depth)
    bool valid_markers = false;
};
struct ToyPoint {
    bool bad = false, stereo = false;
    float xyz[3] = {0, 0, 1};
    std::map<uint32_t, uint32_t> frames;  // MapPoint::frames
    int normals_updated = 0;
};
struct ToyMap {
    std::vector<ToyFrame> frames;
    std::vector<uint32_t> order;          // the container's iteration order (front() = order[0])
    std::vector<ToyPoint> points;
    std::vector<float> sf;
    uint32_t frame_capacity() const { return (uint32_t)frames.size(); }
    bool frame_valid(uint32_t f) const { return f < frames.size() && frames[f].valid; }
    template <class F> void for_each_keyframe(F fn) const { for (uint32_t f : order) fn(f); }
    uint32_t front_keyframe() const { return order.front(); }
    const float* frame_pose_f2g(uint32_t f) const { return frames[f].pose; }
    void frame_intrinsics(uint32_t f, float* out) const { for (int i = 0; i < 4; i++) out[i] = frames[f].intr[i]; }
    size_t frame_n_ids(uint32_t f) const { return frames[f].ids.size(); }
    uint32_t frame_id(uint32_t f, size_t i) const { return frames[f].ids[i]; }
    void frame_keypoint(uint32_t f, uint32_t kp, float& x, float& y, int& octave) const { const ToyKeypoint& k = frames[f].kpts[kp]; x = k.x; y = k.y; octave = k.octave; }
    float frame_depth(uint32_t f, uint32_t kp) const { return frames[f].kpts[kp].depth; }
    bool frame_has_valid_markers(uint32_t f) const { return frames[f].valid_markers; }
    const std::vector<float>& scale_factors() const { return sf; }
    uint32_t point_capacity() const { return (uint32_t)points.size(); }
    bool point_bad(uint32_t p) const { return points[p].bad; }
    bool point_stereo(uint32_t p) const { return points[p].stereo; }
    size_t point_n_observers(uint32_t p) const { return points[p].frames.size(); }
    template <class F> void for_each_observer(uint32_t p, F fn) const { for (const auto& fi : points[p].frames) fn(fi.first, fi.second); }
    void point_coordinates(uint32_t p, float* out) const { for (int i = 0; i < 3; i++) out[i] = points[p].xyz[i]; }
    void set_frame_pose_f2g(uint32_t f, const float* m) { for (int i = 0; i < 16; i++) frames[f].pose[i] = m[i]; }
    void set_point_coordinates(uint32_t p, const float* xyz) { for (int i = 0; i < 3; i++) points[p].xyz[i] = xyz[i]; }
    void update_point_normal_and_distances(uint32_t p) { points[p].normals_updated++; }
    // helper: point p observed by frame f at a new keypoint
    void observe(uint32_t p, uint32_t f, float x, float y, int octave, float depth = 0.f) {
        frames[f].kpts.push_back({x, y, octave, depth});
        frames[f].ids.push_back(p);
        points[p].frames[f] = (uint32_t)frames[f].kpts.size() - 1;
    }
};
