// uh_dev_frame: what FrameExtractor::process leaves behind for the tracker, resident in HBM — the frame's descriptors, its undistorted
// keypoints in kd-leaf order and picoflann's kd-tree over them (src/utils/frameextractor.cpp:430-520, :3985 undistortPoints, :4258
// create_kdtree).  Written by the extractor's last launches (orb.hip) and the build launch (kdbuild.hip), adopted by the projection
// matcher without a host round trip (projmatch.hip: uh_projmatch_set_frame_dev).
#pragma once
#include "common.hpp"
#include "kdbuild.hpp"

struct uh_dev_frame {
    uh_ctx* ctx = nullptr;
    uh::DevBuf buf;          // [descriptors n_cap x 32 | build input n_cap x 16 {und x, und y, bits(octave), -} | nodes | leaf records n_cap x 16 | scratch of the split build]
    uh::MappedBuf meta;      // uh_kd::Meta, written by the build launch; its word is the completion word the host polls
    int n_cap = 0;
    int threads = 512;       // of the build workgroup (UH_KD_THREADS: 256 / 512; 512 lanes leave each 256 registers: the cached row state of kdbuild.hpp)
    bool host_tree = false;  // uh_dev_frame_set_tree_builder(f, 1): no build launch — uh_projmatch_set_frame_dev builds the tree on the calling core from the host copy of the
                             // undistorted keypoints and uploads nodes + leaf records into this object; the descriptors still never leave the device
    bool split = true;       // the build over three launches (kdbuild.hpp); false: one launch of `threads` (UH_KD_SPLIT=0, UH_KD_THREADS, the test hook's explicit sizes)
    size_t o_desc = 0, o_in = 0, o_nodes = 0, o_leaf = 0;
    size_t o_pts = 0, o_dump = 0, o_sums = 0, o_sub = 0;   // scratch of the split build: points after the top levels, top nodes, subtree summaries and records
    int sub_stride = 0;      // records per subtree region
    unsigned long long seq = 0;   // word of the latest build launch (0: none yet)
    bool attr_set = false;
    uh::DevBuf d_clk;        // UH_KD_CLK=1: phase stamps of the build launch (measurement only)

    uint8_t* desc() const { return buf.as<uint8_t>() + o_desc; }
    float4* kd_in() const { return reinterpret_cast<float4*>(buf.as<uint8_t>() + o_in); }
    uh_kd::Node24* nodes() const { return reinterpret_cast<uh_kd::Node24*>(buf.as<uint8_t>() + o_nodes); }
    float4* leaf() const { return reinterpret_cast<float4*>(buf.as<uint8_t>() + o_leaf); }
    float4* pts() const { return reinterpret_cast<float4*>(buf.as<uint8_t>() + o_pts); }
    uh_kd::TopDump* dump() const { return reinterpret_cast<uh_kd::TopDump*>(buf.as<uint8_t>() + o_dump); }
    uh_kd::SubSum* sums() const { return reinterpret_cast<uh_kd::SubSum*>(buf.as<uint8_t>() + o_sums); }
    uh_kd::Node24* sub_nodes() const { return reinterpret_cast<uh_kd::Node24*>(buf.as<uint8_t>() + o_sub); }
};

namespace uh {
// room for up to n_cap keypoints (<= uh_kd::kDevMaxPoints)
int dev_frame_reserve(uh_dev_frame* f, int n_cap);
// enqueue the build on the frame's context stream: n = min(sum of d_level_counts[0 .. nlevels), cap) when d_level_counts is given, else n_direct
int kd_build_launch(uh_dev_frame* f, const int* d_level_counts, int nlevels, int cap, int n_direct);
// wait for the latest build (normally long since complete) and return its meta block
int dev_frame_wait(uh_dev_frame* f, const uh_kd::Meta** meta, const char* what);
}  // namespace uh
