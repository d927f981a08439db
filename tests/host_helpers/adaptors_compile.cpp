// Compiles the C++ adaptors (no OpenCV, no GPU needed) and exercises the CPU-visible error paths of the C ABI.
#include <cstdio>
#include "../../include/ucoslam_hip/adaptors.hpp"
int main() {
    std::printf("version %d\n", uh_version());
    try {
        auto ctx = std::make_shared<ucoslam_hip::Context>(0);
        // a GPU is present: run one tiny search through the adaptor
        ucoslam_hip::Index idx(ctx);
        std::vector<uint8_t> t(64 * 32, 7), q(4 * 32, 7);
        std::vector<int32_t> I(4 * 2), D(4 * 2);
        if (idx.search({q.data(), 4, 32, 1}, 2, {I.data(), 4, 2, 4}, {D.data(), 4, 2, 4})) return 3;   // unbuilt -> false
        idx.build({t.data(), 64, 32, 1});
        if (!idx.search({q.data(), 4, 32, 1}, 2, {I.data(), 4, 2, 4}, {D.data(), 4, 2, 4}, true)) return 4;
        std::printf("gpu path ok: first neighbour %d dist %d\n", I[0], D[0]);
        // projection matcher through the adaptor: one map point in front of an identity camera, on top of keypoint 1
        ucoslam_hip::ProjectionMatcher pm(ctx);
        uh_keypoint kps[3] = {{100, 100, 31, 0, 50, 0, -1}, {320, 240, 31, 0, 50, 0, -1}, {500, 300, 31, 0, 50, 1, -1}};
        std::vector<uint8_t> kd(3 * 32, 0), md(32, 0);
        kd[32] = 1;   // keypoint 1 differs from the map point by one bit
        const float sf[2] = {1.f, 1.2f};
        uh_proj_frame fr{kps, 3, kd.data(), sf, 2, 500.f, 500.f, 320.f, 240.f, 0, 0, 640, 480};
        pm.setFrame(fr);
        const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const uint32_t id = 77;
        const float pos[3] = {0, 0, 5}, nrm[3] = {0, 0, -1}, mind = 1.f, maxd = 5.5f;
        uh_map_points mp{1, &id, pos, nrm, &mind, &maxd, md.data()};
        std::vector<uint8_t> vis;
        auto mm = pm.matchFrameToMapPoints(T, mp, 100.f, 15.f, &vis);
        if (mm.size() != 1 || mm[0].queryIdx != 1 || mm[0].trainIdx != 77 || mm[0].distance != 1.f || vis[0] != 1) return 5;
        std::printf("projection matcher ok: kp %d <- map point %d (d=%g)\n", mm[0].queryIdx, mm[0].trainIdx, mm[0].distance);
        ucoslam_hip::PnPSolver pnp(ctx);   // construction only; the solver is covered by tests/test_pnp.py
    } catch (const std::runtime_error& e) {
        std::printf("no device: %s\n", e.what());   // expected on the CPU-only build box: no fallback exists
    }
    return 0;
}
