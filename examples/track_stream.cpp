// A C++ host over the C ABI, shaped like the reference's own threads (north_star: "host code in C++ calling hand-written HIP kernels
// through a thin C-ABI"): the loop UcoSlam::process() + MapManager run per keyframe interval, with the reference's plugin calls replaced
// by ucoslam_hip's — no Python, no torch, no device pointer in sight of the host program.
//
//   tracker thread (this one)                          mapper thread (the optimiser's worker, uh_ba_solve_async)
//   ------------------------------------------------   --------------------------------------------------------
//   4 frames (pinned host memory) -> detectAndCompute   GlobalOptimizer::setParams(fresh local-BA problem)   mapmanager.cpp:11388-11405
//       (uh_orb_extract_batch: host in, host out)       GlobalOptimizer::optimize()
//   4 x 2000 descriptors -> xflann::Index::search
//       against the 10 000-row map (uh_knn_search: host in, host out; nn = 10, unsorted)
//   uh_ba_wait; GlobalOptimizer::getResults              <- tracker thread, mapmanager.cpp:1267-1305
//
// Same workload as bench.py's step (1241 x 376 frames, 2000 features, 8 levels, 10 x 3000 local BA, nIters 5 + 10); inputs are
// generated here (synthetic scene of rectangles + noise, random map descriptors, a synthetic 10-keyframe window), so the program needs
// nothing but libucoslam_hip.so.  Prints one JSON line: median / min / max ms per step over `reps` repetitions of `steps` steps.
//
//   g++ -std=c++17 -O2 -o track_stream examples/track_stream.cpp -Lucoslam-cv3_amd -lucoslam_hip -Wl,-rpath,$PWD/ucoslam-cv3_amd -Wl,-rpath,/opt/rocm/lib -lpthread
//   ./track_stream [steps=20] [reps=15]
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/ucoslam_hip.h"

#define CHECK(call) do { const int rc_ = (call); if (rc_ < 0) { std::printf("error: %s -> %s\n", #call, uh_last_error()); return 2; } } while (0)

namespace {

constexpr int W = 1241, H = 376, F = 4, NFEAT = 2000, NN = 10, NT = 10000, BA_K = 10, BA_P = 3000;

void make_frame(uint8_t* out, int shift_x, int shift_y, unsigned seed) {
    std::mt19937 scene(1234);   // the scene is fixed, `seed` drives the noise
    const int bw = W + 256, bh = H + 256;
    std::vector<float> img((size_t)bw * bh);
    for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) img[(size_t)y * bw + x] = 110.f + 40.f * std::sin(x / 211.f) + 30.f * std::cos(y / 97.f);
    for (int i = 0; i < 3000; i++) {
        const int cx = scene() % bw, cy = scene() % bh, sx = 3 + scene() % 25, sy = 3 + scene() % 25;
        const float c = (30.f + scene() % 91) * ((scene() & 1) ? 1.f : -1.f);
        for (int y = std::max(cy - sy, 0); y < std::min(cy + sy, bh); y++) for (int x = std::max(cx - sx, 0); x < std::min(cx + sx, bw); x++) img[(size_t)y * bw + x] += c;
    }
    std::mt19937 noise(seed);
    std::normal_distribution<float> nd(0.f, 3.f);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        const float v = img[(size_t)(y + 128 + shift_y) * bw + x + 128 + shift_x] + nd(noise);
        out[(size_t)y * W + x] = (uint8_t)std::min(255.f, std::max(0.f, std::nearbyint(v)));
    }
}

struct Problem {
    std::vector<float> poses, intr, points, uv;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> opt, okf;
    std::vector<double> w;
    uh_ba_problem view() const { return uh_ba_problem{(int32_t)fixed.size(), (int32_t)(points.size() / 3), (int32_t)opt.size(), poses.data(), fixed.data(), intr.data(), points.data(),
                                                      opt.data(), okf.data(), uv.data(), w.data()}; }
};

// a 10-keyframe window on a line (baseline 0.3 m), KITTI intrinsics, landmarks in front, ~87 % visibility, pixel noise 0.5, 2 % outliers,
// octave uniform in 0..7 -> information (double)(float)(1 / 1.2^octave), the first two keyframes fixed
Problem make_problem(unsigned seed) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<double> U(0, 1);
    std::normal_distribution<double> N(0, 1);
    const double fx = 718.856, fy = 718.856, cx = 607.19, cy = 185.22;
    float sf[8]; sf[0] = 1.f; for (int i = 1; i < 8; i++) sf[i] = sf[i - 1] * 1.2f;
    std::vector<std::array<double, 12>> T(BA_K);   // R | t, world -> camera
    for (int k = 0; k < BA_K; k++) {
        const double a = 0.02 * std::sin(0.7 * k), c = std::cos(a), s = std::sin(a);   // a little yaw
        T[k] = {c, 0, s, 0, 1, 0, -s, 0, c, 0, 0, 0};
        const double t[3] = {-0.3 * k, -0.02 * std::sin((double)k), 0};
        for (int r = 0; r < 3; r++) T[k][9 + r] = T[k][3 * r] * t[0] + T[k][3 * r + 1] * t[1] + T[k][3 * r + 2] * t[2];
    }
    Problem pr;
    std::vector<std::array<double, 3>> X;
    std::vector<std::vector<std::array<double, 4>>> obs;   // per point: (kf, u, v, w)
    for (int p = 0; p < BA_P; p++) {
        const double z = 4 + 36 * U(g);
        const std::array<double, 3> x = {(U(g) * W - cx) / fx * z + 0.3 * BA_K / 2, (U(g) * H - cy) / fy * z, z};
        std::vector<std::array<double, 4>> o;
        for (int k = 0; k < BA_K; k++) {
            const auto& Tk = T[k];
            const double pc[3] = {Tk[0] * x[0] + Tk[1] * x[1] + Tk[2] * x[2] + Tk[9], Tk[3] * x[0] + Tk[4] * x[1] + Tk[5] * x[2] + Tk[10], Tk[6] * x[0] + Tk[7] * x[1] + Tk[8] * x[2] + Tk[11]};
            if (pc[2] <= 0.5) continue;
            const double u = fx * pc[0] / pc[2] + cx, v = fy * pc[1] / pc[2] + cy;
            if (u < 0 || u >= W || v < 0 || v >= H || U(g) >= 0.9) continue;
            const int octave = (int)(g() % 8);
            double nu = 0.5 * N(g), nv = 0.5 * N(g);
            if (U(g) < 0.02) { nu += 25 * N(g); nv += 25 * N(g); }
            o.push_back({(double)k, u + nu, v + nv, (double)(float)(1. / sf[octave])});
        }
        if (o.size() >= 2) { X.push_back({x[0] + 0.05 * N(g), x[1] + 0.05 * N(g), x[2] + 0.05 * N(g)}); obs.push_back(o); }
    }
    pr.fixed.assign(BA_K, 0); pr.fixed[0] = pr.fixed[1] = 1;
    for (int k = 0; k < BA_K; k++) {
        std::array<double, 12> Tk = T[k];
        if (!pr.fixed[k]) {   // a small left perturbation exp(d) T: first-order rotation is enough for 0.01 rad
            const double d[6] = {0.01 * N(g), 0.01 * N(g), 0.01 * N(g), 0.01 * N(g), 0.01 * N(g), 0.01 * N(g)};
            const double Rd[9] = {1, -d[2], d[1], d[2], 1, -d[0], -d[1], d[0], 1};
            std::array<double, 12> n{};
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) n[3 * r + c] = Rd[3 * r] * Tk[c] + Rd[3 * r + 1] * Tk[3 + c] + Rd[3 * r + 2] * Tk[6 + c];
                n[9 + r] = Rd[3 * r] * Tk[9] + Rd[3 * r + 1] * Tk[10] + Rd[3 * r + 2] * Tk[11] + d[3 + r];
            }
            Tk = n;
        }
        const float M[16] = {(float)Tk[0], (float)Tk[1], (float)Tk[2], (float)Tk[9], (float)Tk[3], (float)Tk[4], (float)Tk[5], (float)Tk[10],
                             (float)Tk[6], (float)Tk[7], (float)Tk[8], (float)Tk[11], 0, 0, 0, 1};
        pr.poses.insert(pr.poses.end(), M, M + 16);
        const float in[4] = {(float)fx, (float)fy, (float)cx, (float)cy};
        pr.intr.insert(pr.intr.end(), in, in + 4);
    }
    for (size_t p = 0; p < X.size(); p++) {
        for (int i = 0; i < 3; i++) pr.points.push_back((float)X[p][i]);
        for (const auto& o : obs[p]) { pr.opt.push_back((int32_t)p); pr.okf.push_back((int32_t)o[0]); pr.uv.push_back((float)o[1]); pr.uv.push_back((float)o[2]); pr.w.push_back(o[3]); }
    }
    return pr;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace



int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 20, reps = argc > 2 ? std::atoi(argv[2]) : 15;
    uh_ctx *trk = nullptr, *map = nullptr;
    if (uh_ctx_create_private(0, &trk) < 0) { std::printf("no device: %s (there is no CPU path)\n", uh_last_error()); return 0; }
    CHECK(uh_ctx_create_private(0, &map));   // the mapper's own stream: its kernel runs beside the tracker's
    // ---- inputs
    uint8_t* frames = static_cast<uint8_t*>(uh_host_alloc((size_t)F * W * H));
    for (int f = 0; f < F; f++) make_frame(frames + (size_t)f * W * H, 2 * f, f, 1000 + f);
    std::vector<uint8_t> map_desc((size_t)NT * 32);
    { std::mt19937 g(50); for (auto& b : map_desc) b = (uint8_t)g(); }
    std::vector<Problem> problems;
    for (unsigned i = 0; i < 4; i++) problems.push_back(make_problem(i));
    // ---- the plugins
    uh_orb* ext = nullptr; uh_knn* index = nullptr; uh_ba* ba = nullptr;
    CHECK(uh_orb_create(trk, &ext));
    const uh_feat_params fp{2, NFEAT, 8, 1.2f, 0.f};
    CHECK(uh_orb_set_params(ext, &fp));
    CHECK(uh_knn_create(trk, &index));
    CHECK(uh_knn_build(index, map_desc.data(), NT, 32, 32));
    CHECK(uh_ba_create(map, &ba));
    CHECK(uh_ba_want_chi2(ba, 0));   // GlobalOptimizer::getResults returns poses, points and bad associations
    const uh_ba_params bp{5, 0.0, 0.0, 1.0f};
    // ---- outputs (pinned: the copies back are asynchronous DMA)
    uh_keypoint* kps = static_cast<uh_keypoint*>(uh_host_alloc((size_t)F * NFEAT * sizeof(uh_keypoint)));
    uint8_t* desc = static_cast<uint8_t*>(uh_host_alloc((size_t)F * NFEAT * 32));
    int32_t* counts = static_cast<int32_t*>(uh_host_alloc(64));
    int32_t* nn_idx = static_cast<int32_t*>(uh_host_alloc((size_t)F * NFEAT * NN * 4));
    int32_t* nn_dist = static_cast<int32_t*>(uh_host_alloc((size_t)F * NFEAT * NN * 4));
    std::vector<float> poses(16 * BA_K), points(3 * BA_P);
    std::vector<uint8_t> bad(40000);
    int32_t iters[2] = {0, 0};
    std::vector<uh_ba_problem> views;
    for (const auto& p : problems) views.push_back(p.view());

    int n_step = 0;
    auto step = [&]() -> int {
        const uh_ba_problem& pv = views[n_step++ % views.size()];
        CHECK(uh_ba_solve_async(ba, &pv, 0, 0, 0, &bp, nullptr));                                  // mapper: setParams (fresh problem) + optimize
        CHECK(uh_orb_extract_batch(ext, frames, W, H, W, (size_t)W * H, F, kps, desc, NFEAT, counts));   // tracker: 4 frames in, features out
        CHECK(uh_knn_search(index, desc, F * NFEAT, 32, NN, nn_idx, nn_dist, 0, -1));                // ... their match rows out
        CHECK(uh_ba_wait(ba));
        CHECK(uh_ba_get_results(ba, poses.data(), points.data(), nullptr, bad.data(), iters));     // tracker: getResults of the mapper's BA
        return 0;
    };
    for (int i = 0; i < 5; i++) if (step()) return 2;
    std::vector<double> ms;
    for (int r = 0; r < reps; r++) {
        const double t0 = now_ms();
        for (int i = 0; i < steps; i++) if (step()) return 2;
        ms.push_back((now_ms() - t0) / steps);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[ms.size() / 2];
    // per-phase BA timing on this thread (nothing else on the GPU)
    double t_set = 0, t_opt = 0, t_get = 0;
    for (int i = 0; i < 16; i++) {
        const double a = now_ms(); CHECK(uh_ba_set_problem(ba, &views[i % views.size()], &bp));
        const double b = now_ms(); CHECK(uh_ba_optimize(ba, nullptr));
        const double c = now_ms(); CHECK(uh_ba_get_results(ba, poses.data(), points.data(), nullptr, bad.data(), iters));
        const double d = now_ms();
        t_set += b - a; t_opt += c - b; t_get += d - c;
    }
    int lanes = 0;
    const int form = uh_ba_form(ba, &lanes);
    std::printf("{\"host\": \"c++ over the C ABI\", \"workload\": \"4 x (1241x376, 2000 features) + 8000 x 10000 nn10 + fresh local BA 10 x %d (%d observations)\", "
                "\"ms_per_step\": %.4f, \"ms_per_step_min\": %.4f, \"ms_per_step_max\": %.4f, \"frames_per_s\": %.1f, \"steps\": %d, \"reps\": %d, "
                "\"keypoints\": [%d, %d, %d, %d], \"first_row\": [%d, %d], \"ba_iters\": [%d, %d], \"ba_form\": %d, \"ba_lanes\": %d, "
                "\"ba_set_problem_ms\": %.4f, \"ba_optimize_ms\": %.4f, \"ba_get_results_ms\": %.4f}\n",
                (int)(problems[0].points.size() / 3), (int)problems[0].opt.size(), med, ms.front(), ms.back(), 1e3 * F / med, steps, reps, counts[0], counts[1], counts[2], counts[3],
                nn_idx[0], nn_dist[0], iters[0], iters[1], form, lanes, t_set / 16, t_opt / 16, t_get / 16);
    uh_ba_destroy(ba); uh_knn_destroy(index); uh_orb_destroy(ext);
    uh_host_free(frames); uh_host_free(kps); uh_host_free(desc); uh_host_free(counts); uh_host_free(nn_idx); uh_host_free(nn_dist);
    uh_ctx_destroy(map); uh_ctx_destroy(trk);
    return 0;
}
