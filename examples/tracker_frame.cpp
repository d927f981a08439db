// The reference tracker's per-frame chain over the C ABI, ONE frame at a time, host buffers in and out of every call — what a drop-in
// under UcoSlam::process() executes between two camera frames (reference file:line, statement starts of the token-pasted source):
//
//   FrameExtractor::process          ORB detectAndCompute + undistortPoints of the frame        uh_orb_extract_frame     frameextractor.cpp:430-520, :3985
//   Frame::create_kdtree             kd-tree over the undistorted keypoints                   uh_projmatch_set_frame   frameextractor.cpp:4258, map_types/frame.h:124
//                                    route "dev": the frame stays on the device, the tree is   uh_orb_extract_frame_dev + uh_projmatch_set_frame_dev
//                                    built there by one workgroup (csrc/kdbuild.hpp) — the same
//                                    tree, no host CPU time, but ~105 us on one compute unit
//                                    against ~50 us on a host core: not the default
//   tracker: previous-frame search   project the previous frame's map points, match          uh_projmatch_match_prev  utils/system.cpp:5930-6460 (call :6559-6565)
//   PnPSolver::solvePnp              pose from those matches (4 x 10 LM iterations)           uh_pnp_solve             optimization/pnpsolver.cpp:116-409 (call system.cpp:6626)
//   Map::matchFrameToMapPoints       the local map projected with the refined pose, 4-px disc  uh_projmatch_match       map.cpp:651-770 (call system.cpp:6897; radius :6762-6881)
//   PnPSolver::solvePnp              pose from the union of both match sets                   uh_pnp_solve             (call system.cpp:6954)
//
// The host glue between the calls is the reference's own (keypoint / map point look-ups per DMatch, pnpsolver.cpp:199-232; the second
// match set is appended to the first and filter_ambiguous_query runs over the union, system.cpp:6897-6954).
// Scene: a synthetic 1241 x 376 frame (rectangles + noise); its map is made FROM the frame's own features — a map point behind most
// keypoints (back-projected with a ground-truth pose, descriptor = the keypoint's with a few flipped bits), plus unrelated points —
// so every stage works on data the previous one produced.  Prints one JSON line with the median per-stage and per-frame latencies.
//
//   g++ -std=c++17 -O2 -o tracker_frame examples/tracker_frame.cpp -Lucoslam-cv3_amd -lucoslam_hip -Wl,-rpath,$PWD/ucoslam-cv3_amd -Wl,-rpath,/opt/rocm/lib -lpthread
//   ./tracker_frame [frames=200] [warmup=20] [route=host|dev|devhost|fused]     (dev: uh_orb_extract_frame_dev + uh_projmatch_set_frame_dev, the kd-tree built on the device; devhost: the same frame object, tree by the host core)
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

constexpr int W = 1241, H = 376, NFEAT = 2000, NLEV = 8, NSCENES = 4, N_PREV = 800, N_MAP = 3000;
constexpr float FX = 718.856f, FY = 718.856f, CX = 607.19f, CY = 185.22f;
constexpr float MAX_DESC_DIST = 50.f, PROJ_DIST_THR = 15.f;   // ORBextractor::getMinDescDistance (ORBextractor.h:105), Params::projDistThr (ucoslamtypes.cpp:49)

void make_frame(uint8_t* out, int shift_x, int shift_y, unsigned seed) {
    std::mt19937 scene(1234);
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

using Mat34 = std::array<double, 12>;   // R (row-major 3x3) | t : world -> camera

Mat34 se3_left(const Mat34& T, const double d[6]) {   // exp(d) T with a first-order rotation: enough for a few hundredths of a radian
    const double Rd[9] = {1, -d[2], d[1], d[2], 1, -d[0], -d[1], d[0], 1};
    Mat34 n{};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) n[3 * r + c] = Rd[3 * r] * T[c] + Rd[3 * r + 1] * T[3 + c] + Rd[3 * r + 2] * T[6 + c];
        n[9 + r] = Rd[3 * r] * T[9] + Rd[3 * r + 1] * T[10] + Rd[3 * r + 2] * T[11] + d[3 + r];
    }
    // re-orthonormalise the rows (Gram-Schmidt): the matchers take the matrix as a rigid transform
    double* R = n.data();
    auto nrm = [](double* v) { const double s = 1. / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= s; v[1] *= s; v[2] *= s; };
    nrm(R);
    const double d01 = R[0] * R[3] + R[1] * R[4] + R[2] * R[5];
    for (int c = 0; c < 3; c++) R[3 + c] -= d01 * R[c];
    nrm(R + 3);
    R[6] = R[1] * R[5] - R[2] * R[4]; R[7] = R[2] * R[3] - R[0] * R[5]; R[8] = R[0] * R[4] - R[1] * R[3];
    return n;
}

void to_f16(const Mat34& T, float* M) {
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[4 * r + c] = (float)T[3 * r + c]; M[4 * r + 3] = (float)T[9 + r]; }
    M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
}

struct Scene {
    uint8_t* image = nullptr;                 // pinned
    float pose0[16];                          // the tracker's predicted pose (ground truth, perturbed)
    Mat34 Tgt;
    // the previous frame's keypoints that carry a map point (system.cpp:5969-6089): id, position, octave, descriptor
    std::vector<uint32_t> prev_ids; std::vector<float> prev_pos; std::vector<int32_t> prev_oct; std::vector<uint8_t> prev_desc;
    // the local map after the reference's id filtering (map.cpp:657-668)
    std::vector<uint32_t> map_ids; std::vector<float> map_pos, map_nrm, map_min, map_max; std::vector<uint8_t> map_desc;
    std::vector<uint8_t> map_unstable;        // MapPoint::isStable() == false -> edge weight 0.5 (pnpsolver.cpp:215-216)
    std::vector<int32_t> id_to_map;           // map point id -> row of the arrays above (the reference: TheMap->map_points[id])
};

double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Stat {
    std::vector<double> v;
    void add(double x) { v.push_back(x); }
    double med() { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
    double lo() { return v.empty() ? 0 : *std::min_element(v.begin(), v.end()); }
    double p90() { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[(v.size() * 9) / 10]; }
};

}  // namespace

int main(int argc, char** argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 200, warmup = argc > 2 ? std::atoi(argv[2]) : 20;
    const bool fused_route = argc > 3 && std::strcmp(argv[3], "fused") == 0;     // devhost + the four tracker calls as ONE (uh_track_pose)
    const bool hybrid_route = fused_route || (argc > 3 && std::strcmp(argv[3], "devhost") == 0);   // frame resident on the device, its tree built by this core
    const bool dev_route = hybrid_route || (argc > 3 && std::strcmp(argv[3], "dev") == 0);
    uh_ctx* ctx = nullptr;
    if (uh_ctx_create_private(0, &ctx) < 0) { std::printf("no device: %s (there is no CPU path)\n", uh_last_error()); return 0; }
    uh_orb* ext = nullptr; uh_projmatch* pm = nullptr; uh_pnp* pnp = nullptr;
    CHECK(uh_orb_create(ctx, &ext));
    const uh_feat_params fp{2, NFEAT, NLEV, 1.2f, 0.f};
    CHECK(uh_orb_set_params(ext, &fp));
    CHECK(uh_projmatch_create(ctx, &pm));
    CHECK(uh_pnp_create(ctx, &pnp));
    uh_dev_frame* dfr = nullptr;
    CHECK(uh_dev_frame_create(ctx, &dfr));
    if (hybrid_route) CHECK(uh_dev_frame_set_tree_builder(dfr, 1));
    float sf[NLEV]; sf[0] = 1.f; for (int i = 1; i < NLEV; i++) sf[i] = sf[i - 1] * 1.2f;   // the extractor's float chain (ORBextractor.cpp:468-515)
    float inv_sf[NLEV]; for (int i = 0; i < NLEV; i++) inv_sf[i] = (float)(1. / sf[i]);         // pnpsolver.cpp:191-192
    const float intr[4] = {FX, FY, CX, CY};

    uh_keypoint* kps = static_cast<uh_keypoint*>(uh_host_alloc((size_t)NFEAT * sizeof(uh_keypoint)));
    uint8_t* desc = static_cast<uint8_t*>(uh_host_alloc((size_t)NFEAT * 32));
    float* und_xy = static_cast<float*>(uh_host_alloc((size_t)NFEAT * 2 * sizeof(float)));
    const uh_camera cam{FX, FY, CX, CY, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 5};   // a rectified (KITTI-like) camera: zero coefficients, the arithmetic still runs
    CHECK(uh_orb_set_camera(ext, &cam));

    // ---- scenes: one ORB extraction each, the map is made from its output
    std::vector<Scene> scenes(NSCENES);
    for (int s = 0; s < NSCENES; s++) {
        Scene& sc = scenes[s];
        sc.image = static_cast<uint8_t*>(uh_host_alloc((size_t)W * H));
        make_frame(sc.image, 2 * s, s, 1000 + s);
        int n = 0;
        CHECK(uh_orb_extract(ext, sc.image, W, H, W, kps, desc, NFEAT, &n));
        if (n < 200) { std::printf("error: scene %d has only %d keypoints\n", s, n); return 2; }
        std::mt19937 g(77 + s);
        std::uniform_real_distribution<double> U(0, 1);
        std::normal_distribution<double> N(0, 1);
        const double a = 0.01 + 0.004 * s, c = std::cos(a), sn = std::sin(a);
        sc.Tgt = {c, 0, sn, 0, 1, 0, -sn, 0, c, 0.3 + 0.1 * s, -0.05, 0.1};
        const Mat34& T = sc.Tgt;
        const double cam[3] = {-(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]), -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]), -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11])};
        auto back_project = [&](double u, double v, double z, double* X) {
            const double pc[3] = {(u - CX) / FX * z - T[9], (v - CY) / FY * z - T[10], z - T[11]};
            for (int i = 0; i < 3; i++) X[i] = T[i] * pc[0] + T[3 + i] * pc[1] + T[6 + i] * pc[2];   // R^T (Xc - t)
        };
        auto flipped = [&](const uint8_t* d, int max_flips, uint8_t* out) {
            std::memcpy(out, d, 32);
            const int nf = (int)(g() % (unsigned)(max_flips + 1));
            for (int i = 0; i < nf; i++) { const unsigned b = g() % 256u; out[b / 8] ^= (uint8_t)(1u << (b % 8)); }
        };
        uint32_t next_id = 10;
        std::vector<int> perm(n);
        for (int i = 0; i < n; i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), g);
        const int n_prev = std::min(N_PREV, n);
        std::sort(perm.begin(), perm.begin() + n_prev);   // keypoint order, as the reference's loop over the previous frame's ids
        for (int k = 0; k < n_prev; k++) {
            const uh_keypoint& kp = kps[perm[k]];
            double X[3];
            back_project(kp.x + 0.8 * N(g), kp.y + 0.8 * N(g), 4 + 36 * U(g), X);
            sc.prev_ids.push_back(next_id); next_id += 1 + g() % 3;
            for (int i = 0; i < 3; i++) sc.prev_pos.push_back((float)X[i]);
            sc.prev_oct.push_back(kp.octave);
            uint8_t d[32]; flipped(desc + 32 * (size_t)perm[k], 20, d);
            sc.prev_desc.insert(sc.prev_desc.end(), d, d + 32);
        }
        const int n_rel = N_MAP / 3;
        for (int k = 0; k < N_MAP; k++) {
            double X[3], nv[3], maxd, dist;
            uint8_t d[32];
            if (k < n_rel) {
                const int src = (int)(g() % (unsigned)n);
                const uh_keypoint& kp = kps[src];
                back_project(kp.x + 1.0 * N(g), kp.y + 1.0 * N(g), 3 + 42 * U(g), X);
                double view[3] = {cam[0] - X[0], cam[1] - X[1], cam[2] - X[2]};
                dist = std::sqrt(view[0] * view[0] + view[1] * view[1] + view[2] * view[2]);
                for (int i = 0; i < 3; i++) nv[i] = view[i] / dist + 0.3 * N(g);
                const int lev = std::min(NLEV - 1, std::max(0, kp.octave + (int)(g() % 3) - 1));
                maxd = dist * sf[lev] * (0.93 + 0.14 * U(g));
                flipped(desc + 32 * (size_t)src, 28, d);
            } else {   // unrelated: anywhere around the camera, random descriptor
                for (int i = 0; i < 3; i++) X[i] = 20 * N(g) + (i == 2 ? 10 : 0);
                dist = std::sqrt((cam[0] - X[0]) * (cam[0] - X[0]) + (cam[1] - X[1]) * (cam[1] - X[1]) + (cam[2] - X[2]) * (cam[2] - X[2]));
                for (int i = 0; i < 3; i++) nv[i] = N(g);
                maxd = dist * (0.5 + 2.5 * U(g));
                for (int i = 0; i < 32; i++) d[i] = (uint8_t)g();
            }
            const double nn = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
            sc.map_ids.push_back(next_id); next_id += 1 + g() % 3;
            for (int i = 0; i < 3; i++) { sc.map_pos.push_back((float)X[i]); sc.map_nrm.push_back((float)(nv[i] / nn)); }
            sc.map_max.push_back((float)maxd); sc.map_min.push_back((float)(maxd / sf[NLEV - 1]));
            sc.map_desc.insert(sc.map_desc.end(), d, d + 32);
            sc.map_unstable.push_back(U(g) < 0.2);
        }
        // (a random order, like a map's hash containers)
        {
            std::vector<int> o(N_MAP);
            for (int i = 0; i < N_MAP; i++) o[i] = i;
            std::shuffle(o.begin(), o.end(), g);
            Scene t = sc;
            for (int i = 0; i < N_MAP; i++) {
                const int j = o[i];
                sc.map_ids[i] = t.map_ids[j]; sc.map_min[i] = t.map_min[j]; sc.map_max[i] = t.map_max[j]; sc.map_unstable[i] = t.map_unstable[j];
                for (int c2 = 0; c2 < 3; c2++) { sc.map_pos[3 * i + c2] = t.map_pos[3 * j + c2]; sc.map_nrm[3 * i + c2] = t.map_nrm[3 * j + c2]; }
                std::memcpy(&sc.map_desc[32 * (size_t)i], &t.map_desc[32 * (size_t)j], 32);
            }
        }
        sc.id_to_map.assign(next_id, -1);
        for (int i = 0; i < N_MAP; i++) sc.id_to_map[sc.map_ids[i]] = i;
        // previous-frame points are map points too (ids disjoint from the local map's here: position look-up through prev arrays)
        const double dp[6] = {0.004 * N(g), 0.004 * N(g), 0.004 * N(g), 0.02 * N(g), 0.02 * N(g), 0.02 * N(g)};
        to_f16(se3_left(sc.Tgt, dp), sc.pose0);
    }

    // ---- the per-frame chain
    std::vector<uh_dmatch> m_prev(N_PREV), m_map(N_MAP), m_all;
    std::vector<float> p3d, kp2, isg, wgt;
    std::vector<uint8_t> bad;
    // uh_track_pose's view of the map bookkeeping: per previous-frame item its row in the local map (TheMap->map_points[id]), per map point its solver weight
    std::vector<std::vector<int32_t>> prev_row(scenes.size());
    std::vector<std::vector<float>> map_w(scenes.size());
    for (size_t si = 0; si < scenes.size(); si++) {
        Scene& sc = scenes[si];
        for (uint32_t id : sc.prev_ids) prev_row[si].push_back((size_t)id < sc.id_to_map.size() ? sc.id_to_map[id] : -1);
        for (int i = 0; i < N_MAP; i++) map_w[si].push_back(sc.map_unstable[i] ? 0.5f : 1.f);
    }
    std::vector<uh_dmatch> f_prev(N_PREV + 1), f_map(N_MAP + 1), f_all(N_PREV + N_MAP + 1);
    std::vector<uint8_t> f_bad1(N_PREV + 1), f_bad2(N_PREV + N_MAP + 1);
    Stat t_orb, t_set, t_prev, t_pnp1, t_map, t_pnp2, t_glue, t_frame;
    long sum_prev = 0, sum_map = 0, sum_in1 = 0, sum_in2 = 0, sum_kp = 0;
    double pose_err = 0;
    for (int it = -warmup; it < frames; it++) {
        Scene& sc = scenes[(it + warmup) % NSCENES];
        const double t0 = now_us();
        int n = 0;
        // FrameExtractor::process: detectAndCompute + undistortPoints(kpts, ImageParams) in one call (frameextractor.cpp:430-520, :3985;
        // misc.cpp:269-293) — Frame::und_kpts = the keypoints with the undistorted positions
        double t1, t2;
        if (fused_route) {
            // the extraction in two halves: the undistorted keypoints (position + octave) come back as soon as the selection is done, the tree is
            // built from them while the descriptors are still being computed
            const uh_keypoint* early = nullptr;
            CHECK(uh_orb_extract_frame_dev_begin(ext, sc.image, W, H, W, 1, kps, desc, und_xy, NFEAT, &n, dfr, &early));
            t1 = now_us();
            const uh_proj_frame fre{early, n, desc, sf, NLEV, FX, FY, CX, CY, 0, 0, W, H};
            CHECK(uh_projmatch_set_frame_dev(pm, dfr, &fre));
            CHECK(uh_orb_extract_frame_dev_end(ext, &n));
            t2 = now_us();
        } else {
        if (dev_route) CHECK(uh_orb_extract_frame_dev(ext, sc.image, W, H, W, 1, kps, desc, und_xy, NFEAT, &n, dfr));
        else CHECK(uh_orb_extract_frame(ext, sc.image, W, H, W, 1, kps, desc, und_xy, NFEAT, &n));
        for (int i = 0; i < n; i++) { kps[i].x = und_xy[2 * i]; kps[i].y = und_xy[2 * i + 1]; }
        t1 = now_us();
        const uh_proj_frame fr{kps, n, desc, sf, NLEV, FX, FY, CX, CY, 0, 0, W, H};
        if (dev_route) CHECK(uh_projmatch_set_frame_dev(pm, dfr, &fr));
        else CHECK(uh_projmatch_set_frame(pm, &fr));
        t2 = now_us();
        }
        const uh_prev_points pp{(int32_t)sc.prev_ids.size(), sc.prev_ids.data(), sc.prev_pos.data(), sc.prev_oct.data(), sc.prev_desc.data()};
        if (fused_route) {
            const size_t si = (size_t)((it + warmup) % NSCENES);
            const uh_map_points mpf{N_MAP, sc.map_ids.data(), sc.map_pos.data(), sc.map_nrm.data(), sc.map_min.data(), sc.map_max.data(), sc.map_desc.data()};
            const uh_track_args ta{sc.pose0, intr, inv_sf, NLEV, &pp, prev_row[si].data(), &mpf, map_w[si].data(), MAX_DESC_DIST * 1.5f, PROJ_DIST_THR, MAX_DESC_DIST * 2.f, 4.f, PROJ_DIST_THR, 30};
            uh_track_result tr{};
            tr.matches_prev = f_prev.data(); tr.bad_prev = f_bad1.data(); tr.cap_prev = (int32_t)f_prev.size();
            tr.matches_map = f_map.data(); tr.cap_map = (int32_t)f_map.size();
            tr.matches_all = f_all.data(); tr.bad_all = f_bad2.data(); tr.cap_all = (int32_t)f_all.size();
            CHECK(uh_track_pose(pm, pnp, &ta, &tr));
            const double t8f = now_us();
            if (it < 0) continue;
            t_orb.add(t1 - t0); t_set.add(t2 - t1); t_prev.add(t8f - t2); t_frame.add(t8f - t0);
            sum_prev += tr.n_prev; sum_map += tr.n_map; sum_in1 += tr.inliers1; sum_in2 += tr.inliers2; sum_kp += n;
            float Mgf[16]; to_f16(sc.Tgt, Mgf);
            double ef = 0; for (int i = 0; i < 12; i++) ef = std::max(ef, (double)std::fabs(Mgf[i] - tr.pose2[i]));
            pose_err = std::max(pose_err, ef);
            continue;
        }
        const int n1 = uh_projmatch_match_prev(pm, sc.pose0, &pp, MAX_DESC_DIST * 1.5f, PROJ_DIST_THR, m_prev.data(), (int)m_prev.size(), nullptr, nullptr);
        CHECK(n1);
        const double t3 = now_us();
        // PnPSolver::solvePnp's per-match look-ups (pnpsolver.cpp:199-232); previous-frame ids index the prev arrays here
        p3d.resize(3 * (size_t)n1); kp2.resize(2 * (size_t)n1); isg.resize(n1); wgt.resize(n1); bad.resize(std::max(n1, 1));
        {
            size_t cur = 0;   // prev_ids ascend and the matches come in item order: one merge pass
            for (int i = 0; i < n1; i++) {
                while (sc.prev_ids[cur] != (uint32_t)m_prev[i].trainIdx) ++cur;
                const uh_keypoint& k = kps[m_prev[i].queryIdx];
                for (int c2 = 0; c2 < 3; c2++) p3d[3 * i + c2] = sc.prev_pos[3 * cur + c2];
                kp2[2 * i] = k.x; kp2[2 * i + 1] = k.y; isg[i] = inv_sf[k.octave]; wgt[i] = 1.f;
            }
        }
        const double t4 = now_us();
        float pose1[16]; int32_t iters[4];
        const int in1 = uh_pnp_solve(pnp, sc.pose0, intr, n1, p3d.data(), kp2.data(), isg.data(), wgt.data(), pose1, bad.data(), iters, nullptr);
        CHECK(in1);
        const double t5 = now_us();
        const uh_map_points mp{N_MAP, sc.map_ids.data(), sc.map_pos.data(), sc.map_nrm.data(), sc.map_min.data(), sc.map_max.data(), sc.map_desc.data()};
        // system.cpp:6762-6881: with at least 30 inliers the refined pose is kept and the local map is searched in a 4-pixel disc; otherwise
        // the first matches are dropped, the predicted pose stays and the search radius is projDistThr again
        const bool tracked = in1 >= 30;
        const float* pose_for_map = tracked ? pose1 : sc.pose0;
        const int n2 = uh_projmatch_match(pm, pose_for_map, &mp, MAX_DESC_DIST * 2.f, tracked ? 4.f : PROJ_DIST_THR, m_map.data(), (int)m_map.size(), nullptr, nullptr, nullptr);
        CHECK(n2);
        const double t6 = now_us();
        // system.cpp:6897-6954: inliers of the first set + the new matches, filter_ambiguous_query over the union, then the per-match look-ups
        m_all.clear();
        if (tracked) for (int i = 0; i < n1; i++) if (!bad[i]) m_all.push_back(m_prev[i]);
        const int kept1 = (int)m_all.size();
        m_all.insert(m_all.end(), m_map.begin(), m_map.begin() + n2);
        const int na = m_all.empty() ? 0 : uh_filter_ambiguous(m_all.data(), (int)m_all.size(), 0);
        CHECK(na);
        p3d.resize(3 * (size_t)na); kp2.resize(2 * (size_t)na); isg.resize(na); wgt.resize(na); bad.resize(std::max(na, 1));
        for (int i = 0; i < na; i++) {
            const uh_dmatch& m = m_all[i];
            const uh_keypoint& k = kps[m.queryIdx];
            const int row = (size_t)m.trainIdx < sc.id_to_map.size() ? sc.id_to_map[m.trainIdx] : -1;
            if (row >= 0) { for (int c2 = 0; c2 < 3; c2++) p3d[3 * i + c2] = sc.map_pos[3 * row + c2]; wgt[i] = sc.map_unstable[row] ? 0.5f : 1.f; }
            else {
                const size_t cur = std::lower_bound(sc.prev_ids.begin(), sc.prev_ids.end(), (uint32_t)m.trainIdx) - sc.prev_ids.begin();
                for (int c2 = 0; c2 < 3; c2++) p3d[3 * i + c2] = sc.prev_pos[3 * cur + c2];
                wgt[i] = 1.f;
            }
            kp2[2 * i] = k.x; kp2[2 * i + 1] = k.y; isg[i] = inv_sf[k.octave];
        }
        const double t7 = now_us();
        float pose2[16];
        const int in2 = uh_pnp_solve(pnp, pose_for_map, intr, na, p3d.data(), kp2.data(), isg.data(), wgt.data(), pose2, bad.data(), iters, nullptr);
        CHECK(in2);
        const double t8 = now_us();
        if (it < 0) continue;
        (void)kept1;
        t_orb.add(t1 - t0); t_set.add(t2 - t1); t_prev.add(t3 - t2); t_pnp1.add(t5 - t4); t_map.add(t6 - t5); t_pnp2.add(t8 - t7);
        t_glue.add((t4 - t3) + (t7 - t6)); t_frame.add(t8 - t0);
        sum_prev += n1; sum_map += n2; sum_in1 += in1; sum_in2 += in2; sum_kp += n;
        float Mg[16]; to_f16(sc.Tgt, Mg);
        double e = 0; for (int i = 0; i < 12; i++) e = std::max(e, (double)std::fabs(Mg[i] - pose2[i]));
        pose_err = std::max(pose_err, e);
    }
    const double f = frames > 0 ? 1.0 / frames : 0;
    std::printf("{\"host\": \"c++ over the C ABI, one frame at a time, host in / host out\", \"frame_route\": \"%s\", \"frames\": %d, \"tracker_frame_ms\": %.4f, \"tracker_frame_ms_min\": %.4f, "
                "\"tracker_frame_ms_p90\": %.4f, \"tracker_frames_per_s\": %.1f, "
                "\"orb_extract_ms\": %.4f, \"set_frame_ms\": %.4f, \"match_prev_ms\": %.4f, \"pnp1_ms\": %.4f, \"match_map_ms\": %.4f, \"pnp2_ms\": %.4f, \"host_glue_ms\": %.4f, "
                "\"keypoints\": %.1f, \"prev_items\": %d, \"map_points\": %d, \"matches_prev\": %.1f, \"matches_map\": %.1f, \"inliers1\": %.1f, \"inliers2\": %.1f, "
                "\"max_pose_err_vs_truth\": %.5f}\n",
                fused_route ? "device-resident frame, kd-tree built by the host core, searches + solves as one call (uh_track_pose; its time is reported as match_prev_ms)" : hybrid_route ? "device-resident frame, kd-tree built by the host core and uploaded" : dev_route ? "device-resident frame, kd-tree built on the device" : "keypoints to the host, kd-tree built on the host", frames, t_frame.med() / 1e3, t_frame.lo() / 1e3, t_frame.p90() / 1e3, 1e6 / std::max(t_frame.med(), 1e-9), t_orb.med() / 1e3, t_set.med() / 1e3, t_prev.med() / 1e3,
                t_pnp1.med() / 1e3, t_map.med() / 1e3, t_pnp2.med() / 1e3, t_glue.med() / 1e3, sum_kp * f, N_PREV, N_MAP, sum_prev * f, sum_map * f, sum_in1 * f, sum_in2 * f, pose_err);
    uh_pnp_destroy(pnp); uh_projmatch_destroy(pm); uh_orb_destroy(ext); uh_dev_frame_destroy(dfr);
    for (auto& s : scenes) uh_host_free(s.image);
    uh_host_free(kps); uh_host_free(desc);
    uh_ctx_destroy(ctx);
    return 0;
}
