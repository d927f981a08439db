// Compiles the C++ adaptors (no OpenCV, no GPU needed) and exercises the CPU-visible error paths of the C ABI.  With a directory as
// argument (tests/test_abi.py::test_cpp_adaptors_run_on_gpu) EVERY adaptor class runs once on the GPU on inputs the test wrote there,
// and leaves its outputs for the test to compare with the oracle / golden vectors:
//   frame.bin  -> ORBextractor::detectAndCompute          -> orb_out.bin
//   voc.bin    -> Vocabulary::transform(level 3), fBow::score -> bow_out.bin;  FrameMatcherBoW::matchEpipolar (frame vs itself) -> bowmatch_out.bin
//   ba.bin     -> a toy map -> GlobalOptimizer::setParams(map) / optimize / getResults(map) -> ba_out.bin   (a flattened synthetic local BA)
#include <cstdio>
#include <fstream>
#include "../../include/ucoslam_hip/adaptors.hpp"
#include "toy_map.hpp"

static std::vector<char> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot read " + path);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T> static void put(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T))); }
template <class T> static void put1(std::ofstream& f, T v) { put(f, &v, 1); }

static int run_all_adaptors(std::shared_ptr<ucoslam_hip::Context> ctx, const std::string& dir) {
    using namespace ucoslam_hip;
    // ---- extractor
    std::vector<uh_keypoint> kps;
    std::vector<uint8_t> desc;
    {
        const std::vector<char> fr = slurp(dir + "/frame.bin");
        int32_t wh[2];
        std::memcpy(wh, fr.data(), 8);
        auto ext = ORBextractor::create(ctx);
        ext->detectAndCompute(reinterpret_cast<const uint8_t*>(fr.data() + 8), wh[0], wh[1], (size_t)wh[0], kps, desc, FeatParams(1000, 8, 1.2f, 1));
        if (ext->getParams().maxFeatures != 1000 || ext->getMinDescDistance() != 50) return 10;
        std::ofstream o(dir + "/orb_out.bin", std::ios::binary);
        put1<int32_t>(o, (int32_t)kps.size());
        put(o, kps.data(), kps.size());
        put(o, desc.data(), desc.size());
    }
    // ---- vocabulary + BoW matcher (the frame against itself: every keypoint's best candidate is itself at distance 0)
    {
        const std::vector<char> vs = slurp(dir + "/voc.bin");
        Vocabulary voc(ctx);
        voc.fromStream(vs.data(), vs.size());
        fBow bag; fBow2 nodes;
        voc.transform(desc.data(), (int)kps.size(), 32, 32, 3, bag, nodes);
        const fBow nbag = voc.transform(desc.data(), (int)kps.size(), 32, 32);
        std::ofstream o(dir + "/bow_out.bin", std::ios::binary);
        put1<int32_t>(o, (int32_t)bag.size());
        for (auto& e : bag) { put1<uint32_t>(o, e.first); put1<float>(o, e.second); }
        put1<int32_t>(o, (int32_t)nodes.size());
        for (auto& e : nodes) { put1<uint32_t>(o, e.first); put1<int32_t>(o, (int32_t)e.second.size()); put(o, e.second.data(), e.second.size()); }
        put1<double>(o, fBow::score(nbag, nbag));
        std::vector<uint32_t> ids; std::vector<int32_t> ptr{0}; std::vector<uint32_t> feat;
        for (auto& e : nodes) { ids.push_back(e.first); feat.insert(feat.end(), e.second.begin(), e.second.end()); ptr.push_back((int32_t)feat.size()); }
        std::vector<int32_t> oct(kps.size()); std::vector<float> ang(kps.size()), pt(2 * kps.size());
        for (size_t i = 0; i < kps.size(); i++) { oct[i] = kps[i].octave; ang[i] = kps[i].angle; pt[2 * i] = kps[i].x; pt[2 * i + 1] = kps[i].y; }
        const uh_bow_frame f{(int32_t)ids.size(), ids.data(), ptr.data(), feat.data(), (int32_t)kps.size(), desc.data(), oct.data(), ang.data(), pt.data(), nullptr};
        const float sf[8] = {1.f, 1.2f, 1.44f, 1.728f, 2.0736f, 2.48832f, 2.985984f, 3.5831808f};
        FrameMatcherBoW fm(ctx);
        const std::vector<uh_dmatch> mm = fm.matchEpipolar(uh_bow_match_args{f, f, sf, 8, nullptr, 100.f, 0.8f, 1, 1});
        std::ofstream om(dir + "/bowmatch_out.bin", std::ios::binary);
        put1<int32_t>(om, (int32_t)mm.size());
        put(om, mm.data(), mm.size());
    }
    // ---- bundle adjustment through a map: the flattened problem becomes a toy map (frames, points, observer lists), the
    // adaptor flattens it again by the reference's rules, optimises, and writes the results back into the map
    {
        const std::vector<char> b = slurp(dir + "/ba.bin");
        const char* p = b.data();
        int32_t KPE[3];
        std::memcpy(KPE, p, 12); p += 12;
        const int K = KPE[0], P = KPE[1], E = KPE[2];
        const float* poses = reinterpret_cast<const float*>(p); p += 64 * (size_t)K;
        const uint8_t* fixed = reinterpret_cast<const uint8_t*>(p); p += (K + 3) & ~3;
        const float* intr = reinterpret_cast<const float*>(p); p += 16 * (size_t)K;
        const float* pts = reinterpret_cast<const float*>(p); p += 12 * (size_t)P;
        const int32_t* opt = reinterpret_cast<const int32_t*>(p); p += 4 * (size_t)E;
        const int32_t* okf = reinterpret_cast<const int32_t*>(p); p += 4 * (size_t)E;
        const float* ouv = reinterpret_cast<const float*>(p); p += 8 * (size_t)E;
        const int32_t* ooct = reinterpret_cast<const int32_t*>(p); p += 4 * (size_t)E;
        const float* sfs = reinterpret_cast<const float*>(p); p += 32;   // Frame::scaleFactors, 8 levels
        ToyMap m;
        m.frames.resize(K); m.points.resize(P);
        m.sf.assign(sfs, sfs + 8);
        for (int k = 0; k < K; k++) { m.frames[k].valid = true; m.order.push_back(k); std::memcpy(m.frames[k].pose, poses + 16 * k, 64); std::memcpy(m.frames[k].intr, intr + 4 * k, 16); }
        for (int q = 0; q < P; q++) std::memcpy(m.points[q].xyz, pts + 3 * q, 12);
        for (int e = 0; e < E; e++) m.observe(opt[e], okf[e], ouv[2 * e], ouv[2 * e + 1], ooct[e]);
        GlobalOptimizer::ParamSet ps;
        ps.nIters = 5; ps.fixFirstFrame = false;
        for (int k = 0; k < K; k++) if (fixed[k]) ps.fixed_frames.insert(k);
        auto opt_ = GlobalOptimizer::create(ctx, "hip");
        if (opt_->getName() != "hip") return 20;
        bool threw = false;
        try { GlobalOptimizer::create(ctx, "ceres"); } catch (const std::runtime_error&) { threw = true; }   // globaloptimizer.cpp:27-33
        if (!threw) return 21;
        opt_->setParams(m, ps);
        bool stop = false;
        opt_->optimize(&stop);
        opt_->getResults(m);
        const auto bad = opt_->getBadAssociations();
        std::ofstream o(dir + "/ba_out.bin", std::ios::binary);
        put1<int32_t>(o, (int32_t)opt_->index().frame_of.size()); put1<int32_t>(o, (int32_t)opt_->index().point_of.size()); put1<int32_t>(o, opt_->index().n_obs);
        for (int k = 0; k < K; k++) put(o, m.frames[k].pose, 16);
        for (int q = 0; q < P; q++) put(o, m.points[q].xyz, 3);
        put1<int32_t>(o, (int32_t)bad.size());
        for (auto& pr : bad) { put1<uint32_t>(o, pr.first); put1<uint32_t>(o, pr.second); }
        int upd = 0;
        for (int q = 0; q < P; q++) upd += m.points[q].normals_updated;
        put1<int32_t>(o, upd);
    }
    std::printf("all adaptors ran\n");
    return 0;
}

int main(int argc, char** argv) {
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
        if (argc > 1) { const int rc = run_all_adaptors(ctx, argv[1]); if (rc) return rc; }
    } catch (const std::runtime_error& e) {
        std::printf("no device: %s\n", e.what());   // expected on the CPU-only build box: no fallback exists
    }
    return 0;
}
