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
    } catch (const std::runtime_error& e) {
        std::printf("no device: %s\n", e.what());   // expected on the CPU-only build box: no fallback exists
    }
    return 0;
}
