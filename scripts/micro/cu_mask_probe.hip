// Probe: what a CU mask of hipExtStreamCreateWithCUMask selects on gfx950 (8 XCDs x 32 CUs): for a few masks, which (XCC, SE, CU) the
// workgroups of a long-enough launch land on.  build: hipcc --offload-arch=gfx950 -O3 scripts/micro/cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void k_ids(int* out, int spin) {
    if (threadIdx.x == 0) {
        int x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = x & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char* name, std::vector<uint32_t> mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nwg = 2048;
    int* d; hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL(k_ids, dim3(nwg), dim3(64), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<int> h(2 * nwg); hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nwg; i++) per_xcc[h[2 * i]].insert(h[2 * i + 1] & 0xffff00);   // HW_ID without wave / simd ids: (cu, sh, se, ...)
    int total = 0;
    printf("%s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  -> %d distinct CUs\n", total);
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    run("all 256 bits        ", std::vector<uint32_t>(8, 0xffffffffu));
    run("low 96 bits         ", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0});
    run("low 128 bits        ", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
    run("high 160 bits       ", {0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    run("every 8th bit from 0", std::vector<uint32_t>(8, 0x01010101u));
    run("bits 0-2 of each 8  ", std::vector<uint32_t>(8, 0x07070707u));
    return 0;
}
