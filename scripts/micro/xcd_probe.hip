// Probes two hardware behaviours a same-XCD pyramid kernel relies on (scripts/micro, not part of the library; the kernel itself was measured
// and not kept: DESIGN.md section 8, "a frame per XCD"):
//  1. which XCD (HW_REG_XCC_ID) workgroup i of a launch runs on;
//  2. whether a buffer load with the sc0 bit (aux = 1) sees data another CU of the SAME XCD stored after this CU had the line in its L1.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/xcd_probe.hip -o scripts/micro/xcd_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_ids(int* out) {
    if (threadIdx.x == 0) {
        int x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        int cu;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
        out[2 * blockIdx.x] = x & 0xf;
        out[2 * blockIdx.x + 1] = cu;
    }
}
// workgroups 0 and 8 (same XCD if the dispatch is round-robin over 8 XCDs): B reads the buffer (L1 now holds it), signals, A rewrites it and
// signals back, B re-reads with plain loads and with sc0 loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_coh(unsigned* buf, unsigned* flags, unsigned* result, int n, int round) {
    const int role = blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : -1);
    if (role < 0) return;
    const unsigned tag = 0x1000u * (round + 1);
    if (role == 1) {   // B
        unsigned s = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s += buf[i];   // warm L1 with the old contents
        __syncthreads();
        if (threadIdx.x == 0) { result[0] = s; __hip_atomic_store(flags + 0, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        if (threadIdx.x == 0) while (__hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        unsigned stale_plain = 0, stale_sc0 = 0;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, n * 4, 0x00020000);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned v0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, i * 4, 0, 1);   // sc0
            stale_sc0 += v0 != tag + i;
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) stale_plain += ((volatile unsigned*)buf)[i] != tag + i;
        atomicAdd(result + 1, stale_plain);
        atomicAdd(result + 2, stale_sc0);
    } else {   // A
        if (threadIdx.x == 0) while (__hip_atomic_load(flags + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = tag + i;   // plain stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    int* d; hipMalloc(&d, 512 * 8);
    hipLaunchKernelGGL(k_ids, dim3(256), dim3(64), 0, 0, d);
    std::vector<int> h(512); hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    printf("xcc of workgroups 0..31:"); for (int i = 0; i < 32; i++) printf(" %d", h[2 * i]); printf("\n");
    int bad = 0; for (int i = 0; i < 256; i++) bad += h[2 * i] != (h[0] + i) % 8;
    printf("workgroups whose xcc != (xcc0 + i) mod 8: %d of 256\n", bad);
    unsigned *buf, *flags, *res; const int n = 16384;
    hipMalloc(&buf, n * 4); hipMalloc(&flags, 64); hipMalloc(&res, 64);
    hipMemset(buf, 0, n * 4); hipMemset(flags, 0, 64);
    for (int r = 0; r < 20; r++) {
        hipMemset(res, 0, 64);
        hipLaunchKernelGGL(k_coh, dim3(16), dim3(256), 0, 0, buf, flags, res, n, r);
        hipDeviceSynchronize();
        unsigned hr[4]; hipMemcpy(hr, res, 16, hipMemcpyDeviceToHost);
        if (r < 3 || hr[1] || hr[2]) printf("round %d: stale words with plain loads %u, with sc0 loads %u (of %d)\n", r, hr[1], hr[2], n);
    }
    printf("done\n");
    return 0;
}
