// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on gfx950 (cycles per instruction, one wave per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(long long* out, double* sink, int iters, int chains) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(a0) : "v"(x), "v"(y));
        if (chains > 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(a1) : "v"(x), "v"(y));
        if (chains > 2) { asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(a2) : "v"(x), "v"(y));
                          asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(a3) : "v"(x), "v"(y)); }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void k_fma(long long* out, double* sink, int iters) {
    double a[16];
    for (int j = 0; j < 16; j++) a[j] = j;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a[j] = fma(a[j], x, y);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    double s = 0; for (int j = 0; j < 16; j++) s += a[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    long long* d_out; double* d_sink; long long h[4];
    hipMalloc(&d_out, 64 * 8); hipMalloc(&d_sink, 1 << 20);
    const int iters = 2000;
    for (int threads : {64, 256}) for (int chains : {1, 2, 4}) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters, chains);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters, chains);
        hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4: %d waves/CU, %d independent chains: %.1f shader cycles per MFMA (1024 FMA each)\n", threads / 64, chains, (double)h[0] / (iters * (chains == 4 ? 4 : chains)));
    }
    for (int threads : {64, 256}) {
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters);
        hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64: %d waves/CU, 16 independent chains: %.2f shader cycles per wave instruction (64 FMA each)\n", threads / 64, (double)h[0] / (iters * 16));
    }
    // wall-clock check of the shader clock
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, d_out, d_sink, 200000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
    printf("shader clock: %.0f MHz\n", h[0] / (ms * 1e3));
    return 0;
}
