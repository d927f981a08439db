// Microbenchmark: cost of a short burst of v_mfma_f64_16x16x4_f64 between stretches of vector work (gfx950).
// The LDL^T trailing update issues three MFMAs per 16 x 16 tile between LDS traffic; its measured per-tile time (1.4 k clocks) was
// far above 3 x the back-to-back issue interval, so this separates: burst length, the accumulators' way in and out (AGPR vs VGPR),
// and the length of the vector stretch between bursts.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_burst mfma_f64_burst.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int K, int FILL, bool AGPR>
__global__ void k(long long* out, double* sink, int iters) {
    d4 acc = {1, 2, 3, 4};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4, f = 0.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        d4 a = acc;
#pragma unroll
        for (int s = 0; s < K; s++) {
            if constexpr (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(a) : "v"(x), "v"(y));
            else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
        }
        if (K > 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        acc = a;
#pragma unroll
        for (int j = 0; j < FILL; j++) f = fma(f, x, y);
        acc[0] += f;
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + f;
}
template <int K, int FILL, bool AGPR>
void run(long long* d_out, double* d_sink) {
    const int iters = 2000;
    long long h;
    for (int threads : {64, 256}) {
        hipLaunchKernelGGL((k<K, FILL, AGPR>), dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters);
        hipLaunchKernelGGL((k<K, FILL, AGPR>), dim3(1), dim3(threads), 0, 0, d_out, d_sink, iters);
        hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        printf("burst of %d MFMA (%s accumulators) + %3d dependent v_fma_f64, %d waves: %.0f clocks per iteration\n", K, AGPR ? "AGPR" : "VGPR", FILL, threads / 64, (double)h / iters);
    }
}
int main() {
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, 64 * 8); hipMalloc(&d_sink, 1 << 20);
    run<0, 0, true>(d_out, d_sink); run<0, 100, true>(d_out, d_sink);
    run<1, 0, true>(d_out, d_sink); run<1, 100, true>(d_out, d_sink);
    run<3, 0, true>(d_out, d_sink); run<3, 100, true>(d_out, d_sink);
    run<6, 100, true>(d_out, d_sink);
    run<1, 100, false>(d_out, d_sink); run<3, 100, false>(d_out, d_sink); run<6, 100, false>(d_out, d_sink);
    return 0;
}
