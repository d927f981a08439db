// Microbenchmark: the floor of "update a 32 x 16 f64 tile that lives in LDS with a rank-8 product" on gfx950 — the inner step of the
// LDL^T trailing updates (csrc/ldlt_mfma.hpp, ldlt_rowlane2_lds in ba.hip), which cost ~1.3 k clocks per macro tile however
// their code was shaped.  One workgroup, W waves, every wave updates `iters` tiles of a row-stride matrix: 4 operand reads, 8
// accumulator reads, 4 MFMAs (two independent chains of two), 8 writes.  Variants: MFMA replaced by 8 FMAs; two macro tiles
// per iteration (loads of both issued first); accumulator traffic only.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_tile_update lds_tile_update.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT>   // MODE 0: full; 1: FMAs instead of MFMAs; 2: no arithmetic (copy); NT: macro tiles per iteration
__global__ __launch_bounds__(512) void k(long long* out, int iters, int ld) {
    extern __shared__ double M[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 18000; i += blockDim.x) M[i] = 1e-3 * (i & 255);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        d4 a0[NT], a1[NT]; double oa0[NT], oa1[NT], ob0[NT], ob1[NT]; int i0[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            const int rb = 32 + 32 * ((it * NT + u + wv) & 1), cb = 16 * ((it + u + wv) & 3);
            const int ab = (rb + i16) * ld + kq;
            oa0[u] = M[ab]; oa1[u] = M[ab + 16 * ld]; ob0[u] = M[(cb + i16) * ld + kq]; ob1[u] = M[(cb + i16) * ld + 4 + kq];
            i0[u] = (rb + kq) * ld + 40 + cb + i16;
#pragma unroll
            for (int v = 0; v < 4; v++) { a0[u][v] = M[i0[u] + 4 * v * ld]; a1[u][v] = M[i0[u] + (16 + 4 * v) * ld]; }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) {
            if (MODE == 0) {
                a0[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa0[u], ob0[u], a0[u], 0, 0, 0);
                a1[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa1[u], ob0[u], a1[u], 0, 0, 0);
                a0[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa0[u], ob1[u], a0[u], 0, 0, 0);
                a1[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa1[u], ob1[u], a1[u], 0, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int v = 0; v < 4; v++) { a0[u][v] = fma(oa0[u], ob0[u], a0[u][v]); a1[u][v] = fma(oa1[u], ob1[u], a1[u][v]); }
            }
        }
#pragma unroll
        for (int u = 0; u < NT; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) { M[i0[u] + 4 * v * ld] = a0[u][v] * 0.999; M[i0[u] + (16 + 4 * v) * ld] = a1[u][v] * 0.999; }
    }
    const long long t1 = clock64();
    if (lane == 0) out[wv] = t1 - t0;
}
template <int MODE, int NT>
void run(long long* d_out, int waves) {
    long long h[8];
    const int iters = 400;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(1), dim3(64 * waves), 150000, 0, d_out, iters, 129);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(1), dim3(64 * waves), 150000, 0, d_out, iters, 129);
    hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
    printf("%s, %d macro tile(s) per iteration, %d wave(s): %.0f clocks per macro tile (wave 0)\n", MODE == 0 ? "4 MFMAs " : (MODE == 1 ? "8 FMAs  " : "copy only"), NT, waves, (double)h[0] / (iters * NT));
}
int main() {
    long long* d_out; hipMalloc(&d_out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    for (int w : {1, 3, 4, 8}) { run<0, 1>(d_out, w); run<1, 1>(d_out, w); run<2, 1>(d_out, w); run<0, 2>(d_out, w); run<0, 4>(d_out, w); }
    return 0;
}
